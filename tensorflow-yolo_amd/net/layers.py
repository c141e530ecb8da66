"""Layer vocabulary of the HIP backend.

Mirrors the class names and constructor arguments of the reference's net/layers.py:17-134 so
that network builders read the same, but nothing here touches a tensor: each object only
records topology (its producer layers and static shape).  The list of these objects is what
`engine.HipNetwork` lowers, through the C ABI, to fused gfx950 kernels.

`.out` is a `Symbol` (the stand-in for the reference's tf.Tensor); `.variable_names` keeps the
Darknet stream order (beta, gamma, moving_mean, moving_variance, kernel | bias, kernel --
reference net/layers.py:53-63).
"""
from .. import _hip

BATCH_NORM_EPSILON = 1e-5   # folded at load time by libyolo_hip (reference net/layers.py:5)
LEAKY_RELU = 0.1            # conv epilogue slope (reference net/layers.py:6)


class _StaticShape(tuple):
    def as_list(self):
        return list(self)


class Symbol(object):
    """Static NHWC shape (batch = None) + the layer that produces it."""

    def __init__(self, layer, h, w, c):
        self.layer = layer
        self.shape = _StaticShape((None, int(h), int(w), int(c)))

    def get_shape(self):
        return self.shape

    @property
    def hwc(self):
        return self.shape[1], self.shape[2], self.shape[3]


class _Layer(object):
    op = None

    def __init__(self, inputs):
        self.inputs = [s.layer for s in inputs]
        self.variable_names = []
        self.index = None       # position in the network list, set by engine.number_layers

    def fill_desc(self, d):
        pass


def _out_hw(h, w, ksize, stride):
    # stride 1 -> SAME (shape kept); stride > 1 -> zero pad k-1 in total, then VALID
    if stride == 1:
        return h, w
    return (h + ksize - 1 - ksize) // stride + 1, (w + ksize - 1 - ksize) // stride + 1


class conv2d_bn_act(_Layer):
    """conv (+ folded BN) (+ leaky 0.1).  Reference: net/layers.py:17-67."""
    op = _hip.OP_CONV
    name_count = 0

    def __init__(self, prev, filter_size, kernel_size, stride=1, use_batch_normalization=True,
                 activation_fn="leaky", is_training=False, scope="yolo"):
        if is_training:
            raise NotImplementedError("the HIP backend is inference-only (TEST mode)")
        super(conv2d_bn_act, self).__init__([prev])
        name = "{}/{}_{}".format(scope, type(self).__name__, conv2d_bn_act.name_count)
        conv2d_bn_act.name_count += 1
        self.filters, self.ksize, self.stride = int(filter_size), int(kernel_size), int(stride)
        self.batch_norm = bool(use_batch_normalization)
        self.activation = activation_fn
        h, w, cin = prev.hwc
        self.in_channels = cin
        oh, ow = _out_hw(h, w, self.ksize, self.stride)
        self.out = Symbol(self, oh, ow, self.filters)
        tail = ("beta", "gamma", "moving_mean", "moving_variance") if self.batch_norm else ("bias",)
        self.variable_names = ["{}/{}".format(name, t) for t in tail + ("kernel",)]

    @staticmethod
    def reset():
        conv2d_bn_act.name_count = 0

    def weight_count(self):
        k = self.filters * self.in_channels * self.ksize * self.ksize
        return k + (4 if self.batch_norm else 1) * self.filters

    def fill_desc(self, d):
        d.filters, d.ksize, d.stride = self.filters, self.ksize, self.stride
        d.batch_norm = int(self.batch_norm)
        d.leaky = int(self.activation == "leaky")


class max_pool2d(_Layer):
    """Reference: net/layers.py:70-81."""
    op = _hip.OP_MAXPOOL

    def __init__(self, prev, kernel_size, stride=2):
        super(max_pool2d, self).__init__([prev])
        self.ksize, self.stride = int(kernel_size), int(stride)
        h, w, c = prev.hwc
        oh, ow = _out_hw(h, w, self.ksize, self.stride)
        self.out = Symbol(self, oh, ow, c)

    def fill_desc(self, d):
        d.ksize, d.stride = self.ksize, self.stride


class route(_Layer):
    """Channel concat (single input = alias).  Reference: net/layers.py:84-87."""
    op = _hip.OP_ROUTE

    def __init__(self, prevs):
        super(route, self).__init__(list(prevs))
        h, w, _ = prevs[0].hwc
        self.out = Symbol(self, h, w, sum(p.hwc[2] for p in prevs))


class reorg(_Layer):
    """Block-major space-to-depth.  Reference: net/layers.py:90-97."""
    op = _hip.OP_REORG

    def __init__(self, prev, stride):
        super(reorg, self).__init__([prev])
        self.stride = int(stride)
        h, w, c = prev.hwc
        self.out = Symbol(self, h // self.stride, w // self.stride, c * self.stride * self.stride)

    def fill_desc(self, d):
        d.stride = self.stride


class shortcut(_Layer):
    """Residual add, no activation after.  Reference: net/layers.py:100-103."""
    op = _hip.OP_SHORTCUT

    def __init__(self, prev, shortcut_out):
        super(shortcut, self).__init__([prev, shortcut_out])
        self.out = Symbol(self, *prev.hwc)


class input_layer(_Layer):
    """Feed point; shape = [None, h, w, c].  Reference: net/layers.py:106-109."""
    op = _hip.OP_INPUT

    def __init__(self, shape, name="input"):
        super(input_layer, self).__init__([])
        self.name = name
        self.out = Symbol(self, shape[1], shape[2], shape[3])

    def fill_desc(self, d):
        d.h, d.w, d.c = self.out.hwc


class upsample(_Layer):
    """Nearest-neighbour x stride.  Reference: net/layers.py:112-116."""
    op = _hip.OP_UPSAMPLE

    def __init__(self, prev, stride):
        super(upsample, self).__init__([prev])
        self.stride = int(stride)
        h, w, c = prev.hwc
        self.out = Symbol(self, h * self.stride, w * self.stride, c)

    def fill_desc(self, d):
        d.stride = self.stride


class yolo_layer(_Layer):
    """Head view [B, h*w*b, 5+C]; anchors (pixels) -> grid units.  Reference: net/layers.py:126-134."""
    op = _hip.OP_YOLO

    def __init__(self, prev, sub_anchors, no_c, input_shape):
        super(yolo_layer, self).__init__([prev])
        self.h, self.w, ch = prev.hwc
        stride = (input_shape[0] / self.h, input_shape[1] / self.w)
        self.anchors = [(a[0] / stride[0], a[1] / stride[1]) for a in sub_anchors]    # (w, h)
        self.b = len(self.anchors)
        self.no_c = int(no_c)
        self.out = Symbol(self, self.h, self.w, ch)
        self.rows = self.h * self.w * self.b

    def fill_desc(self, d):
        d.n_anchors = self.b
        for i, (aw, ah) in enumerate(self.anchors):
            d.anchors[2 * i], d.anchors[2 * i + 1] = float(aw), float(ah)


class detection_layer(_Layer):
    """Concat of the yolo heads on the row axis, coarse -> fine.  Reference: net/layers.py:119-123."""
    op = _hip.OP_DETECTION

    def __init__(self, yolos):
        super(detection_layer, self).__init__([y.out for y in yolos])
        self.yolos = list(yolos)
        self.out = Symbol(self, sum(y.rows for y in yolos), 1, 5 + yolos[0].no_c)
