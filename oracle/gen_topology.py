#!/usr/bin/env python3
"""Oracle (test infrastructure): pin the network TOPOLOGY and the Darknet WEIGHT ORDER to the reference's own
builders.  Runs ONLY in the build container (needs /root/reference).

The reference's graph builders (net/v2.py:11-60 create_full_network, net/v3.py:9-94 create_network, over the layer
classes of net/layers.py:17-134) are executed as they are; the absent `tensorflow` module is replaced by a
RECORDING module whose ops compute nothing and return shape-carrying tensor records (op name, attributes, input
records).  What the builders did is then read back from the returned layer list:

    per layer: class name, the chain of recorded TF ops behind `.out`, the source layers (found by walking the op
    records back to other layers' `.out` objects BY IDENTITY), the output shape, `.variable_names` (the Darknet stream
    order, net/base.py:26-46), and for yolo layers h, w, b and the stride-scaled anchors (net/layers.py:126-134)

and written to tests/golden/topology_{v2_416,v3_416,v3_608}.json -- data, not source.  tests/test_oracle_golden.py
asserts   product layer list == oracle/topology.py == this fixture.   Nothing of the reference is copied.

    python oracle/gen_topology.py          # rewrites the three fixtures (byte-identical when nothing changed)
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("YOLO_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import cases  # noqa: E402


class Rec(object):
    """A recorded tensor: what the stand-in ops return."""

    def __init__(self, op, inputs, shape, **attrs):
        self.op, self.inputs, self.shape, self.attrs = op, list(inputs), list(shape), attrs

    def get_shape(self):
        return self

    def as_list(self):
        return list(self.shape)

    def __add__(self, other):               # net/layers.py:102  prev + shortcut_out
        assert self.shape == other.shape, (self.shape, other.shape)
        return Rec("add", [self, other], self.shape)


def _conv_out(n, k, s, padding):
    if n is None:
        return None
    return -(-n // s) if padding == "SAME" else (n - k) // s + 1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def recording_tensorflow():
    tf = types.ModuleType("tensorflow")
    tf.float32 = "float32"
    tf.reset_default_graph = lambda: None

    class _Scope(object):
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    tf.variable_scope = _Scope
    tf.placeholder = lambda dtype, shape, name=None: Rec("placeholder", [], shape, dtype=dtype, name=name)

    def pad(t, paddings, mode="CONSTANT"):
        shape = [None if d is None else d + p[0] + p[1] for d, p in zip(t.shape, paddings)]
        return Rec("pad", [t], shape, paddings=[list(p) for p in paddings], mode=mode)

    tf.pad = pad

    def concat(values, axis):
        shape = list(values[0].shape)
        shape[axis] = sum(v.shape[axis] for v in values)
        for v in values:
            assert [d for i, d in enumerate(v.shape) if i != axis] == [d for i, d in enumerate(shape) if i != axis]
        return Rec("concat", values, shape, axis=axis)

    tf.concat = concat
    tf.identity = lambda t, name=None: Rec("identity", [t], t.shape, name=name)

    def reshape(t, shape):
        return Rec("reshape", [t], [None if d == -1 else d for d in shape], new_shape=list(shape))

    tf.reshape = reshape

    def extract_image_patches(t, ksizes, strides, rates, padding):
        n, h, w, c = t.shape
        return Rec("extract_image_patches", [t], [n, _conv_out(h, ksizes[1], strides[1], padding), _conv_out(w, ksizes[2], strides[2], padding),
                                                  c * ksizes[1] * ksizes[2]], ksizes=list(ksizes), strides=list(strides), rates=list(rates), padding=padding)

    tf.extract_image_patches = extract_image_patches

    layers = types.ModuleType("tensorflow.layers")

    def conv2d(inputs, filters, kernel_size, padding, strides, use_bias, name):
        n, h, w, c = inputs.shape
        k, s = _pair(kernel_size), _pair(strides)
        return Rec("conv2d", [inputs], [n, _conv_out(h, k[0], s[0], padding), _conv_out(w, k[1], s[1], padding), filters],
                   filters=filters, kernel_size=list(k), padding=padding, strides=list(s), use_bias=bool(use_bias), name=name, in_channels=c)

    def batch_normalization(t, training, momentum, epsilon, name):
        return Rec("batch_normalization", [t], t.shape, training=bool(training), momentum=momentum, epsilon=epsilon, name=name)

    def max_pooling2d(inputs, pool_size, strides, padding):
        n, h, w, c = inputs.shape
        k, s = _pair(pool_size), _pair(strides)
        return Rec("max_pooling2d", [inputs], [n, _conv_out(h, k[0], s[0], padding), _conv_out(w, k[1], s[1], padding), c],
                   pool_size=list(k), strides=list(s), padding=padding)

    layers.conv2d, layers.batch_normalization, layers.max_pooling2d = conv2d, batch_normalization, max_pooling2d
    tf.layers = layers
    nn = types.ModuleType("tensorflow.nn")
    nn.leaky_relu = lambda t, alpha, name=None: Rec("leaky_relu", [t], t.shape, alpha=alpha, name=name)
    tf.nn = nn
    image = types.ModuleType("tensorflow.image")
    image.resize_nearest_neighbor = lambda t, size: Rec("resize_nearest_neighbor", [t], [t.shape[0], size[0], size[1], t.shape[3]], size=list(size))
    tf.image = image
    return tf


def import_reference():
    sys.dont_write_bytecode = True

    class _Inert(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Inert(self.__name__ + "." + k)

        def __call__(self, *a, **kw):
            return _Inert("call")

    sys.modules["tensorflow"] = recording_tensorflow()
    for name in ("cv2", "imgaug", "imgaug.augmenters"):         # net/base.py:4-9: never touched by the builders
        sys.modules.setdefault(name, _Inert(name))
    sys.path.insert(0, REF)
    from net import v2, v3
    return v2, v3


def dump(layers):
    """The reference's layer list -> JSON-able records (see module docstring)."""
    owner = {id(l.out): i for i, l in enumerate(layers)}
    # v2/v3 rename the last layer's `.out` through tf.identity (net/v2.py:59, net/v3.py:93): look through it
    out = []
    for i, l in enumerate(layers):
        ops, srcs = [], []

        def walk(t, top=False):
            if not top and id(t) in owner and owner[id(t)] < i:
                srcs.append(owner[id(t)])
                return
            ops.append(dict({"op": t.op}, **{k: (v if not isinstance(v, tuple) else list(v)) for k, v in t.attrs.items()}))
            for u in t.inputs:
                walk(u)

        walk(l.out, top=True)
        rec = {"index": i, "class": type(l).__name__, "src": srcs, "ops": ops[::-1], "shape": l.out.shape,
               "variable_names": list(l.variable_names)}
        if type(l).__name__ == "yolo_layer":
            rec.update(h=l.h, w=l.w, b=l.b, anchors=[[float(a[0]), float(a[1])] for a in l.anchors])
        if type(l).__name__ == "detection_layer":
            rec["yolos"] = [next(k for k, m in enumerate(layers) if m is y) for y in l.yolos]
            rec["src"] = rec["yolos"]
        out.append(rec)
    return out


CONFIGS = {
    "v2_416": ("v2", cases.COCO_V2_ANCHORS, 80, (416, 416, 3)),
    "v3_416": ("v3", cases.COCO_V3_ANCHORS, 80, (416, 416, 3)),
    "v3_608": ("v3", cases.COCO_V3_ANCHORS, 80, (608, 608, 3)),
}


def main():
    v2, v3 = import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (kind, anchors, ncls, shape) in CONFIGS.items():
        a = np.reshape(anchors, [-1, 2])                            # net/yolo.py:47
        names = ["c%d" % i for i in range(ncls)]
        layers = (v2.create_full_network if kind == "v2" else v3.create_network)(a, names, False, input_shape=shape)
        rec = {"generator": "oracle/gen_topology.py", "source": "reference builders under a recording tensorflow module",
               "net": kind, "input_shape": list(shape), "num_classes": ncls, "anchors": [float(v) for v in np.ravel(anchors)],
               "layers": dump(layers)}
        path = os.path.join(out_dir, "topology_%s.json" % name)
        with open(path, "w") as f:
            json.dump(rec, f, indent=0, sort_keys=True)
            f.write("\n")
        print(name, len(layers), "layers ->", path)


if __name__ == "__main__":
    main()
