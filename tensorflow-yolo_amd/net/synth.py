"""Seeded synthetic ("random-init") weights in the Darknet stream format the reference reads
(reference net/base.py:26-46; per conv: beta, gamma, moving_mean, moving_variance, kernel
[out][in][kh][kw] for BN layers, bias + kernel otherwise).

No pretrained weights exist offline (the reference links external URLs), so benchmarks and
parity tests run on these.  The distributions are chosen so that the signal stays O(1) through
75 layers and 23 residual adds, as in a trained network (otherwise fp16 storage would overflow):
  kernel            ~ N(0, 2 / (1.01 * k*k*Cin))     second-moment preserving under leaky(0.1)
  moving_variance   ~ U(0.5, 1.5)
  gamma             = sqrt(moving_variance) * U(0.8, 1.2) * gain
                      gain = 0.25 for the conv that feeds a shortcut (damped residual branch), else 1
  beta, moving_mean ~ N(0, 0.1)
  head conv (no BN)   bias ~ N(0, 1); the objectness logit of every anchor is shifted by `obj_bias`
                      (the "prior probability" initialisation detectors use), so that a realistic
                      handful of cells -- not half of them -- pass the 0.5 score threshold.
"""
import numpy as np

from .layers import conv2d_bn_act, shortcut


# per network: (head_gain, fraction of cells per anchor that pass sigmoid(obj) >= 0.5 after
# calibrate_model) -- a realistic few dozen candidates per image (see DESIGN.md "Synthetic weights")
HEAD_DEFAULTS = {"v3": (0.6, 0.004), "v2": (4.0, 0.08), "v2-tiny": (2.0, 0.15)}


def calibrate_model(model, x_calib, fraction):
    """Run the model's own (HIP) forward on a calibration batch, re-centre the objectness biases
    (calibrate_objectness) and reload the weights.  Returns the new Darknet stream."""
    logits = model.forward(x_calib)
    w = calibrate_objectness(model.net.darknet_weights, model.net, logits, len(model.class_names), fraction)
    model.net.darknet_weights = w
    model.net.engine.load_weights(w)
    return w


def darknet_stream(net, seed=0, obj_bias=-5.0, num_classes=None, head_gain=1.0):
    """Flat float32 array for the conv layers of `net`, in layer-list order.
    head_gain scales the kernels of the linear head convs (logit spread)."""
    rng = np.random.RandomState(seed)
    damped = set()
    for l in net:
        if isinstance(l, shortcut):
            damped.add(id(l.inputs[0]))
    parts = []
    for l in net:
        if not isinstance(l, conv2d_bn_act):
            continue
        f, cin, k = l.filters, l.in_channels, l.ksize
        if l.batch_norm:
            var = rng.uniform(0.5, 1.5, f)
            gain = 0.25 if id(l) in damped else 1.0
            gamma = np.sqrt(var) * rng.uniform(0.8, 1.2, f) * gain
            parts.append((rng.randn(f) * 0.1).astype(np.float32))               # beta
            parts.append(gamma.astype(np.float32))                              # gamma
            parts.append((rng.randn(f) * 0.1).astype(np.float32))               # moving_mean
            parts.append(var.astype(np.float32))                                # moving_variance
        else:
            bias = rng.randn(f).astype(np.float32)
            if num_classes is not None and f % (5 + num_classes) == 0:
                # objectness logit of every anchor: the prior, with little spread between anchors, so the
                # number of cells that pass the threshold does not hinge on a handful of random draws
                bias[4::5 + num_classes] = np.float32(obj_bias) + 0.1 * bias[4::5 + num_classes]
            parts.append(bias)
        std = np.sqrt(2.0 / (1.01 * k * k * cin)) * (1.0 if l.batch_norm else head_gain)
        parts.append((rng.randn(f * cin * k * k) * std).astype(np.float32))
    return np.concatenate(parts)


def head_bias_offsets(net):
    """Offset (in the flat Darknet stream) of the bias vector of every non-BN (head) conv, in order."""
    out, pos = [], 0
    for l in net:
        if isinstance(l, conv2d_bn_act):
            if not l.batch_norm:
                out.append((pos, l.filters))
            pos += l.weight_count()
    return out


def calibrate_objectness(weights, net, logits, num_classes, fraction=0.004):
    """Data-dependent prior: shift each anchor's objectness bias so that `fraction` of its cells have
    sigmoid(obj) >= 0.5 on the calibration batch whose head `logits` (NumPy, reference layout, computed
    with `weights`) are given.  Random features are spatially homogeneous, so without this a fixed
    prior passes either none or nearly all cells of an anchor.  Returns a new weight stream."""
    w = np.array(weights, dtype=np.float32, copy=True)
    width = 5 + num_classes
    heads = head_bias_offsets(net)
    logits = np.asarray(logits, dtype=np.float32)
    if logits.ndim == 4:                                    # v2: [B, h, w, A*(5+C)], one head
        per_head = [logits.reshape(logits.shape[0], -1, logits.shape[3] // width, width)]
    else:                                                   # v3: [B, rows, 5+C], heads coarse -> fine
        per_head, r0 = [], 0
        for y in net[-1].yolos:
            per_head.append(logits[:, r0:r0 + y.rows].reshape(logits.shape[0], -1, y.b, width))
            r0 += y.rows
    for (pos, filters), t in zip(heads, per_head):
        for a in range(filters // width):
            q = np.quantile(t[:, :, a, 4].astype(np.float64), 1.0 - fraction)
            w[pos + a * width + 4] -= np.float32(q)
    return w


def synthetic_input(batch, h, w, c=3, seed=0):
    """uniform[0,1) float32 NHWC, the shape/range preprocess_image produces."""
    return np.random.RandomState(seed).random_sample((batch, h, w, c)).astype(np.float32)
