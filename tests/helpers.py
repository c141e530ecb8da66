"""Shared test helpers: bridge between the product's layer objects and the oracle's tuples."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import tensorflow_yolo_amd  # noqa: E402,F401  (shim -> tensorflow-yolo_amd/)
from tensorflow_yolo_amd.net import engine, layers as PL, synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def to_oracle(net):
    """Product layer objects -> oracle/topology tuples (same indices)."""
    engine.number_layers(net)
    out = []
    for l in net:
        src = [s.index for s in l.inputs]
        if isinstance(l, PL.input_layer):
            out.append(("input",) + tuple(l.out.hwc))
        elif isinstance(l, PL.conv2d_bn_act):
            out.append(("conv", src[0], l.filters, l.ksize, l.stride, l.batch_norm, l.activation))
        elif isinstance(l, PL.max_pool2d):
            out.append(("maxpool", src[0], l.ksize, l.stride))
        elif isinstance(l, PL.route):
            out.append(("route", src))
        elif isinstance(l, PL.reorg):
            out.append(("reorg", src[0], l.stride))
        elif isinstance(l, PL.shortcut):
            out.append(("shortcut", src[0], src[1]))
        elif isinstance(l, PL.upsample):
            out.append(("upsample", src[0], l.stride))
        elif isinstance(l, PL.yolo_layer):
            out.append(("yolo", src[0], [tuple(a) for a in l.anchors]))
        elif isinstance(l, PL.detection_layer):
            out.append(("detection", src))
        else:
            raise TypeError(l)
    return out


class Graph(list):
    """A free-form layer list for operator tests."""
    engine = None
    darknet_weights = None


def new_graph(h, w, c):
    PL.conv2d_bn_act.reset()
    g = Graph()
    g.append(PL.input_layer([None, h, w, c]))
    return g


def run_hip(net, weights, x, dtype, keep_all=False, max_batch=None, force_tile=None, **engine_kw):
    eng = engine.HipNetwork(net, dtype=dtype, max_batch=max_batch or x.shape[0], keep_all=keep_all, force_tile=force_tile, **engine_kw)
    eng.load_weights(weights)
    out = eng.forward(x).cpu().numpy()
    return out, eng


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-6, float(np.max(np.abs(b)))))


def match_boxes(got, want, atol_xy=2e-5, atol_p=2e-6):
    """got/want: lists of (x,y,w,h,cls,prob) in output order.  Exact order, class and count; coordinates/prob to float32
    rounding.  Order: up to boxes whose scores tie within 4 float32 ulps -- the reference's score is float32 arithmetic on
    NumPy's float32 exp (net/base.py:171-172), a few ulp accurate and not the same function as the GPU's expf, so two rows
    whose exact scores are closer than that may be sorted either way (oracle/parity.py: kTieUlps)."""
    assert len(got) == len(want), "count %d vs %d" % (len(got), len(want))

    def same(g, w):
        return (int(g[4]) == int(w[4]) and abs(g[5] - w[5]) <= atol_p and
                all(abs(g[i] - w[i]) <= atol_xy * max(1.0, abs(w[i])) for i in range(4)))

    tie = 4 * 2.0 ** -24
    used = [False] * len(want)
    for k, g in enumerate(got):
        if not used[k] and same(g, want[k]):
            used[k] = True
            continue
        lo = k
        while lo > 0 and abs(want[lo - 1][5] - want[lo][5]) <= tie:
            lo -= 1
        hi = k
        while hi + 1 < len(want) and abs(want[hi + 1][5] - want[hi][5]) <= tie:
            hi += 1
        j = next((j for j in range(lo, hi + 1) if not used[j] and same(g, want[j])), None)
        w = want[k]
        assert j is not None, "box %d: %r vs %r (no box of the tied-score run %d..%d matches)" % (k, tuple(g), tuple(w), lo, hi)
        used[j] = True
