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
    """got/want: lists of (x,y,w,h,cls,prob) in output order.  Exact order, class and count;
    coordinates/prob to float32 rounding."""
    assert len(got) == len(want), "count %d vs %d" % (len(got), len(want))
    for k, (g, w) in enumerate(zip(got, want)):
        assert int(g[4]) == int(w[4]), "box %d: class %s vs %s" % (k, g[4], w[4])
        assert abs(g[5] - w[5]) <= atol_p, "box %d: prob %r vs %r" % (k, g[5], w[5])
        for i in range(4):
            assert abs(g[i] - w[i]) <= atol_xy * max(1.0, abs(w[i])), "box %d field %d: %r vs %r" % (k, i, g[i], w[i])
