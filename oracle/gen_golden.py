#!/usr/bin/env python3
"""Oracle (test infrastructure): generate tests/golden/*.npz by running the REFERENCE's own
NumPy decode + NMS (net/v2.py:83-119, net/v3.py:109-151, net/base.py:171-209) on the seeded
heads of oracle/cases.py.

Runs ONLY in the build container (needs /root/reference).  The reference's module-level
imports of tensorflow / cv2 / imgaug (net/base.py:4-9, absent here) are satisfied by inert
placeholder modules; the functions called below touch NumPy only.  Nothing from the
reference is copied: the fixtures hold inputs-by-seed and the reference's OUTPUTS.

    python oracle/gen_golden.py            # rewrites tests/golden/decode_*.npz, nms_cases.npz
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("YOLO_REFERENCE_DIR", "/root/reference")
sys.path.insert(0, ROOT)
from oracle import cases, decode_ref  # noqa: E402


def import_reference():
    sys.dont_write_bytecode = True

    class _Inert(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Inert(self.__name__ + "." + k)

        def __call__(self, *a, **kw):
            return _Inert("call")

    for name in ("tensorflow", "cv2", "imgaug", "imgaug.augmenters"):
        sys.modules.setdefault(name, _Inert(name))
    sys.path.insert(0, REF)
    from net import base, v2, v3
    return base, v2, v3


def as_array(boxes):
    return np.array([(float(b.x), float(b.y), float(b.w), float(b.h), int(b.class_idx), float(b.prob))
                     for b in boxes], dtype=np.float64).reshape(-1, 6)


class _Obj(object):
    pass


def fake_v3_net(c):
    """What find_bounding_boxes reads from `net`: net[-1].yolos[i].{h,w,b,anchors}, built the
    way net/v3.py:11 and net/layers.py:126-134 build them."""
    anchors = np.reshape(c["anchors"], [-1, 2])                     # net/yolo.py:47
    anc = np.reshape(anchors, [3, -1, 2])[::-1, :, :]               # net/v3.py:11
    det = _Obj()
    det.yolos = []
    for i, s in enumerate((32, 16, 8)):
        y = _Obj()
        y.h = y.w = c["input"] // s
        stride = (c["input"] / y.h, c["input"] / y.w)
        y.anchors = [(a[0] / stride[0], a[1] / stride[1]) for a in anc[i]]
        y.b = len(y.anchors)
        det.yolos.append(y)
    return [det]


def main():
    base, v2, v3 = import_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    meta = {"numpy": np.__version__, "generator": "oracle/gen_golden.py", "source": "reference functions"}
    np.seterr(over="ignore")
    for name, c in cases.CASES.items():
        head = cases.make_head(name)
        names = ["c%d" % i for i in range(c["classes"])]
        arrays = {}
        if c["version"] == 2:
            anchors = np.reshape(c["anchors"], [-1, 2])             # net/yolo.py:47
            post = v2.find_bounding_boxes(head, None, c["threshold"], c["iou"], anchors, names)
            h5 = np.reshape(head, [-1, head.shape[1], head.shape[2], len(anchors), 5 + c["classes"]])
            pre = [v2._find_bounding_boxes(o, anchors, c["threshold"]) for o in h5]
        else:
            net = fake_v3_net(c)
            anchors = np.reshape(c["anchors"], [-1, 2])
            post = v3.find_bounding_boxes(head, net, c["threshold"], c["iou"], anchors, names)
            pre = []
            for o in head:
                idx, bxs = 0, []
                for l in net[-1].yolos:
                    dim = l.h * l.w * l.b
                    bxs.extend(v3._find_bounding_boxes(np.reshape(o[idx:idx + dim], [l.h, l.w, l.b, -1]),
                                                       l.anchors, c["threshold"]))
                    idx += dim
                pre.append(bxs)
        for i in range(c["batch"]):
            arrays["post%d" % i] = as_array(post[i])
            arrays["pre%d" % i] = as_array(pre[i])
        arrays["meta"] = np.array(json.dumps(dict(meta, case=name, **{k: v for k, v in c.items()})))
        np.savez_compressed(os.path.join(out_dir, "decode_%s.npz" % name), **arrays)
        print(name, [len(p) for p in pre], "->", [len(p) for p in post])

    arrays = {}
    for name, (boxes, thr) in cases.NMS_CASES.items():
        bb = [base.BoundingBox(x=np.float32(b[0]), y=np.float32(b[1]), w=np.float64(b[2]), h=np.float64(b[3]),
                               class_idx=b[4], prob=np.float32(b[5])) for b in boxes]
        kept = base.non_maximum_suppression(bb, thr)
        arrays[name] = as_array(kept)
        print("nms", name, len(boxes), "->", len(kept))
    arrays["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out_dir, "nms_cases.npz"), **arrays)

    # self-check: the restatement must agree with what was just written
    for name, c in cases.CASES.items():
        head = cases.make_head(name)
        if c["version"] == 2:
            mine = decode_ref.find_bounding_boxes_v2(head, c["threshold"], c["iou"], c["anchors"], c["classes"])
        else:
            sc = decode_ref.v3_scales(c["anchors"], (c["input"], c["input"]))
            mine = decode_ref.find_bounding_boxes_v3(head, c["threshold"], c["iou"], sc)
        g = np.load(os.path.join(out_dir, "decode_%s.npz" % name))
        for i in range(c["batch"]):
            a, b = decode_ref.boxes_to_array(mine[i]), g["post%d" % i]
            assert a.shape == b.shape, (name, i, a.shape, b.shape)
            assert np.array_equal(a[:, 4], b[:, 4]) and np.allclose(a, b, rtol=0, atol=1e-6), (name, i)
    print("restatement agrees with the reference on all cases")


if __name__ == "__main__":
    main()
