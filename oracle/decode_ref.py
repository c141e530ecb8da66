"""Oracle (test infrastructure): NumPy restatement of the reference's head decode + NMS.

PINNED against the reference's own functions (imported in the build container by
oracle/gen_golden.py) through tests/golden/decode_*.npz.

Follows net/v2.py:83-119, net/v3.py:109-151, net/base.py:171-209,257-272.
dtypes mirror what the reference produces under NumPy 2.x: x, y, prob are float32;
w, h are float64 (the anchors reach the arithmetic as np.float64: net/yolo.py:47 for
v2, net/layers.py:131 for v3); class_idx is an integer.

For speed the scan is pre-filtered with a vectorised pass at a slightly relaxed
threshold; every surviving cell is then recomputed with the reference's scalar
expressions, and the reference's own `p < threshold` test decides.  `full_scan=True`
walks EVERY cell with the scalar expressions instead, the way the reference's triple
loop does (net/v2.py:98-100, net/v3.py:113-115): same result, the reference's cost --
bench.py times it as the CPU decode baseline.
"""
import numpy as np

_RELAX = 1e-5


class Box(object):
    """Same fields as net/base.py:257-272 BoundingBox (cx, cy unused on this path)."""
    __slots__ = ("x", "y", "w", "h", "class_idx", "prob", "scan")

    def __init__(self, x, y, w, h, class_idx, prob, scan=-1):
        self.x, self.y, self.w, self.h = x, y, w, h
        self.class_idx, self.prob, self.scan = class_idx, prob, scan

    def top_left(self):                      # base.py:267-269 with h=w=1.
        return (self.x - self.w / 2.) * 1., (self.y - self.h / 2.) * 1.

    def bottom_right(self):                  # base.py:271-272
        return (self.x + self.w / 2.) * 1., (self.y + self.h / 2.) * 1.

    def astuple(self):
        return (float(self.x), float(self.y), float(self.w), float(self.h), int(self.class_idx), float(self.prob))


def sigmoid(x):                              # base.py:171-172
    return 1. / (1. + np.exp(-x))


def softmax(x):                              # base.py:175-177
    e_x = np.exp(x - np.max(x))
    return e_x / e_x.sum()


def iou_score(b1, b2):                       # base.py:180-192
    b1_min, b1_max = b1.top_left(), b1.bottom_right()
    a1 = b1.w * b1.h
    b2_min, b2_max = b2.top_left(), b2.bottom_right()
    a2 = b2.w * b2.h
    imin = np.maximum(b1_min, b2_min)
    imax = np.minimum(b1_max, b2_max)
    iwh = np.maximum(imax - imin, 0)
    inter = iwh[0] * iwh[1]
    union = np.maximum(a1 + a2 - inter, 1e-8)
    return inter / union


def non_maximum_suppression(boxes, iou_threshold):   # base.py:195-209
    if len(boxes) == 0:
        return []
    boxes = sorted(boxes, key=lambda b: b.prob, reverse=True)    # stable, like list.sort
    kept = [boxes[0]]
    for b in boxes[1:]:
        if not any(iou_score(k, b) >= iou_threshold for k in kept):
            kept.append(b)
    return kept


def _decode_cells(out, anchors, threshold, version, scan_base=0, full_scan=False):
    """out: [h, w, A, 5+C] float32.  anchors: sequence of (aw, ah) np.float64 in grid units.
    Returns boxes in the reference scan order (cy, cw, anchor)."""
    with np.errstate(over="ignore"):
        h, w, A = out.shape[0:3]
        if full_scan:
            cand = ((cy, cw, b) for cy in range(h) for cw in range(w) for b in range(A))     # v2.py:98-100 / v3.py:113-115
        else:
            po_all = sigmoid(out[..., 4])
            if version == 2:
                cls = out[..., 5:]
                e = np.exp(cls - cls.max(axis=-1, keepdims=True))
                p_all = po_all * (e.max(axis=-1) / e.sum(axis=-1))
            else:
                p_all = po_all
            cand = np.argwhere(p_all >= threshold - _RELAX)         # row-major == scan order
        boxes = []
        for cy, cw, b in cand:
            prob_obj = sigmoid(out[cy, cw, b, 4])
            if version == 2:                                    # v2.py:102-106
                prob_classes = softmax(out[cy, cw, b, 5:])
                class_idx = np.argmax(prob_classes)
                p = prob_obj * prob_classes[class_idx]
            else:                                               # v3.py:119-123
                prob_classes = sigmoid(out[cy, cw, b, 5:])
                class_idx = np.argmax(prob_classes)
                p = prob_obj
            if p < threshold:
                continue
            c = out[cy, cw, b, 0:4]
            boxes.append(Box(
                x=(sigmoid(c[0]) + int(cw)) / w,                # v2.py:112 / v3.py:129
                y=(sigmoid(c[1]) + int(cy)) / h,
                w=(anchors[b][0] * np.exp(c[2])) / w,
                h=(anchors[b][1] * np.exp(c[3])) / h,
                class_idx=class_idx, prob=p,
                scan=scan_base + (int(cy) * w + int(cw)) * A + int(b)))
        return boxes


def find_bounding_boxes_v2(net_out, threshold, iou_threshold, anchors, num_classes, nms=True, full_scan=False):
    """net/v2.py:83-90.  net_out: [B,h,w,A*(5+C)] float32; anchors [A,2] in grid units."""
    anchors = np.reshape(np.asarray(anchors, np.float64), [-1, 2])
    net_out = np.asarray(net_out, np.float32)
    net_out = np.reshape(net_out, [-1, net_out.shape[1], net_out.shape[2], len(anchors), 5 + num_classes])
    res = []
    for out in net_out:
        boxes = _decode_cells(out, anchors, threshold, 2, full_scan=full_scan)
        res.append(non_maximum_suppression(boxes, iou_threshold) if nms else boxes)
    return res


def v3_scales(anchors_px, input_hw, strides=(32, 16, 8)):
    """net/v3.py:11 + net/layers.py:126-134: per head (coarse first) -> (h, w, anchors in grid units)."""
    anc = np.reshape(np.asarray(anchors_px), [3, -1, 2])[::-1, :, :]
    out = []
    for i, s in enumerate(strides):
        h, w = input_hw[0] // s, input_hw[1] // s
        st = (input_hw[0] / h, input_hw[1] / w)
        out.append((h, w, [(a[0] / st[0], a[1] / st[1]) for a in anc[i]]))
    return out


def find_bounding_boxes_v3(net_out, threshold, iou_threshold, scales, nms=True, full_scan=False):
    """net/v3.py:140-151.  net_out: [B, sum(h*w*b), 5+C] float32; scales from v3_scales()."""
    net_out = np.asarray(net_out, np.float32)
    res = []
    for out in net_out:
        idx = 0
        boxes = []
        for (h, w, anc) in scales:
            dim = h * w * len(anc)
            l_out = np.reshape(out[idx:idx + dim, ...], [h, w, len(anc), -1])
            boxes.extend(_decode_cells(l_out, anc, threshold, 3, scan_base=idx, full_scan=full_scan))
            idx += dim
        res.append(non_maximum_suppression(boxes, iou_threshold) if nms else boxes)
    return res


def boxes_to_array(boxes):
    """[(x, y, w, h, class_idx, prob)] as float64 [N,6] (for fixtures / comparisons)."""
    return np.array([b.astuple() for b in boxes], dtype=np.float64).reshape(-1, 6)


def non_maximum_suppression_per_class(boxes, iou_threshold):
    """north_star's "per-class NMS" (opt-in; NOT what the reference does, net/base.py:195-209 never looks at class_idx):
    the same stable sort and greedy pass, but only a kept box of the SAME class suppresses.  Output order = one list by
    descending prob, like the reference's."""
    if len(boxes) == 0:
        return []
    boxes = sorted(boxes, key=lambda b: b.prob, reverse=True)
    kept = [boxes[0]]
    for b in boxes[1:]:
        if not any(int(k.class_idx) == int(b.class_idx) and iou_score(k, b) >= iou_threshold for k in kept):
            kept.append(b)
    return kept
