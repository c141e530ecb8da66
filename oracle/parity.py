"""Oracle (test infrastructure): the parity gate reported next to every throughput number.

BASELINE.json's metric is "images/sec ... + post-NMS box-set match vs CPU ref"; the CPU reference of the
reference's per-batch body (net/yolo.py:83-86: sess.run + find_bounding_boxes) is this package's fp32
pipeline forward_ref.forward -> decode_ref.find_bounding_boxes_{v2,v3}.  `check()` compares a set of HIP
results with it:

  * max |logit - oracle logit|                      (north_star: 1e-4 on the fp32 path; reported for fp16)
  * post-NMS box sets: same count, order, class, and the same cell (coordinates within `coord_tol`)
  * the MARGIN RULE of SURVEY 7.3 #3: thresholds are discontinuous (`p < thr`, net/v2.py:107;
    `iou >= iou_thr`, net/base.py:204), so a logit error e can flip a row whose score sits within
    dp(e) of the threshold, or a suppression whose IoU sits within diou(e) of the IoU threshold.
    The oracle's own margins -- min |p - thr| over ALL rows, min |IoU - iou_thr| over every comparison the
    greedy pass makes -- are measured; identity is REQUIRED when both exceed the error propagated from the
    measured logit error, otherwise the differing boxes must be explained by a borderline row or IoU.

Only tests/, __graft_entry__.smoke() and bench.py (outside its timed region) import this.
"""
import numpy as np

from . import decode_ref


def _scores(logits, version, num_classes, anchors_per_cell=None):
    """Score p of every row the way the reference computes it (v2: sigmoid(obj) * max softmax, v3: sigmoid(obj))."""
    with np.errstate(over="ignore"):
        if version == 3:
            return decode_ref.sigmoid(np.asarray(logits, np.float32)[..., 4]).reshape(logits.shape[0], -1)
        x = np.asarray(logits, np.float32)
        x = x.reshape(x.shape[0], -1, 5 + num_classes)
        cls = x[..., 5:]
        e = np.exp(cls - cls.max(axis=-1, keepdims=True))
        return decode_ref.sigmoid(x[..., 4]) * (e.max(axis=-1) / e.sum(axis=-1))


def _iou_margin(cands, iou_threshold):
    """Greedy NMS as net/base.py:195-209 runs it, recording min |IoU - thr| over the comparisons it makes."""
    margin = np.inf
    if not cands:
        return margin
    boxes = sorted(cands, key=lambda b: b.prob, reverse=True)
    kept = [boxes[0]]
    for b in boxes[1:]:
        drop = False
        for k in kept:
            v = float(decode_ref.iou_score(k, b))
            margin = min(margin, abs(v - iou_threshold))
            if v >= iou_threshold:
                drop = True
                break
        if not drop:
            kept.append(b)
    return margin


def _same_box(g, w, coord_tol):
    if int(g[4]) != int(w[4]):
        return False
    return (abs(g[0] - w[0]) <= coord_tol and abs(g[1] - w[1]) <= coord_tol and
            abs(g[2] - w[2]) <= coord_tol * max(1.0, 10 * abs(w[2])) and abs(g[3] - w[3]) <= coord_tol * max(1.0, 10 * abs(w[3])))


def check(ref_logits, got_logits, got_boxes, version, threshold, iou_threshold, scales=None, anchors=None, num_classes=80,
          coord_tol=None):
    """ref_logits: oracle fp32 logits; got_logits: the HIP path's logits of the same images (or None);
    got_boxes: per image [(x, y, w, h, class_idx, prob)] from the HIP detect.  v3 needs `scales`
    (decode_ref.v3_scales), v2 needs `anchors`.  Returns a JSON-able dict (see module docstring)."""
    ref_logits = np.asarray(ref_logits, np.float32)
    n = ref_logits.shape[0]
    if version == 3:
        want = decode_ref.find_bounding_boxes_v3(ref_logits, threshold, iou_threshold, scales)
        pre = decode_ref.find_bounding_boxes_v3(ref_logits, threshold, iou_threshold, scales, nms=False)
    else:
        want = decode_ref.find_bounding_boxes_v2(ref_logits, threshold, iou_threshold, anchors, num_classes)
        pre = decode_ref.find_bounding_boxes_v2(ref_logits, threshold, iou_threshold, anchors, num_classes, nms=False)
    err = None
    if got_logits is not None:
        err = float(np.max(np.abs(np.asarray(got_logits, np.float64) - ref_logits.astype(np.float64))))
    e = err if err is not None else 0.0
    # propagated error bounds: d sigmoid <= e/4; v2 score = sigmoid * softmax-max, |d| <= e/4 + e/2 < e;
    # box centre moves <= e/4 of a cell, box size by a factor exp(+-e): IoU of two boxes moves by at most ~4e for e << 1
    dp = 0.25 * e if version == 3 else e
    diou = 4.0 * e
    p = _scores(ref_logits, version, num_classes)
    prob_margin = float(np.min(np.abs(p.astype(np.float64) - np.float64(np.float32(threshold)))))
    iou_margin = float(min(_iou_margin(c, iou_threshold) for c in pre)) if n else np.inf
    identity_required = bool(prob_margin > dp and iou_margin > diou)
    if coord_tol is None:       # a box is "the same" when class and cell agree; fp32: float rounding, fp16: a few 1e-3 of the image
        coord_tol = max(2e-5, 2.0 * e)
    matched = unmatched = 0
    identical = True
    for i in range(n):
        w_i = [b.astuple() for b in want[i]]
        g_i = [tuple(b) for b in got_boxes[i]]
        same_order = len(w_i) == len(g_i) and all(_same_box(g, w, coord_tol) for g, w in zip(g_i, w_i))
        if same_order:
            matched += len(w_i)
            continue
        identical = False
        used = [False] * len(g_i)
        for w in w_i:                                   # set comparison for the report
            hit = next((k for k, g in enumerate(g_i) if not used[k] and _same_box(g, w, coord_tol)), None)
            if hit is None:
                unmatched += 1
            else:
                used[hit] = True
                matched += 1
        unmatched += used.count(False)
    return {"images_checked": int(n), "max_abs_logit_err": err, "box_set_match": bool(identical),
            "boxes_ref": int(sum(len(b) for b in want)), "boxes_hip": int(sum(len(b) for b in got_boxes)),
            "boxes_matched": int(matched), "boxes_unmatched": int(unmatched),
            "prob_margin": prob_margin, "iou_margin": (None if not np.isfinite(iou_margin) else iou_margin),
            "prob_flip_band": dp, "iou_flip_band": diou, "identity_required": identity_required,
            "threshold": float(threshold), "iou_threshold": float(iou_threshold),
            "reference": "oracle fp32 pipeline (forward_ref + decode_ref), restatement of net/yolo.py:83-86"}


def assert_ok(rep, min_matched_frac=0.9):
    """The gate: identity where the margins demand it; elsewhere at most borderline boxes may differ."""
    if rep["identity_required"]:
        assert rep["box_set_match"], "box sets differ although the margins exceed the logit error: %r" % (rep,)
    total = max(1, rep["boxes_ref"])
    assert rep["boxes_matched"] >= min_matched_frac * total - 1, "too few boxes agree with the fp32 reference: %r" % (rep,)
