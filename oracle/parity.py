"""Oracle (test infrastructure): the parity gate reported next to every throughput number.

BASELINE.json's metric is "images/sec ... + post-NMS box-set match vs CPU ref"; the CPU reference of the
reference's per-batch body (net/yolo.py:83-86: sess.run + find_bounding_boxes) is this package's fp32
pipeline forward_ref.forward -> decode_ref.find_bounding_boxes_{v2,v3}.  `check()` compares a set of HIP
results with it:

  * max |logit - oracle logit|, and whether it stays within a bound the HIP path CANNOT influence (`logit_err_bound`):
    1e-4 absolute on the fp32 path (north_star), 1.5 x e_ref on the fp16 path, where e_ref = max |forward_ref(storage=
    "fp16") - forward_ref(fp32)| on the SAME images -- what fp16 storage alone does to these logits, computed by the oracle
  * post-NMS box sets: same count, order, class, cell and score
  * EVERY differing box must be EXPLAINED.  Thresholds are discontinuous (`p < thr`, net/v2.py:107, net/v3.py:124;
    `iou >= iou_thr`, net/base.py:204), so a logit error can flip a row whose score sits on the threshold, a suppression
    whose IoU sits on the IoU threshold, or the order of two boxes with (almost) equal scores -- and nothing else.  The
    width of each band comes from the logit errors OF THE ROWS INVOLVED, capped by the independent bound above: a forward
    defect that raises the error does not widen the bands that are meant to catch it (round 3 used the HIP path's own
    global maximum).  With e_s = min(the row's own |logit error| in the channels that enter the quantity, logit_err_bound):

      score of a row           dp(s)      = e_s / 4 (v3: sigmoid(obj)) or e_s (v2: sigmoid(obj) * max softmax)
      IoU of two boxes         diou(k, s) = 4 max(e_k, e_s) over the eight coordinate logits (centre moves <= e/4 of a cell,
                                            size by a factor exp(+-e))
      order of two rows        |p_k - p_s| <= dp(k) + dp(s)
      class of a row           the oracle's two class logits within 2 e_s (class channels)

    The gate works in two links:

      link 1  HIP boxes == the oracle's decode + NMS applied to the HIP path's OWN logits, exactly (order, class,
              score to float32 rounding; order up to scores that tie within 4 float32 ulps: `score_ties_reordered`, see kTieUlps).  Any difference is a decode / NMS defect of the HIP path: unexplained.
      link 2  the oracle's decode + NMS trace of the HIP logits against its trace of the oracle logits, row by row
              (rows are identified by their scan index).  A row whose candidate / survivor status differs is explained
              only by (a) its oracle score within dp of the threshold, (b) the IoU (oracle geometry) with the box that
              suppresses it in one trace within diou of the IoU threshold, (c) that suppressor being a flipped row that
              is itself explained, or (d) an order swap with that suppressor at scores within the two rows' bands.  A class
              change of a common survivor is explained only by the two class logits lying within 2 e_s in the oracle.

    `boxes_unexplained` counts everything else; assert_ok() demands 0 and `logit_err_within_bound`.  When the oracle's own
    margins (min |p - thr| over ALL rows, min |IoU - iou_thr| over every comparison its greedy pass makes) exceed the widest
    bands the bound allows, no flip is possible at all and identity is REQUIRED.  `prob_flip_band` / `iou_flip_band` report the
    widest band that actually explained something (0 when nothing had to be explained), `*_cap` what the bound would allow.

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


def nms_trace(cands, iou_threshold, per_class=False):
    """Greedy NMS exactly as net/base.py:195-209 runs it (stable sort by prob, first suppressor wins), recording for every
    candidate whether it survives and, if not, which survivor suppressed it at which IoU; plus min |IoU - thr| over the
    comparisons made.  per_class: the north_star opt-in (only same-class boxes suppress each other; not in the reference).
    Returns (order [boxes by descending prob], kept [bool], by [index into order of the suppressor or -1], iou [float], margin)."""
    margin = np.inf
    order = sorted(cands, key=lambda b: b.prob, reverse=True)
    kept, by, ious = [], [], []
    kept_idx = []
    for i, b in enumerate(order):
        sup, v_sup = -1, np.nan
        for k in kept_idx:
            if per_class and int(order[k].class_idx) != int(b.class_idx):
                continue
            v = float(decode_ref.iou_score(order[k], b))
            margin = min(margin, abs(v - iou_threshold))
            if v >= iou_threshold:
                sup, v_sup = k, v
                break
        kept.append(sup < 0)
        by.append(sup)
        ious.append(v_sup)
        if sup < 0:
            kept_idx.append(i)
    return order, kept, by, ious, margin


def _iou_margin(cands, iou_threshold):
    return nms_trace(cands, iou_threshold)[4]


def _same_box(g, w, coord_tol, prob_tol=None):
    if int(g[4]) != int(w[4]):
        return False
    if prob_tol is not None and abs(g[5] - w[5]) > prob_tol:
        return False
    return (abs(g[0] - w[0]) <= coord_tol and abs(g[1] - w[1]) <= coord_tol and
            abs(g[2] - w[2]) <= coord_tol * max(1.0, 10 * abs(w[2])) and abs(g[3] - w[3]) <= coord_tol * max(1.0, 10 * abs(w[3])))


# Scores that tie to float32 rounding.  The reference's score is float32 arithmetic on NumPy's float32 exp (net/base.py:171-172), which is
# accurate to a few ulp but not correctly rounded -- and not the same function on every host (NumPy dispatches SIMD variants by CPU) or on
# the GPU (expf).  Two rows whose exact scores differ by less than that can come out equal on one side (then the scan order decides,
# net/base.py:199: stable sort) and ordered the other way on the other.  Nothing in the reference pins that order; the gate accepts a
# permutation of output boxes inside a run of scores that lie within kTieUlps float32 ulps of each other -- and nothing else: the boxes
# themselves must be the same ones (measured: YOLOv3-608 b32 float32, 1 such pair among 2 781 boxes: scores 0.50947118 and 0.50947124).
kTieUlps = 4


def _match_with_ties(g_list, w_list, coord_tol, prob_tol):
    """HIP boxes g_list against reference boxes w_list (both in output order).  Returns (mismatches, tie_swaps): position k
    matches when g_list[k] is w_list[k], or w_list[j] for an unused j in the same run of tied scores (see kTieUlps)."""
    if len(g_list) != len(w_list):
        return abs(len(g_list) - len(w_list)) + sum(1 for g, w in zip(g_list, w_list) if not _same_box(g, w, coord_tol, prob_tol)), 0
    tie = kTieUlps * 2.0 ** -24             # scores are < 1: one float32 ulp is at most 2^-24 there
    group = [0] * len(w_list)               # run id of every reference position
    for k in range(1, len(w_list)):
        group[k] = group[k - 1] if abs(float(w_list[k - 1][5]) - float(w_list[k][5])) <= tie else group[k - 1] + 1
    used = [False] * len(w_list)
    bad = swaps = 0
    for k, g in enumerate(g_list):
        if not used[k] and _same_box(g, w_list[k], coord_tol, prob_tol):
            used[k] = True
            continue
        j = next((j for j in range(len(w_list)) if group[j] == group[k] and not used[j] and _same_box(g, w_list[j], coord_tol, prob_tol)), None)
        if j is None:
            bad += 1
        else:
            used[j] = True
            swaps += 1
    return bad, swaps


def _decode(logits, version, threshold, iou_threshold, scales, anchors, num_classes):
    if version == 3:
        return decode_ref.find_bounding_boxes_v3(logits, threshold, iou_threshold, scales, nms=False)
    return decode_ref.find_bounding_boxes_v2(logits, threshold, iou_threshold, anchors, num_classes, nms=False)


def _explain_image(img, ref_row, got_row, p_ref, pre_ref, pre_got, got_boxes, threshold, iou_threshold, version, e_cap, per_class, notes,
                   used):
    """One image.  ref_row / got_row: [rows, 5+C] logits; p_ref: oracle score of every row; pre_*: the oracle's pre-NMS
    candidates of the two logit sets; got_boxes: the HIP records; e_cap: the independent bound on the logit error (module
    docstring); used: {"dp", "diou"} widest bands that explained something.  Returns (differing, unexplained)."""
    thr32 = np.float64(np.float32(threshold))
    unexplained = 0
    d = np.abs(got_row.astype(np.float64) - ref_row.astype(np.float64))
    d_xy = d[:, 0:4].max(axis=1)
    d_obj = d[:, 4]
    d_cls = d[:, 5:].max(axis=1) if d.shape[1] > 5 else np.zeros(d.shape[0])

    def dp_of(s):          # score band of row s from ITS OWN logit errors, capped by the independent bound
        if version == 3:
            return 0.25 * min(float(d_obj[s]), e_cap)
        return min(float(max(d_obj[s], d_cls[s])), e_cap)

    def diou_of(k, s):
        return 4.0 * min(float(max(d_xy[k], d_xy[s])), e_cap)

    def note(msg):
        if len(notes) < 12:
            notes.append("image %d: %s" % (img, msg))

    # ---- link 1: HIP boxes against the oracle's decode + NMS of the HIP logits (exact) -------------------------------
    o_g, kept_g, by_g, iou_g, _ = nms_trace(pre_got, iou_threshold, per_class)
    own = [o_g[i].astuple() for i in range(len(o_g)) if kept_g[i]]
    g_i = [tuple(b) for b in got_boxes]
    link1_bad, link1_ties = _match_with_ties(g_i, own, 2e-5, 2e-6)
    used["ties"] = used.get("ties", 0) + link1_ties
    if link1_bad:
        # (an exp() ulp can only matter for a row whose HIP-logit score sits on the threshold itself)
        note("link 1: %d HIP box(es) differ from the oracle's decode + NMS of the HIP path's own logits" % link1_bad)
        unexplained += link1_bad

    # ---- link 2: trace of the HIP logits against the trace of the oracle logits --------------------------------------
    o_r, kept_r, by_r, iou_r, _ = nms_trace(pre_ref, iou_threshold, per_class)
    pos_r = {b.scan: i for i, b in enumerate(o_r)}
    pos_g = {b.scan: i for i, b in enumerate(o_g)}
    cand_r, cand_g = set(pos_r), set(pos_g)
    surv_r = {o_r[i].scan for i in range(len(o_r)) if kept_r[i]}
    surv_g = {o_g[i].scan for i in range(len(o_g)) if kept_g[i]}
    ok_flip = {}                                    # row -> explained?
    for s in cand_r ^ cand_g:                       # (a) candidate flips
        band = dp_of(s) + 1e-7
        good = abs(float(p_ref[s]) - thr32) <= band
        ok_flip[s] = good
        if good:
            used["dp"] = max(used["dp"], band)
        else:
            note("row %d is a candidate on one side only although its oracle score %.6f is %.2e from the threshold (band %.2e from "
                 "the row's own logit error %.2e, bound %.2e)" % (s, float(p_ref[s]), abs(float(p_ref[s]) - thr32), band,
                                                                  float(max(d_obj[s], d_cls[s] if version == 2 else 0.0)), e_cap))
    box_of = {b.scan: b for b in o_g}
    box_of.update({b.scan: b for b in o_r})         # oracle geometry wins where both exist

    differing = sorted(surv_r ^ surv_g, key=lambda s: -float(p_ref[s]))
    status = {}                                     # memo: row that survives on one side only -> explained?
    sides = {"r": (pos_r, o_r, kept_r, by_r, surv_r, cand_r), "g": (pos_g, o_g, kept_g, by_g, surv_g, cand_g)}

    def explain_survivor(s):
        """s survives on side X only.  Y = the other side."""
        if s in status:
            return status[s]                        # (None while in progress: a cycle explains nothing)
        status[s] = None
        X, Y = ("r", "g") if s in surv_r else ("g", "r")
        pos_x, _, _, _, surv_x, cand_x = sides[X]
        pos_y, o_y, _, by_y, _, cand_y = sides[Y]
        if s not in cand_y:                         # (a) not even a candidate on the other side
            good = ok_flip.get(s, False)
        else:
            k = o_y[by_y[pos_y[s]]].scan            # the survivor of Y that suppresses s there
            v = float(decode_ref.iou_score(box_of[k], box_of[s]))
            band = diou_of(k, s)
            if abs(v - iou_threshold) <= band:      # (b) borderline IoU (oracle geometry)
                good = True
                used["diou"] = max(used["diou"], band)
            elif k not in cand_x:                   # (c) the suppressor is a flipped candidate
                good = ok_flip.get(k, False)
            elif pos_x[s] < pos_x[k] and pos_y[k] < pos_y[s]:       # (d) the two changed places in the score order
                band = dp_of(s) + dp_of(k) + 2e-7
                good = abs(float(p_ref[s]) - float(p_ref[k])) <= band
                if good:
                    used["dp"] = max(used["dp"], band)
            elif k not in surv_x:                   # (c) the suppressor survives on Y only: explained iff that is
                good = bool(explain_survivor(k))
            else:                                   # both survive on X: the IoU test itself changed sides, (b) failed
                good = False
        status[s] = good
        return good

    for s in differing:
        good = bool(explain_survivor(s))
        if not good:
            unexplained += 1
            side = "the oracle" if s in surv_r else "the HIP path"
            note("row %d (oracle score %.6f) survives only in %s and no borderline score / IoU / order explains it"
                 % (s, float(p_ref[s]), side))
    for s in (cand_r ^ cand_g):                     # unexplained candidate flips that did not show up as survivors
        if not ok_flip[s] and s not in differing:
            unexplained += 1

    # common survivors: class (argmax) and order
    width = ref_row.shape[-1]
    for s in surv_r & surv_g:
        cr, cg = int(o_r[pos_r[s]].class_idx), int(o_g[pos_g[s]].class_idx)
        if cr != cg:
            gap = abs(float(ref_row[s, 5 + cr]) - float(ref_row[s, 5 + cg])) if width > 5 + max(cr, cg) else np.inf
            band = 2.0 * min(float(d_cls[s]), e_cap) + 1e-7
            if gap > band:
                unexplained += 1
                note("row %d: class %d vs %d although the oracle's two class logits are %.3e apart (band %.3e)" % (s, cg, cr, gap, band))
    common = [b.scan for i, b in enumerate(o_g) if kept_g[i] and b.scan in surv_r]
    for a, b in zip(common, common[1:]):
        if float(p_ref[a]) < float(p_ref[b]) - (dp_of(a) + dp_of(b) + 2e-7):
            unexplained += 1
            note("rows %d, %d are output in the wrong order (oracle scores %.6f < %.6f)" % (a, b, float(p_ref[a]), float(p_ref[b])))
    return len(differing), unexplained


def check(ref_logits, got_logits, got_boxes, version, threshold, iou_threshold, scales=None, anchors=None, num_classes=80,
          coord_tol=None, per_class=False, e_ref=None, abs_bound=None):
    """ref_logits: oracle fp32 logits; got_logits: the HIP path's logits of the same images;
    got_boxes: per image [(x, y, w, h, class_idx, prob)] from the HIP detect.  v3 needs `scales`
    (decode_ref.v3_scales), v2 needs `anchors`.
    The bound on the logit error that the flip bands may use (module docstring) is `abs_bound` (fp32 path: 1e-4) or
    1.5 * `e_ref` (fp16 path: e_ref = max |oracle with fp16 storage - oracle fp32| on these images); with neither the HIP
    path's own error is the only thing known and the report says so (`logit_err_bound_source`).
    Returns a JSON-able dict (see module docstring)."""
    ref_logits = np.asarray(ref_logits, np.float32)
    n = ref_logits.shape[0]
    pre = _decode(ref_logits, version, threshold, iou_threshold, scales, anchors, num_classes)
    traces = [nms_trace(c, iou_threshold, per_class) for c in pre]
    want = [[o[i] for i in range(len(o)) if k[i]] for (o, k, _, _, _) in traces]
    err = None
    if got_logits is not None:
        got_logits = np.asarray(got_logits, np.float32)
        err = float(np.max(np.abs(got_logits.astype(np.float64) - ref_logits.astype(np.float64))))
    e = err if err is not None else 0.0
    if abs_bound is not None:
        e_cap, cap_src = float(abs_bound), "absolute bound %g (fp32 contract)" % abs_bound
    elif e_ref is not None:
        e_cap, cap_src = 1.5 * float(e_ref), "1.5 x e_ref, e_ref = max|oracle(fp16 storage) - oracle(fp32)| = %.4g on these images" % e_ref
    else:
        e_cap, cap_src = e, "none given: the HIP path's own max error (bands not independent)"
    within = bool(e <= e_cap * (1.0 + 1e-9) + 1e-12)
    # propagated error bounds at the cap: d sigmoid <= e/4; v2 score = sigmoid * softmax-max, |d| <= e/4 + e/2 < e;
    # box centre moves <= e/4 of a cell, box size by a factor exp(+-e): IoU of two boxes moves by at most ~4e for e << 1
    dp_cap = 0.25 * e_cap if version == 3 else e_cap
    diou_cap = 4.0 * e_cap
    p = _scores(ref_logits, version, num_classes)
    prob_margin = float(np.min(np.abs(p.astype(np.float64) - np.float64(np.float32(threshold)))))
    iou_margin = float(min(t[4] for t in traces)) if n else np.inf
    identity_required = bool(prob_margin > dp_cap and iou_margin > diou_cap)
    if coord_tol is None:       # a box is "the same" when class and cell agree; fp32: float rounding, fp16: a few 1e-3 of the image
        coord_tol = max(2e-5, 2.0 * min(e, e_cap))
    prob_tol = max(2e-6, 0.25 * min(e, e_cap) if version == 3 else min(e, e_cap))
    matched = unmatched = 0
    identical = True
    for i in range(n):
        w_i = [b.astuple() for b in want[i]]
        g_i = [tuple(b) for b in got_boxes[i]]
        same_order = len(w_i) == len(g_i) and _match_with_ties(g_i, w_i, coord_tol, prob_tol)[0] == 0     # (up to float32 score ties: kTieUlps)
        if same_order:
            matched += len(w_i)
            continue
        identical = False
        used_ = [False] * len(g_i)
        for w in w_i:                                   # set comparison for the report
            hit = next((k for k, g in enumerate(g_i) if not used_[k] and _same_box(g, w, coord_tol, prob_tol)), None)
            if hit is None:
                unmatched += 1
            else:
                used_[hit] = True
                matched += 1
        unmatched += used_.count(False)
    # every difference explained?
    notes = []
    differing = unexplained = 0
    used = {"dp": 0.0, "diou": 0.0}
    if got_logits is None:
        unexplained = unmatched                          # nothing to trace the differences with
    else:
        pre_got = _decode(got_logits, version, threshold, iou_threshold, scales, anchors, num_classes)
        rows_ref = ref_logits.reshape(n, -1, ref_logits.shape[-1] if version == 3 else 5 + num_classes)
        rows_got = got_logits.reshape(rows_ref.shape)
        for i in range(n):
            d, u = _explain_image(i, rows_ref[i], rows_got[i], p[i], pre[i], pre_got[i], got_boxes[i], threshold, iou_threshold,
                                  version, e_cap, per_class, notes, used)
            differing += d
            unexplained += u
    return {"images_checked": int(n), "max_abs_logit_err": err, "logit_err_bound": e_cap, "logit_err_bound_source": cap_src,
            "logit_err_within_bound": within, "e_ref": (None if e_ref is None else float(e_ref)),
            "box_set_match": bool(identical),
            # the same boxes irrespective of output order (a swap of two near-equal scores moves a box, it does not change the set)
            "box_set_match_unordered": bool(unmatched == 0 and sum(len(b) for b in want) == sum(len(b) for b in got_boxes)),
            "boxes_ref": int(sum(len(b) for b in want)), "boxes_hip": int(sum(len(b) for b in got_boxes)),
            "boxes_matched": int(matched), "boxes_unmatched": int(unmatched),
            "rows_differing": int(differing), "boxes_unexplained": int(unexplained), "unexplained_notes": notes,
            "prob_margin": prob_margin, "iou_margin": (None if not np.isfinite(iou_margin) else iou_margin),
            "prob_flip_band": used["dp"], "iou_flip_band": used["diou"], "prob_flip_band_cap": dp_cap, "iou_flip_band_cap": diou_cap,
            "score_ties_reordered": int(used.get("ties", 0)),
            "identity_required": identity_required,
            "nms_mode": "per_class" if per_class else "agnostic",
            "threshold": float(threshold), "iou_threshold": float(iou_threshold),
            "reference": "oracle fp32 pipeline (forward_ref + decode_ref), restatement of net/yolo.py:83-86"}


def assert_ok(rep):
    """The gate: the logit error within the independent bound; identity where the margins demand it; elsewhere EVERY differing
    box must be explained by a borderline score, IoU or order (module docstring) -- a defect that drops or adds even one box
    away from the thresholds fails, and so does a forward whose error exceeds what fp16 storage explains."""
    assert rep["logit_err_within_bound"], "max |logit error| %.4g exceeds the bound %.4g (%s)" % (
        rep["max_abs_logit_err"], rep["logit_err_bound"], rep["logit_err_bound_source"])
    if rep["identity_required"]:
        assert rep["box_set_match"], "box sets differ although the margins exceed the logit error: %r" % (rep,)
    assert rep["boxes_unexplained"] == 0, "box differences that no borderline score / IoU / order explains: %r" % (rep,)
