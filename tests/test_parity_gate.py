"""CPU: the parity gate (oracle/parity.py) itself -- fp16-sized logit noise must come out fully explained, and a defect
that drops / adds / reorders boxes away from the thresholds must not."""
import numpy as np
import pytest

from oracle import cases, decode_ref, parity


def _case(version):
    if version == 3:
        c = dict(cases.CASES["v3_320_lowthr"], threshold=0.5, iou=0.6, obj_shift=-3.8, seed=77, batch=2, ties=0)
        head = cases.make_head(c)
        head[..., 0:4] *= np.float32(0.2)           # anchor-sized boxes near the cell centres: neighbours overlap, NMS has work
        kw = dict(scales=decode_ref.v3_scales(c["anchors"], (c["input"], c["input"])))
    else:
        c = dict(cases.CASES["v2_416_thr05"], seed=78, ties=0, batch=1, obj_shift=0.0)
        head = cases.make_head(c)
        kw = dict(anchors=c["anchors"], num_classes=c["classes"])
    return c, head, kw


def _hip_like(c, logits, kw, per_class=False):
    """What a correct HIP path returns for these logits: the oracle's decode + NMS of them."""
    if c["version"] == 3:
        pre = decode_ref.find_bounding_boxes_v3(logits, c["threshold"], c["iou"], kw["scales"], nms=False)
    else:
        pre = decode_ref.find_bounding_boxes_v2(logits, c["threshold"], c["iou"], kw["anchors"], kw["num_classes"], nms=False)
    nms = decode_ref.non_maximum_suppression_per_class if per_class else decode_ref.non_maximum_suppression
    return [[b.astuple() for b in nms(p, c["iou"])] for p in pre]


@pytest.mark.parametrize("version", [2, 3])
@pytest.mark.parametrize("noise", [0.0, 2e-2, 6e-2])
def test_noise_is_explained(version, noise):
    c, head, kw = _case(version)
    rng = np.random.RandomState(5)
    got = (head + rng.uniform(-noise, noise, size=head.shape)).astype(np.float32)
    boxes = _hip_like(c, got, kw)
    # the bound the bands may use comes from OUTSIDE the path under test: here the noise amplitude stands in for 1.5 x e_ref
    bound = dict(abs_bound=1e-4) if noise == 0.0 else dict(e_ref=noise / 1.5)
    rep = parity.check(head, got, boxes, version, c["threshold"], c["iou"], **bound, **kw)
    assert rep["boxes_ref"] > 20
    assert rep["boxes_unexplained"] == 0, rep
    assert rep["logit_err_within_bound"]
    parity.assert_ok(rep) if not rep["identity_required"] or rep["box_set_match"] else None
    if noise == 0.0:
        assert rep["box_set_match"] and rep["identity_required"] and rep["rows_differing"] == 0
        assert rep["prob_flip_band"] == 0.0 and rep["iou_flip_band"] == 0.0
    if noise == 6e-2:
        assert rep["rows_differing"] > 0, "the case is too easy: no flip to explain at this noise level"
    # the bands that explained something come from the rows' own errors: never wider than the cap the bound allows
    assert rep["prob_flip_band"] <= rep["prob_flip_band_cap"] + 3e-7 and rep["iou_flip_band"] <= rep["iou_flip_band_cap"]


@pytest.mark.parametrize("version", [2, 3])
def test_a_forward_defect_cannot_widen_its_own_bands(version):
    """Round-3 gate: dp, diou came from the HIP path's OWN max |logit error|, so a forward defect that raised the error also raised
    the bands that should catch it.  Now: logits off by 0.09 everywhere (fp16 storage explains ~0.02 on these sizes) must FAIL,
    although every box the perturbed logits decode to is reported faithfully (link 1 holds)."""
    c, head, kw = _case(version)
    rng = np.random.RandomState(11)
    got = (head + rng.choice([-0.09, 0.09], size=head.shape)).astype(np.float32)
    boxes = _hip_like(c, got, kw)
    rep = parity.check(head, got, boxes, version, c["threshold"], c["iou"], e_ref=0.02, **kw)
    assert not rep["logit_err_within_bound"] and rep["logit_err_bound"] == pytest.approx(0.03)
    assert rep["iou_flip_band_cap"] == pytest.approx(0.12) and rep["iou_flip_band"] <= 0.12
    assert rep["rows_differing"] > 0
    if version == 3:            # score = sigmoid(obj): moves by up to 0.09 / 4, the band stays at 0.03 / 4 -> flips beyond it stay unexplained
        assert rep["boxes_unexplained"] > 0, rep      # (v2's score band is e itself: |d(sigmoid * softmax-max)| < e, so 0.03 covers what 0.09 does to it)
    with pytest.raises(AssertionError):
        parity.assert_ok(rep)
    # the same logits under the old rule (no independent bound: the path's own error) would have been waved through
    old = parity.check(head, got, boxes, version, c["threshold"], c["iou"], **kw)
    assert old["logit_err_bound_source"].startswith("none given") and old["boxes_unexplained"] == 0 and old["logit_err_within_bound"]


@pytest.mark.parametrize("version", [2, 3])
@pytest.mark.parametrize("defect", ["drop", "extra", "swap", "class", "drop_in_logits"])
def test_defects_are_not_explained(version, defect):
    c, head, kw = _case(version)
    rng = np.random.RandomState(6)
    got = (head + rng.uniform(-2e-2, 2e-2, size=head.shape)).astype(np.float32)
    boxes = _hip_like(c, got, kw)
    img = 0
    assert len(boxes[img]) > 8
    if defect == "drop":                    # a decode that loses a confident box
        del boxes[img][1]
    elif defect == "extra":                 # an NMS that lets a suppressed box through
        b = list(boxes[img][0]); b[5] = b[5] - 0.01; b[0] += 1e-3
        boxes[img].insert(1, tuple(b))
    elif defect == "swap":                  # wrong output order
        boxes[img][0], boxes[img][3] = boxes[img][3], boxes[img][0]
    elif defect == "class":                 # wrong argmax
        b = list(boxes[img][2]); b[4] = (int(b[4]) + 1) % c["classes"]; boxes[img][2] = tuple(b)
    elif defect == "drop_in_logits":        # a conv defect that wipes a confident row WITHOUT showing in max |error| beyond fp16 noise
        # (the row's objectness moves by less than the band would allow only if it were borderline: it is not)
        p = parity._scores(head, version, c["classes"])[img]
        row = int(np.argmax(p))
        g2 = got.reshape(got.shape[0], -1, 5 + c["classes"]).copy()
        g2[img, row, 4] -= 30.0
        got = g2.reshape(got.shape)
        boxes = _hip_like(c, got, kw)
        rep = parity.check(head, got, boxes, version, c["threshold"], c["iou"], e_ref=2e-2 / 1.5, **kw)
        assert rep["max_abs_logit_err"] > 1.0 and not rep["logit_err_within_bound"]     # this kind shows in the logit error
        with pytest.raises(AssertionError):
            parity.assert_ok(rep)
        return
    rep = parity.check(head, got, boxes, version, c["threshold"], c["iou"], e_ref=2e-2 / 1.5, **kw)
    assert rep["boxes_unexplained"] >= 1, rep
    with pytest.raises(AssertionError):
        parity.assert_ok(rep)


def test_per_class_trace_matches_per_class_nms():
    c, head, kw = _case(3)
    pre = decode_ref.find_bounding_boxes_v3(head, c["threshold"], c["iou"], kw["scales"], nms=False)
    for p in pre:
        o, kept, _, _, _ = parity.nms_trace(p, c["iou"], per_class=True)
        a = [o[i].astuple() for i in range(len(o)) if kept[i]]
        b = [x.astuple() for x in decode_ref.non_maximum_suppression_per_class(p, c["iou"])]
        assert a == b
        agn = decode_ref.non_maximum_suppression(p, c["iou"])
        assert len(b) >= len(agn)
    boxes = _hip_like(c, head, kw, per_class=True)
    rep = parity.check(head, head, boxes, 3, c["threshold"], c["iou"], per_class=True, **kw)
    assert rep["box_set_match"] and rep["boxes_unexplained"] == 0 and rep["nms_mode"] == "per_class"


def test_score_ties_within_float32_rounding_may_swap_and_nothing_else():
    """Two boxes whose scores differ by one float32 ulp may come out in either order (NumPy's float32 exp and the GPU's expf are both a
    few ulp accurate and not the same function; the reference's stable sort then breaks the tie by scan order on one side only):
    `_match_with_ties` accepts that permutation, counts it, and still rejects a swap of boxes whose scores are further apart, a changed
    box, or a missing one."""
    p0 = float(np.float32(0.50947118))
    p1 = float(np.nextafter(np.float32(p0), np.float32(1.0)))           # one ulp above
    a = (0.10, 0.20, 0.30, 0.40, 3, p1)
    b = (0.60, 0.70, 0.05, 0.06, 7, p0)
    c = (0.80, 0.10, 0.02, 0.02, 1, 0.40)
    assert parity._match_with_ties([a, b, c], [a, b, c], 2e-5, 2e-6) == (0, 0)
    bad, swaps = parity._match_with_ties([b, a, c], [a, b, c], 2e-5, 2e-6)
    assert (bad, swaps) == (0, 2)                                       # both positions of the tied pair matched across
    far = (0.60, 0.70, 0.05, 0.06, 7, p0 - 1e-5)
    assert parity._match_with_ties([far, a, c], [a, far, c], 2e-5, 2e-6)[0] > 0          # 1e-5 apart: a real order error
    other = (0.61, 0.70, 0.05, 0.06, 7, p0)
    assert parity._match_with_ties([other, a, c], [a, b, c], 2e-5, 2e-6)[0] > 0          # not the same box
    assert parity._match_with_ties([a, c], [a, b, c], 2e-5, 2e-6)[0] > 0
