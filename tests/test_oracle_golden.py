"""CPU: the oracle's decode + NMS restatement against the committed outputs of the REFERENCE's own
functions (tests/golden/*.npz, produced by oracle/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import cases, decode_ref


def _load(name):
    g = np.load(os.path.join(GOLDEN, "decode_%s.npz" % name))
    meta = json.loads(str(g["meta"]))
    return g, meta


@pytest.mark.parametrize("full_scan", [False, True])
@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_decode_matches_reference(name, full_scan):
    """full_scan: the per-cell Python scan exactly as the reference's triple loop walks it (what bench.py times as the CPU
    decode baseline) instead of the vectorised pre-filter -- both against the reference's own outputs"""
    c = cases.CASES[name]
    if full_scan and c["input"] >= 608 and c["batch"] > 1:
        pytest.skip("the full scan of the large multi-image case takes tens of seconds; the single-image 608 case covers it")
    g, meta = _load(name)
    assert meta["source"] == "reference functions"
    head = cases.make_head(name)
    for nms in (False, True):
        if c["version"] == 2:
            mine = decode_ref.find_bounding_boxes_v2(head, c["threshold"], c["iou"], c["anchors"], c["classes"], nms=nms, full_scan=full_scan)
        else:
            sc = decode_ref.v3_scales(c["anchors"], (c["input"], c["input"]))
            mine = decode_ref.find_bounding_boxes_v3(head, c["threshold"], c["iou"], sc, nms=nms, full_scan=full_scan)
        for i in range(c["batch"]):
            want = g[("post%d" if nms else "pre%d") % i]
            got = decode_ref.boxes_to_array(mine[i])
            assert got.shape == want.shape
            if len(want):
                assert np.array_equal(got[:, 4], want[:, 4])                  # class ids, order
                assert np.allclose(got, want, rtol=0, atol=1e-6)


def test_fixture_margins():
    """The fixtures must not sit on a decision boundary: no candidate within 1e-5 of the score
    threshold (so float32 exp rounding on another platform cannot flip a box)."""
    for name, c in cases.CASES.items():
        g, _ = _load(name)
        for i in range(c["batch"]):
            pre = g["pre%d" % i]
            if len(pre):
                assert np.min(np.abs(pre[:, 5] - c["threshold"])) > 1e-5, name


@pytest.mark.parametrize("name", sorted(cases.NMS_CASES))
def test_nms_cases_match_reference(name):
    g = np.load(os.path.join(GOLDEN, "nms_cases.npz"))
    boxes, thr = cases.NMS_CASES[name]
    bb = [decode_ref.Box(np.float32(b[0]), np.float32(b[1]), np.float64(b[2]), np.float64(b[3]), b[4], np.float32(b[5]))
          for b in boxes]
    kept = decode_ref.boxes_to_array(decode_ref.non_maximum_suppression(bb, thr))
    assert kept.shape == g[name].shape
    assert np.allclose(kept, g[name], rtol=0, atol=1e-7)


def test_engineered_semantics():
    g = np.load(os.path.join(GOLDEN, "nms_cases.npz"))
    assert len(g["iou_exactly_at_threshold"]) == 2          # iou == thr suppresses (>=)
    assert len(g["iou_just_below_threshold"]) == 2          # iou < thr keeps both
    assert [int(c) for c in g["equal_prob_keeps_scan_order"][:, 4]] == [3, 4, 6]   # stable order, 5 suppressed by 3
    assert len(g["zero_area_union_floor"]) == 2             # 0/1e-8 = 0 < thr
    assert len(g["empty"]) == 0


def test_preprocess_restatement_properties():
    """oracle/preprocess_ref.py (OpenCV 8-bit INTER_LINEAR, unpinned: no cv2 here): identity at equal size, constants
    stay constant, corners of an up-scale equal the source corners, a 2x2 -> 1x1 reduction is the rounded mean, and
    the result is within one grey level of a float64 bilinear evaluation at half-pixel centres"""
    from oracle import preprocess_ref as P
    rng = np.random.RandomState(1)
    im = rng.randint(0, 256, size=(9, 13, 3)).astype(np.uint8)
    assert np.array_equal(P.resize_linear_u8(im, 9, 13), im)
    assert (P.resize_linear_u8(np.full((5, 6, 3), 77, np.uint8), 11, 3) == 77).all()
    up = P.resize_linear_u8(im, 27, 39)
    assert (up[0, 0] == im[0, 0]).all() and (up[-1, -1] == im[-1, -1]).all()
    q = np.array([[[10], [20]], [[30], [41]]], np.uint8).repeat(3, axis=2)
    assert int(P.resize_linear_u8(q, 1, 1)[0, 0, 0]) in (25, 26)
    H, W, dh, dw = 9, 13, 20, 7
    ys = np.clip((np.arange(dh) + 0.5) * H / dh - 0.5, 0, H - 1); xs = np.clip((np.arange(dw) + 0.5) * W / dw - 0.5, 0, W - 1)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int); y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    f = im.astype(np.float64)
    ref = (f[y0][:, x0] * (1 - fx) + f[y0][:, x1] * fx) * (1 - fy) + (f[y1][:, x0] * (1 - fx) + f[y1][:, x1] * fx) * fy
    assert np.abs(P.resize_linear_u8(im, dh, dw).astype(np.float64) - ref).max() <= 1.0
    assert P.preprocess(im, (4, 4)).dtype == np.float32
