"""CPU: the conv-stack restatement (PARITY UNPINNED by the reference, see oracle/__init__.py),
cross-checked two independent ways, plus the known-answer totals of SURVEY 4 / App. B, F."""
import numpy as np
import pytest
import torch

from helpers import to_oracle
from oracle import cases, forward_ref as FR, topology as T
from tensorflow_yolo_amd.net import v2, v3

NAMES80 = ["c%d" % i for i in range(80)]


def test_known_answer_totals():
    assert T.conv_weight_count(T.yolov2(5, 80)) == 50983561
    assert T.conv_weight_count(T.yolov2(5, 20)) == 50676061
    assert T.conv_weight_count(T.yolov2(1, 1)) == 50554086
    assert T.conv_weight_count(T.yolov3(cases.COCO_V3_ANCHORS, 80)) == 62001757
    assert T.conv_weight_count(T.tiny_yolov2(5, 20)) == 15867885
    assert round(T.conv_flops(T.yolov2(5, 80)) / 1e9, 3) == 29.464
    assert round(T.conv_flops(T.yolov3(cases.COCO_V3_ANCHORS, 80)) / 1e9, 3) == 65.864
    assert round(T.conv_flops(T.yolov3(cases.COCO_V3_ANCHORS, 80, (608, 608, 3))) / 1e9, 3) == 140.692
    assert round(T.conv_flops(T.tiny_yolov2(5, 20)) / 1e9, 3) == 6.971


def test_shapes_and_route_sources():
    L = T.yolov2(5, 80)
    S = T.shapes(L)
    assert len(L) == 32 and S[-1] == (13, 13, 425)
    assert L[26] == ("route", [17]) and L[29] == ("route", [28, 25]) and S[29] == (13, 13, 1280)
    L = T.yolov3(cases.COCO_V3_ANCHORS, 80, (608, 608, 3))
    S = T.shapes(L)
    assert len(L) == 109
    assert L[84] == ("route", [80]) and L[87] == ("route", [86, 62]) and L[96] == ("route", [92]) and L[99] == ("route", [98, 37])
    assert S[87] == (38, 38, 768) and S[99] == (76, 76, 384)
    assert [S[i][:2] for i in (83, 95, 107)] == [(19, 19), (38, 38), (76, 76)]
    assert sum(S[i][0] * S[i][1] * 3 for i in (83, 95, 107)) == 22743
    # anchors reversed: the coarsest head gets the largest three (pixels here; /stride in the yolo layer)
    assert L[83][2][0] == (116, 90) and L[107][2][0] == (10, 13)


def test_product_layer_lists_equal_oracle_topology():
    anchors = np.reshape(cases.COCO_V2_ANCHORS, [-1, 2])
    assert to_oracle(v2.create_full_network(anchors, NAMES80, False)) == T.yolov2(5, 80)
    assert to_oracle(v2.create_tiny_network(np.reshape(cases.VOC_TINY_ANCHORS, [-1, 2]), NAMES80[:20], False)) == T.tiny_yolov2(5, 20)
    a3 = np.reshape(cases.COCO_V3_ANCHORS, [-1, 2])
    for size in (416, 608):
        got = to_oracle(v3.create_network(a3, NAMES80, False, input_shape=(size, size, 3)))
        want = T.yolov3(cases.COCO_V3_ANCHORS, 80, (size, size, 3))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            if g[0] == "yolo":      # product: grid units (layers.py:131); oracle topology: pixels
                stride = size / T.shapes(want)[w[1]][0]
                assert g[:2] == w[:2] and np.allclose(np.array(g[2]) * stride, np.array(w[2]))
            else:
                assert g == w


@pytest.mark.parametrize("k,s,h,w", [(3, 1, 7, 9), (3, 2, 8, 8), (3, 2, 7, 9), (1, 1, 5, 6)])
def test_conv_torch_vs_naive_numpy(k, s, h, w):
    rng = np.random.RandomState(k * 10 + s)
    x = rng.randn(2, h, w, 5).astype(np.float32)
    kern = rng.randn(6, 5, k, k).astype(np.float32)
    wd = {"kernel_oihw": kern, "bias": np.zeros(6, np.float32)}
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double()
    y = FR._conv(xt, wd, k, s, False, "linear", torch.float64).permute(0, 2, 3, 1).numpy()
    pad_b = (k - 1) // 2
    pad_a = (k - 1) - pad_b
    ref = FR.conv_naive_numpy(x, kern, s, pad_b, pad_a)
    assert y.shape == ref.shape
    assert np.allclose(y, ref, atol=1e-10)


def test_stride2_is_symmetric_darknet_padding_not_tf_same():
    # pad 1 before / 1 after: output pixel (0,0) must see input row -1 as zero and row 0 at kh=1
    x = np.zeros((1, 4, 4, 1), np.float32)
    x[0, 0, 0, 0] = 1.0
    kern = np.arange(9, dtype=np.float32).reshape(1, 1, 3, 3)
    wd = {"kernel_oihw": kern, "bias": np.zeros(1, np.float32)}
    y = FR._conv(torch.from_numpy(x).permute(0, 3, 1, 2), wd, 3, 2, False, "linear", torch.float32)
    assert y.shape[-2:] == (2, 2)
    assert float(y[0, 0, 0, 0]) == 4.0      # centre tap (1,1), not TF-SAME's (0,0)


def test_batchnorm_and_leaky():
    wd = {"kernel_oihw": np.ones((1, 1, 1, 1), np.float32), "gamma": np.array([2.0], np.float32),
          "beta": np.array([0.5], np.float32), "mean": np.array([1.0], np.float32), "var": np.array([4.0], np.float32)}
    x = torch.tensor([[[[3.0, -5.0]]]], dtype=torch.float64)
    y = FR._conv(x, wd, 1, 1, True, "leaky", torch.float64).numpy().ravel()
    inv = 2.0 / np.sqrt(4.0 + 1e-5)
    assert np.allclose(y, [(3 - 1) * inv + 0.5, 0.1 * ((-5 - 1) * inv + 0.5)], atol=1e-7)


def test_maxpool_pad_semantics():
    x = -torch.ones(1, 1, 3, 3)
    y = FR._maxpool(x, 2, 2)                # odd H: last window sees the zero padding
    assert y.shape[-2:] == (2, 2) and float(y[0, 0, 0, 0]) == -1.0 and float(y[0, 0, 1, 1]) == 0.0
    y1 = FR._maxpool(x, 2, 1)               # SAME stride 1: padding ignored
    assert y1.shape[-2:] == (3, 3) and float(y1.max()) == -1.0


def test_reorg_is_block_major():
    x = torch.arange(2 * 4 * 4 * 3, dtype=torch.float32).view(2, 4, 4, 3)      # NHWC
    y = FR._reorg(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    for di in range(2):
        for dj in range(2):
            for c in range(3):
                assert torch.equal(y[:, :, :, (di * 2 + dj) * 3 + c], x[:, di::2, dj::2, c])
    assert not torch.equal(y.permute(0, 3, 1, 2), torch.nn.functional.pixel_unshuffle(x.permute(0, 3, 1, 2), 2))


def test_weight_stream_order_and_strictness():
    L = [("input", 4, 4, 3), ("conv", 0, 2, 3, 1, True, "leaky"), ("conv", 1, 5, 1, 1, False, "linear")]
    n = T.conv_weight_count(L)
    assert n == (2 * 3 * 9 + 8) + (5 * 2 + 5)
    flat = np.arange(n, dtype=np.float32)
    W = FR.parse_darknet_weights(L, flat)
    assert W[1]["beta"][0] == 0 and W[1]["gamma"][0] == 2 and W[1]["mean"][0] == 4 and W[1]["var"][0] == 6
    assert W[1]["kernel_oihw"].shape == (2, 3, 3, 3) and W[1]["kernel_oihw"][0, 0, 0, 1] == 9
    assert W[2]["bias"][0] == 8 + 54
    with pytest.raises(AssertionError):
        FR.parse_darknet_weights(L, flat[:-1])


def test_small_v3_forward_runs_and_fp16_emulation_is_close():
    L = T.yolov3(cases.COCO_V3_ANCHORS, 80, (64, 64, 3))
    rng = np.random.RandomState(0)
    w = (rng.randn(T.conv_weight_count(L)) * 0.02).astype(np.float32)
    # make BN variances positive
    Wd = FR.parse_darknet_weights(L, w)
    for d in Wd.values():
        if "var" in d:
            d["var"][:] = np.abs(d["var"]) + 0.5
            d["gamma"][:] = 1.0
    x = rng.rand(1, 64, 64, 3).astype(np.float32)
    y = FR.forward(L, Wd, x)
    assert y.shape == (1, (4 + 16 + 64) * 3, 85)
    y16 = FR.forward(L, Wd, x, storage="fp16")
    assert np.max(np.abs(y - y16)) < 5e-2 * max(1.0, np.max(np.abs(y)))
