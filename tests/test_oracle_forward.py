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


# ---- topology + weight order pinned to the reference's OWN builders --------------------------------------------
# tests/golden/topology_*.json are what net/v2.py:11-60 / net/v3.py:9-94 built when run under a recording `tensorflow`
# module in the build container (oracle/gen_topology.py): per layer the class, the recorded TF ops with their
# arguments, the source layers by `.out` identity, the shape and `.variable_names`.

def _fixture_to_tuples(fx):
    """golden records -> oracle/topology tuples, checking the op arguments the restatement relies on."""
    out = []
    for l in fx["layers"]:
        ops = {o["op"]: o for o in l["ops"]}
        names = [o["op"] for o in l["ops"] if o["op"] != "identity"]     # tf.identity only renames the last layer (v2.py:59, v3.py:93)
        c = l["class"]
        if c == "input_layer":
            assert names == ["placeholder"] and ops["placeholder"]["dtype"] == "float32" and l["shape"][0] is None
            out.append(("input",) + tuple(l["shape"][1:]))
        elif c == "conv2d_bn_act":
            cv = ops["conv2d"]
            k, s = cv["kernel_size"][0], cv["strides"][0]
            bn, leaky = "batch_normalization" in ops, "leaky_relu" in ops
            assert names == (["pad"] if s > 1 else []) + ["conv2d"] + (["batch_normalization"] if bn else []) + (["leaky_relu"] if leaky else [])
            assert cv["kernel_size"] == [k, k] and cv["strides"] == [s, s] and cv["padding"] == ("SAME" if s == 1 else "VALID")     # layers.py:28-30
            assert cv["use_bias"] == (not bn)                                                                                    # layers.py:37
            if s > 1:       # layers.py:9-14: (k-1)//2 before, the rest after, H and W only, zeros
                a = (k - 1) // 2
                assert ops["pad"]["paddings"] == [[0, 0], [a, k - 1 - a], [a, k - 1 - a], [0, 0]] and ops["pad"]["mode"] == "CONSTANT"
            if bn:
                assert ops["batch_normalization"]["epsilon"] == 1e-5 and ops["batch_normalization"]["training"] is False      # layers.py:5
            if leaky:
                assert ops["leaky_relu"]["alpha"] == 0.1                                                                          # layers.py:6
            out.append(("conv", l["src"][0], cv["filters"], k, s, bn, "leaky" if leaky else "linear"))
        elif c == "max_pool2d":
            mp = ops["max_pooling2d"]
            k, s = mp["pool_size"][0], mp["strides"][0]
            assert names == (["pad"] if s > 1 else []) + ["max_pooling2d"] and mp["padding"] == ("SAME" if s == 1 else "VALID")
            if s > 1:
                assert ops["pad"]["paddings"] == [[0, 0], [0, 1], [0, 1], [0, 0]]           # (k-1)//2 = 0 before, 1 after
            out.append(("maxpool", l["src"][0], k, s))
        elif c == "route":
            assert names == ["concat"] and ops["concat"]["axis"] == 3
            out.append(("route", list(l["src"])))
        elif c == "reorg":
            e = ops["extract_image_patches"]
            assert names == ["extract_image_patches"] and e["ksizes"] == e["strides"] and e["rates"] == [1, 1, 1, 1] and e["padding"] == "VALID"
            out.append(("reorg", l["src"][0], e["ksizes"][1]))
        elif c == "shortcut":
            assert names == ["add"]
            out.append(("shortcut", l["src"][0], l["src"][1]))
        elif c == "upsample":
            assert names == ["resize_nearest_neighbor"]
            out.append(("upsample", l["src"][0], l["shape"][1] // fx["layers"][l["src"][0]]["shape"][1]))
        elif c == "yolo_layer":
            assert names == ["reshape"] and ops["reshape"]["new_shape"] == [-1, l["h"] * l["w"] * l["b"], 5 + fx["num_classes"]]
            out.append(("yolo", l["src"][0], [tuple(a) for a in l["anchors"]]))
        elif c == "detection_layer":
            assert names == ["concat"] and ops["concat"]["axis"] == 1
            out.append(("detection", list(l["yolos"])))
        else:
            raise AssertionError(c)
    return out


@pytest.mark.parametrize("name", ["v2_416", "v3_416", "v3_608"])
def test_topology_and_weight_order_equal_the_reference_builders(name):
    import json
    import os
    from helpers import GOLDEN
    fx = json.load(open(os.path.join(GOLDEN, "topology_%s.json" % name)))
    size = tuple(fx["input_shape"])
    anchors = np.reshape(fx["anchors"], [-1, 2])
    if fx["net"] == "v2":
        net = v2.create_full_network(anchors, NAMES80, False, input_shape=size)
        oracle = T.yolov2(len(anchors), 80, size)
    else:
        net = v3.create_network(anchors, NAMES80, False, input_shape=size)
        oracle = T.yolov3(fx["anchors"], 80, size)
    ref = _fixture_to_tuples(fx)
    got = to_oracle(net)
    assert len(ref) == len(got) == len(oracle)
    S = T.shapes(oracle)
    for i, (r, g, o) in enumerate(zip(ref, got, oracle)):
        if r[0] == "yolo":          # reference and product: grid units, float64 (layers.py:131); oracle topology: pixels
            stride = size[0] / S[o[1]][0]
            assert r[:2] == g[:2] == o[:2]
            assert np.array_equal(np.array(r[2]), np.array(g[2])) and np.allclose(np.array(r[2]) * stride, np.array(o[2]))
        else:
            assert r == g == o, (i, r, g, o)
        shp = fx["layers"][i]["shape"][1:]      # output shapes as TensorFlow's shape inference gives them
        if r[0] == "yolo":          # [-1, h*w*b, 5+C] view of the head conv (layers.py:133)
            h, w, ch = S[i]
            assert shp == [h * w * len(r[2]), 85] and ch == len(r[2]) * 85 and net[i].rows == shp[0]
        elif r[0] == "detection":
            assert shp == [sum(net[j].rows for j in r[1]), 85] == [net[i].out.hwc[0], net[i].out.hwc[2]]
        else:
            assert tuple(shp) == tuple(S[i]) == tuple(net[i].out.hwc), i
    # Darknet stream order (net/base.py:26-46 walks layers x variable_names): same names in the same order
    assert [l["variable_names"] for l in fx["layers"]] == [list(l.variable_names) for l in net]
    # ... and the oracle's weight parser consumes per conv exactly those tensors in that order
    flat = np.arange(T.conv_weight_count(oracle), dtype=np.float32)
    parsed = FR.parse_darknet_weights(oracle, flat)
    pos = 0
    for i, l in enumerate(fx["layers"]):
        for vn in l["variable_names"]:
            key = vn.rsplit("/", 1)[1]
            arr = parsed[i][{"beta": "beta", "gamma": "gamma", "moving_mean": "mean", "moving_variance": "var", "bias": "bias", "kernel": "kernel_oihw"}[key]]
            assert arr.ravel()[0] == pos, (i, vn)
            pos += arr.size
    assert pos == flat.size
