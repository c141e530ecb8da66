"""GPU: every kernel / fusion of the HIP path against the CPU oracle, through the C ABI.

fp32 path: compared with the oracle's fp32 arithmetic (the reference's), tolerance 1e-4 of the
tensor's magnitude (north_star: logits within 1e-4).
fp16 path: compared with the oracle run with the same fp16 storage roundings (tight bound:
2 fp16 ulps of the tensor magnitude) -- fp16 storage vs the fp32 reference is reported by
test_gpu_nets.py.
"""
import zlib

import numpy as np
import pytest

from helpers import new_graph, rel_err, run_hip, to_oracle
from oracle import forward_ref as FR, topology as T
from tensorflow_yolo_amd.net import layers as PL, synth

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-4, "fp16": 2.5e-3}


def oracle_out(net, w, x, dtype, keep=None):
    L = to_oracle(net)
    return FR.forward(L, w, x, keep=keep, storage="fp16" if dtype == "fp16" else None)


def check_graph(net, x, dtype, seed=0, read=(), max_batch=None, tile=None, **engine_kw):
    w = synth.darknet_stream(net, seed=seed)
    want, kept = oracle_out(net, w, x, dtype, keep=set(read))
    got, eng = run_hip(net, w, x, dtype, keep_all=bool(read), max_batch=max_batch, force_tile=tile, **engine_kw)
    assert got.shape == want.shape, (got.shape, want.shape)
    errs = {"final": rel_err(got, want)}
    for i in read:
        errs[i] = rel_err(eng.read_layer(i, x.shape[0]), kept[i])
    print(dtype, "kernels=%d" % eng.num_kernels, {k: "%.2e" % v for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not v <= TOL[dtype]}
    assert not bad, "relative error above %g: %s\n%s" % (TOL[dtype], bad, eng.describe())
    return eng


CONV_CASES = {
    # name: (B, H, W, Cin, Cout, k, s, bn, act)
    "first_3to32": (2, 17, 19, 3, 32, 3, 1, True, "leaky"),
    "first_3to16_tiny": (2, 16, 16, 3, 16, 3, 1, True, "leaky"),
    "c16to32": (2, 12, 12, 16, 32, 3, 1, True, "leaky"),
    "c32to64_s2_odd": (2, 15, 13, 32, 64, 3, 2, True, "leaky"),
    "c32to64_s1": (1, 20, 20, 32, 64, 3, 1, True, "leaky"),
    "c64to128": (2, 9, 11, 64, 128, 3, 1, True, "leaky"),
    "c128to64_1x1": (2, 9, 11, 128, 64, 1, 1, True, "leaky"),
    "c64to32_1x1": (2, 10, 10, 64, 32, 1, 1, True, "leaky"),
    "c128to256_s2": (2, 10, 10, 128, 256, 3, 2, True, "leaky"),
    "c256to512_linear_bias": (1, 7, 7, 256, 512, 3, 1, False, "linear"),
    "c768_1x1": (1, 6, 6, 768, 256, 1, 1, True, "leaky"),
    "c384_1x1": (1, 6, 6, 384, 128, 1, 1, True, "leaky"),
    "m_tail_3069": (3, 33, 31, 64, 64, 3, 1, True, "leaky"),
    "k_long_1280": (1, 13, 13, 1280, 256, 3, 1, True, "leaky"),
}


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("name", sorted(CONV_CASES))
def test_conv_mid_and_pool(name, dtype):
    """conv in the middle of a graph (T-typed vector stores) followed by a stride-1 max-pool."""
    B, H, W, cin, cout, k, s, bn, act = CONV_CASES[name]
    g = new_graph(H, W, cin)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, k, s, use_batch_normalization=bn, activation_fn=act))
    g.append(PL.max_pool2d(g[-1].out, 2, stride=1))
    x = synth.synthetic_input(B, H, W, cin, seed=3) * 2 - 1
    check_graph(g, x, dtype, seed=zlib.crc32(name.encode()) % 1000, read=(1,))


HEAD_CASES = {
    "head255": (2, 5, 5, 256, 255),
    "head425": (2, 13, 13, 1024, 425),
    "head125": (1, 13, 13, 1024, 125),
    "head18_tower": (1, 13, 13, 1024, 18),
    "head_vec_256": (2, 6, 6, 128, 256),
}


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("name", sorted(HEAD_CASES))
def test_conv_last_writes_float32(name, dtype):
    """the last conv (linear, bias, no BN) writes float32 logits straight to the caller's tensor"""
    B, H, W, cin, cout = HEAD_CASES[name]
    g = new_graph(H, W, cin)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 1, 1, use_batch_normalization=False, activation_fn="linear"))
    x = synth.synthetic_input(B, H, W, cin, seed=4) * 2 - 1
    check_graph(g, x, dtype, seed=7)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_residual_blocks_fused(dtype):
    g = new_graph(16, 16, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 2))
    for _ in range(2):
        g.append(PL.conv2d_bn_act(g[-1].out, 32, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 3))
        g.append(PL.shortcut(g[-1].out, g[-3].out))
    x = synth.synthetic_input(2, 16, 16, 3, seed=5)
    eng = check_graph(g, x, dtype, read=(2, 5))
    assert eng.num_kernels == 6              # convs only (the first reads the f32 input itself): the adds are epilogues
    assert "fused: +shortcut" in eng.describe()


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_upsample_concat_fused(dtype):
    g = new_graph(8, 8, 64)
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))                  # 1  skip source (writes a concat slice)
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 2))                 # 2  4x4
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))                  # 3
    g.append(PL.upsample(g[-1].out, 2))                              # 4  fused into 3
    g.append(PL.route([g[-1].out, g[1].out]))                        # 5  [up, skip] = 128 ch
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 1, 1))                  # 6
    x = synth.synthetic_input(2, 8, 8, 64, seed=6) * 2 - 1
    eng = check_graph(g, x, dtype, read=(4, 5))
    d = eng.describe()
    assert eng.num_kernels == 5 and "fused: upsample x2" in d and "concat slice" in d


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_reorg_concat_fused_v2_style(dtype):
    g = new_graph(8, 8, 64)
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))                  # 1  8x8x64
    g.append(PL.max_pool2d(g[-1].out, 2, 2))                         # 2  4x4
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))                 # 3  4x4x128 -> concat slice
    g.append(PL.route([g[1].out]))                                   # 4  alias of 1
    g.append(PL.conv2d_bn_act(g[-1].out, 16, 1, 1))                  # 5  8x8x16
    g.append(PL.reorg(g[-1].out, 2))                                 # 6  4x4x64, fused into 5
    g.append(PL.route([g[-1].out, g[3].out]))                        # 7  192 ch
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))                  # 8
    x = synth.synthetic_input(2, 8, 8, 64, seed=7) * 2 - 1
    eng = check_graph(g, x, dtype, read=(6, 7))
    assert "fused: reorg x2" in eng.describe()


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_standalone_fallback_kernels(dtype):
    """graphs where the fusions do not apply: generic strided element-wise kernels take over"""
    g = new_graph(8, 8, 16)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))                  # 1  two consumers -> shortcut not fusable
    g.append(PL.max_pool2d(g[-1].out, 2, 1))                         # 2
    g.append(PL.shortcut(g[1].out, g[2].out))                        # 3  standalone add
    g.append(PL.max_pool2d(g[-1].out, 2, 2))                         # 4  4x4x32
    g.append(PL.upsample(g[-1].out, 2))                              # 5  standalone upsample (of a pool)
    g.append(PL.reorg(g[3].out, 2))                                  # 6  standalone reorg 4x4x128
    g.append(PL.route([g[4].out, g[6].out]))                         # 7  4x4x160
    g.append(PL.route([g[7].out, g[4].out]))                         # 8  nested concat -> copies, 192 ch
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 1, 1))                  # 9
    g.append(PL.route([g[5].out, g[3].out]))                         # 10 8x8x64 last layer = route
    x = synth.synthetic_input(2, 8, 8, 16, seed=8) * 2 - 1
    check_graph(g, x, dtype, read=(3, 5, 6, 8, 9))


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_maxpool_odd_and_same(dtype):
    g = new_graph(7, 9, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))
    g.append(PL.max_pool2d(g[-1].out, 2, 2))                         # odd dims: zero pad takes part
    g.append(PL.max_pool2d(g[-1].out, 2, 1))
    g.append(PL.max_pool2d(g[-1].out, 2, 2))
    x = synth.synthetic_input(3, 7, 9, 3, seed=9)
    check_graph(g, x, dtype, read=(2, 3))


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_conv_maxpool_fused_in_2d_tap_tiles(dtype):
    """conv 3x3/1 -> max-pool 2x2/2 on wide maps (net/v2.py:18-37 pairs): the pool is taken in the conv's epilogue (2-D tap tiles:
    the two rows of a window are fragments of one lane, the two columns neighbouring lanes), the full-resolution tensor is never
    written.  Partial tiles in both directions, leaky values of both signs, 64 couts (fp16 + fp32) and 128 couts (fp16 only: no
    float32 128-cout 2-D tile, that pool stays a kernel)."""
    g = new_graph(12, 200, 32)
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))                  # 1
    g.append(PL.max_pool2d(g[-1].out, 2, 2))                         # 2  fused into 1 (tile 13)
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))                 # 3  6 x 100
    g.append(PL.max_pool2d(g[-1].out, 2, 2))                         # 4  fused into 3 for fp16 (tile 12)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 1, 1))                  # 5  3 x 50
    x = synth.synthetic_input(3, 12, 200, 32, seed=17)
    eng = check_graph(g, x, dtype, seed=6)
    d = eng.describe()
    assert d.count("fused 2x2/2 max-pool") == (2 if dtype == "fp16" else 1), d
    assert eng.num_kernels == (4 if dtype == "fp16" else 5)          # prep + convs (+ the pool fp32 keeps)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    assert "+pool" in names, names
    # the unfused plan (keep_all materialises every layer) gives the same values layer by layer
    check_graph(g, x, dtype, seed=6, read=(2, 4))


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_conv_maxpool_fused_32_cout_tile(dtype):
    """the 32-cout 2-D tap tile (tiny-YOLOv2's second conv, 16 -> 32 at 208 x 208 + max-pool): float32 with ONE 16-channel slice
    (K loop of nine taps), fp16 with one 32-channel slice; partial tiles, pool fused, the unfused plan gives the same values"""
    cin = 16 if dtype == "fp32" else 32
    g = new_graph(26, 104, cin)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))                  # 1
    g.append(PL.max_pool2d(g[-1].out, 2, 2))                         # 2  fused into 1 (tile 17)
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))                  # 3
    x = synth.synthetic_input(4, 26, 104, cin, seed=19)
    eng = check_graph(g, x, dtype, seed=8)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    assert "32x256,tap9,2d,x2>+pool" in names and eng.describe().count("fused 2x2/2 max-pool") == 1, names
    eng2 = check_graph(g, x, dtype, seed=8, read=(1, 2))
    assert "32x256,tap9,2d,x2>" in " ".join(ki.name.decode() for ki in eng2.kernel_infos())


@pytest.mark.parametrize("cout", [32, 16])
@pytest.mark.parametrize("hw", [(20, 44), (32, 64), (6, 130)])
def test_first_layer_pool_on_the_matrix_cores(hw, cout):
    """Darknet-19's (32 filters) / tiny-YOLOv2's (16) first layer + pool (conv 3x3/1 3 -> cout + BN + leaky + 2x2/2 max-pool) on the
    matrix cores: fp16 nets with 32 filters run first_pool_mfma_kernel, float32 nets first_pool_mfma_f32_kernel<cout / 16>
    (8 x 16 pooled-output tiles, partial tiles in both directions, image borders = zero padding, several images), against the
    oracle through a following conv; an fp16 net with 16 filters keeps the direct VALU kernel (same check)"""
    H, W = hw
    g = new_graph(H, W, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))
    g.append(PL.max_pool2d(g[-1].out, 2, 2))
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))
    x = synth.synthetic_input(3, H, W, 3, seed=23)
    for dtype in ("fp16", "fp32"):
        eng = check_graph(g, x, dtype, seed=2)
        names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
        syms = " ".join(ki.symbol.decode() for ki in eng.kernel_infos())
        assert "conv_first_pool" in names and eng.num_kernels == 2, names
        if dtype == "fp32":
            assert "first_pool_mfma_f32_kernel<%d>(" % (cout // 16) in syms, syms
        elif cout == 32:
            assert "yolo::first_pool_mfma_kernel(" in syms, syms
        else:
            assert "conv_first_kernel<false, 16, true>" in syms, syms


def test_variable_batch_below_max_batch():
    g = new_graph(12, 12, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 2))
    g.append(PL.conv2d_bn_act(g[-1].out, 18, 1, 1, use_batch_normalization=False, activation_fn="linear"))
    w = synth.darknet_stream(g, seed=1)
    x = synth.synthetic_input(4, 12, 12, 3, seed=10)
    from tensorflow_yolo_amd.net import engine
    eng = engine.HipNetwork(g, dtype="fp32", max_batch=4)
    eng.load_weights(w)
    full = eng.forward(x).cpu().numpy()
    for b in (1, 3):
        part = eng.forward(x[:b]).cpu().numpy()
        assert np.array_equal(part, full[:b])
    with pytest.raises(ValueError):
        eng.forward(np.concatenate([x, x]))


def test_errors_are_reported():
    from tensorflow_yolo_amd import _hip
    from tensorflow_yolo_amd.net import engine
    g = new_graph(8, 8, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 16, 3, 1))
    eng = engine.HipNetwork(g, dtype="fp16", max_batch=1)
    with pytest.raises(_hip.YoloHipError, match="weights not loaded"):
        eng.forward(synth.synthetic_input(1, 8, 8, 3))
    with pytest.raises(_hip.YoloHipError, match="weight stream holds"):
        eng.load_weights(np.zeros(5, np.float32))


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
@pytest.mark.parametrize("shape,cout", [((2, 32, 40), 32), ((1, 18, 300), 16), ((3, 6, 130), 32), ((2, 9, 20), 16)])
def test_first_conv_with_fused_pool(shape, cout, dtype):
    """Darknet-19 / tiny-YOLO head of the graph: first 3x3/1 conv (32 or 16 filters) + the 2x2/2 max-pool behind it in
    one kernel (x blocks of 128 with a tail, several images); an odd height keeps the two-kernel path"""
    B, H, W = shape
    g = new_graph(H, W, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))
    g.append(PL.max_pool2d(g[-1].out, 2, stride=2))
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))
    g.append(PL.max_pool2d(g[-1].out, 2, stride=1))
    x = synth.synthetic_input(B, H, W, 3, seed=41)
    eng = check_graph(g, x, dtype, seed=9)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    assert ("conv_first_pool" in names) == (H % 2 == 0 and W % 2 == 0), names


@pytest.mark.parametrize("shape", [(3, 40, 56), (2, 64, 64), (1, 18, 34), (2, 21, 30)])
def test_fused_stem(shape):
    """Darknet-53 stem (stem.hip): first 3x3/1 3->32 conv and the 3x3/2 32->64 conv behind it as one kernel; partial
    tiles in both directions, several images, image borders (zero padding of BOTH convs); an odd height keeps the
    two-kernel path"""
    B, H, W = shape
    g = new_graph(H, W, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 2))
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 1, 1))
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))
    g.append(PL.shortcut(g[-1].out, g[-3].out))
    g.append(PL.max_pool2d(g[-1].out, 2, stride=1))
    x = synth.synthetic_input(B, H, W, 3, seed=31)
    eng = check_graph(g, x, "fp16", seed=6)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    assert ("conv_stem<f16,3-32-64-32>" in names) == (H % 2 == 0 and W % 2 == 0), names       # incl. the 1x1 64->32 behind it
    eng32 = check_graph(g, x, "fp32", seed=6)
    assert "conv_stem" not in " ".join(ki.name.decode() for ki in eng32.kernel_infos())
    # a different third layer stays a kernel of its own: two-layer stem
    g2 = new_graph(H, W, 3)
    g2.append(PL.conv2d_bn_act(g2[-1].out, 32, 3, 1))
    g2.append(PL.conv2d_bn_act(g2[-1].out, 64, 3, 2))
    g2.append(PL.conv2d_bn_act(g2[-1].out, 64, 1, 1))
    g2.append(PL.max_pool2d(g2[-1].out, 2, stride=1))
    eng2 = check_graph(g2, x, "fp16", seed=8)
    names2 = " ".join(ki.name.decode() for ki in eng2.kernel_infos())
    assert ("conv_stem<f16,3-32-64>" in names2) == (H % 2 == 0 and W % 2 == 0) and "3-32-64-32" not in names2, names2


@pytest.mark.parametrize("tile", [8, 9, 10, 11, 12, 13, 15, 16, 17, 18])
def test_tap_reuse_tile_configs(tile):
    """tap-reuse tiles of conv_tap.hip (3x3/1 only: patch of 1, 2, 4 and 6 channel slices, image borders inside a
    block, position tail; the other layers fall back to the default choice) forced through yolo_net_options.force_tile: K-stage counts 1, 2 (shorter than the
    LDS ring), 4, 9 (odd), 18, 36, 54 and 72, residual, stride 2, M tails, Cout 255 head"""
    g = new_graph(21, 19, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))                  # 1  first-layer kernel
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 1, 1))                 # 2  Cin 32 1x1: 1 stage
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 2))                  # 3  (narrow: other kernel)
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 1, 1))                 # 4  Cin 64 1x1: 2 stages
    g.append(PL.conv2d_bn_act(g[-1].out, 192, 1, 1))                 # 5  Cin 128 1x1: 4 stages, Cout tail
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))                 # 6  Cin 192 3x3: 54 stages
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 1, 1))                  # 7
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))                 # 8  Cin 32 3x3: 9 stages
    g.append(PL.shortcut(g[-1].out, g[-3].out))                      # 9  fused residual
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))                  # 10
    g.append(PL.conv2d_bn_act(g[-1].out, 256, 3, 2))                 # 11 Cin 64 3x3/2: 18 stages
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 1, 1))                 # 12 8 stages
    g.append(PL.conv2d_bn_act(g[-1].out, 256, 3, 1))                 # 13 36 stages
    g.append(PL.shortcut(g[-1].out, g[-3].out))                      # 14
    g.append(PL.conv2d_bn_act(g[-1].out, 512, 3, 1))                 # 15 72 stages, two cout tiles
    g.append(PL.conv2d_bn_act(g[-1].out, 255, 1, 1, use_batch_normalization=False, activation_fn="linear"))   # 16 head
    x = synth.synthetic_input(5, 21, 19, 3, seed=12)
    eng = check_graph(g, x, "fp16", seed=4, read=(2, 4, 6, 9, 11, 14, 15), tile=tile)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    assert "tap9" in names, names


@pytest.mark.parametrize("shape", [(2, 19, 19, 512, 256), (2, 20, 21, 128, 256), (5, 13, 13, 128, 256), (3, 38, 38, 64, 128), (1, 76, 76, 128, 256), (2, 7, 78, 32, 128),
                                   (2, 5, 110, 64, 128), (1, 9, 152, 64, 128), (2, 33, 100, 32, 64), (1, 48, 304, 32, 64), (2, 21, 70, 32, 32)])
@pytest.mark.parametrize("tile", [8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 22])
def test_tap_reuse_conv_shapes(shape, tile):
    """conv_tap.hip on the feature-map sizes of YOLOv3-608 (19, 38, 76), the widest rows its padded-linear tiles take
    (78, 110, 158 >= 152), wide maps for the 2-D tiles (partial 16x16 tiles in both directions, Cout 64) and a residual
    input: blocks span image rows, images and the end of the batch"""
    B, H, W, cin, cout = shape
    g = new_graph(H, W, cin)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))
    g.append(PL.conv2d_bn_act(g[-1].out, cin, 1, 1))
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))
    g.append(PL.shortcut(g[-1].out, g[-3].out))
    g.append(PL.max_pool2d(g[-1].out, 2, stride=1))
    x = synth.synthetic_input(B, H, W, cin, seed=21)
    eng = check_graph(g, x, "fp16", seed=5, read=(1, 4), tile=tile)
    if tile in (11, 13, 17):    # the float32 tiles (16-channel slices, fp32 FMA chains restarted every 288 k: conv_common.h flush_acc): 1e-4 contract
        eng32 = check_graph(g, x, "fp32", seed=5, read=(1, 4), tile=tile)
        if dict(((11, cout > 64 and W <= 158), (13, cout == 64), (17, cout == 32)))[tile]:
            assert "tap9" in " ".join(ki.name.decode() for ki in eng32.kernel_infos())
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    max_w = {8: 78, 9: 78, 10: 109, 11: 158, 12: 1 << 20, 13: 1 << 20, 15: 19, 16: 1 << 20, 17: 1 << 20, 18: 19, 22: 13}[tile]     # (10, 15: whole planes of the position-interleaved patch)
    need = {8: cout > 64, 9: cout >= 256, 10: cout > 64, 11: cout > 64, 12: cout > 64, 13: cout == 64, 15: cout >= 256, 16: cout > 64, 17: cout == 32, 18: cout > 64 and H == W and W >= 18, 22: cout > 64 and H == W and W >= 12}[tile]
    if W <= max_w and need:             # else: the forced tile is not valid for this layer, the default one runs
        assert "tap9" in names, names


@pytest.mark.parametrize("shape", [(3, 38, 38, 256, 512), (2, 76, 76, 128, 256), (1, 152, 152, 128, 256), (2, 22, 150, 64, 128),
                                   (5, 12, 20, 32, 128), (2, 36, 36, 192, 384), (1, 34, 34, 128, 320)])
@pytest.mark.parametrize("tile", [20, 21])
def test_stride2_tap_reuse_tiles(shape, tile):
    """conv_tap.hip MODE 4: 3x3 / stride 2 as nine (parity plane, shift) taps over the padded-linear grid of the OUTPUT map
    (net/layers.py:17-30: explicit pad 1 + VALID, stride 2): Darknet-53's stage transitions (38 -> 19, 76 -> 38, 152 -> 76), a wide
    map (patch of 256 + 77 positions), six channel slices (the patch buffers swap roles per slice), Cout with a tail, one slice only; tiles that span image rows, images and the end of the batch.  Tile 21 (one whole output image per tile) is
    valid for 18 x 18 and 19 x 19 outputs only; elsewhere the default tile runs."""
    B, H, W, cin, cout = shape
    g = new_graph(H, W, cin)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 2))                # 1: the stride-2 conv
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))                  # 2
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))                # 3
    g.append(PL.shortcut(g[-1].out, g[1].out))                       # 4
    g.append(PL.max_pool2d(g[-1].out, 2, stride=1))
    x = synth.synthetic_input(B, H, W, cin, seed=31)
    eng = check_graph(g, x, "fp16", seed=9, read=(1, 4), tile=tile)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    if (tile == 20 and B * (H // 2) * (W // 2) >= 1024) or (H == W and 36 <= W <= 38):     # (tile 21: outputs of 18 x 18 and 19 x 19)
        assert ",s2," in names, names


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_random_layer_shapes_through_the_default_rules(dtype):
    """Seeded random conv stacks through the built-in tile rules (no forced tile): map sizes around the rule boundaries (13/14, 96,
    110, 152 columns), channel counts 16..1024, batches that put a launch on either side of the 128 / 256 / 512-workgroup limits,
    stride 2, residuals and pools -- every layer read back and compared with the oracle."""
    rng = np.random.RandomState(1234 if dtype == "fp16" else 4321)
    widths = [13, 14, 19, 26, 38, 52, 76, 96, 104, 112, 152]
    chans = [32, 64, 128, 256] if dtype == "fp16" else [16, 32, 64, 128]
    for case in range(10):
        W = int(widths[rng.randint(len(widths))])
        H = int(rng.randint(6, 20)) if W > 60 else W
        B = int(rng.randint(1, 7))
        cin = int(chans[rng.randint(len(chans))])
        c1 = int(chans[rng.randint(len(chans))])
        g = new_graph(H, W, cin)
        g.append(PL.conv2d_bn_act(g[-1].out, c1, 3, 1))                                  # 1
        g.append(PL.conv2d_bn_act(g[-1].out, max(16, c1 // 2) if dtype == "fp32" else max(32, c1 // 2), 1, 1))     # 2
        g.append(PL.conv2d_bn_act(g[-1].out, c1, 3, 1))                                  # 3
        g.append(PL.shortcut(g[-1].out, g[-3].out))                                      # 4
        read = [1, 4]
        if H % 2 == 0 and W % 2 == 0 and rng.randint(2):
            g.append(PL.max_pool2d(g[-1].out, 2, 2))                                     # 5
        else:
            g.append(PL.conv2d_bn_act(g[-1].out, min(1024, 2 * c1), 3, 2))               # 5
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))                                  # 6
        x = synth.synthetic_input(B, H, W, cin, seed=100 + case)
        print("case", case, (B, H, W, cin, c1))
        check_graph(g, x, dtype, seed=40 + case, read=tuple(read))


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 14, 19])
def test_every_dma_tile_config(tile):
    """each LDS-DMA tile shape of conv_dma.hip, forced through yolo_net_options.force_tile (a tile
    that is not valid for a layer falls back to the heuristic), on a graph with 3x3/1, 3x3/2, 1x1, residual,
    Cin=32 (one tap per K=32 stage), Cout=64 (weight tile smaller than the wave count) and M tails"""
    g = new_graph(20, 24, 3)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))                  # 1  first-layer kernel
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 2))                  # 2  Cin 32, Cout 64 (tile 7)
    g.append(PL.conv2d_bn_act(g[-1].out, 32, 1, 1))                  # 3
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))                  # 4
    g.append(PL.shortcut(g[-1].out, g[-3].out))                      # 5  fused residual
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 2))                 # 6  Cin 64
    g.append(PL.conv2d_bn_act(g[-1].out, 256, 3, 1))                 # 7  Cin 128 -> 256 (tiles 1-6)
    g.append(PL.conv2d_bn_act(g[-1].out, 128, 1, 1))                 # 8
    g.append(PL.conv2d_bn_act(g[-1].out, 256, 3, 1))                 # 9
    g.append(PL.shortcut(g[-1].out, g[-3].out))                      # 10
    g.append(PL.conv2d_bn_act(g[-1].out, 255, 1, 1, use_batch_normalization=False, activation_fn="linear"))   # 11 head
    x = synth.synthetic_input(3, 20, 24, 3, seed=11)
    eng = check_graph(g, x, "fp16", seed=3, read=(2, 5, 7, 10), tile=tile)
    names = " ".join(ki.name.decode() for ki in eng.kernel_infos())
    assert "conv_igemm_dma" in names


@pytest.mark.parametrize("case", ["residual_block_3x3", "stride2_into_stage", "stride2_into_stage_tap", "default_rules_152"])
def test_back_to_back_1x1_fusion(case, monkeypatch):
    """conv_common.h: conv_epilogue_fused_1x1 -- the 1x1 128 -> 64 conv behind a conv whose workgroups hold all 128 couts of their 256
    positions is computed by that launch (second MFMA pass over the epilogue's fp16 values through LDS), no launch of its own:
    Darknet-53's two sites at 152 x 152 (net/v3.py:16-19, 26-29): the residual block's 3x3 64 -> 128 + shortcut on the 2-D tap tile
    (partial tiles in both directions, several images) and the stride-2 conv 64 -> 128 into the stage on the LDS-DMA tile (M tail).
    Against the oracle, against the unfused plan of the same graph (YOLO_NO_FUSE2), and -- third case -- at the size where the
    default tile rules pick the fusing tile by themselves."""
    if case.startswith("stride2_into_stage"):
        # LDS-DMA tile 6 (what the rules pick) and the parity-plane tap tile 23 (round 4: same fusion, no faster, forced here)
        B, H, W, tile = (3, 44, 58, 6) if case == "stride2_into_stage" else (3, 44, 58, 23)
        g = new_graph(H, W, 64)
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 2))            # 1: the host (no residual)
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))             # 2: computed by 1
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))            # 3
        g.append(PL.shortcut(g[-1].out, g[1].out))                  # 4: y of layer 1 is still needed
        cin = 64
    else:
        B, H, W, tile = (3, 37, 50, 12) if case == "residual_block_3x3" else (12, 152, 152, None)
        g = new_graph(H, W, 32)
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))            # 1: x
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))             # 2
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))            # 3: the host ...
        g.append(PL.shortcut(g[-1].out, g[1].out))                  # 4: ... with its residual
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))             # 5: computed by 3
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))            # 6
        g.append(PL.shortcut(g[-1].out, g[4].out))                  # 7
        cin = 32
    x = synth.synthetic_input(B, H, W, cin, seed=55)
    eng = check_graph(g, x, "fp16", seed=14, tile=tile)
    names = [ki.name.decode() for ki in eng.kernel_infos()]
    syms = [ki.symbol.decode() for ki in eng.kernel_infos()]
    assert sum("+1x1" in n for n in names) == 1 and sum("fused into the conv in front" in n for n in names) == 1, names
    host = next(i for i, n in enumerate(names) if "+1x1" in n)
    assert "fused into the conv in front" in names[host + 1] and syms[host + 1] == ""
    assert ("conv_igemm_dma_kernel<2, 4, 4, 4, 3, 4, 4, true, 0>" if case == "stride2_into_stage" else
            "conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 4, 4, false, true, true>" if "stride2" in case else
            "conv3x3_tap_kernel<false, 2, 4, 4, 4, 27, 4, 2, false, true, true>") in syms[host], syms[host]
    fused = eng.forward(x).cpu().numpy()
    assert np.array_equal(fused, eng.forward(x).cpu().numpy())
    assert np.array_equal(fused[:1], eng.forward(x[:1]).cpu().numpy()[:1]) or tile is None      # (a forced tile also fuses at batch 1)
    monkeypatch.setenv("YOLO_NO_FUSE2", "1")
    plain, eng_p = run_hip(g, synth.darknet_stream(g, seed=14), x, "fp16", force_tile=tile)
    assert not any("+1x1" in ki.name.decode() for ki in eng_p.kernel_infos()) and eng_p.num_kernels == eng.num_kernels
    assert rel_err(fused, plain) <= 2e-3       # same products; only the K order of the 1x1's 128-deep sum differs


@pytest.mark.parametrize("dtype,shape", [("fp16", (24, 13, 13, 256, 512)), ("fp16", (20, 19, 19, 512, 256)),
                                         ("fp32", (24, 13, 13, 128, 512)), ("fp32", (24, 19, 19, 128, 256))])
def test_in_launch_pair_split_k(dtype, shape):
    """The K split INSIDE one launch (conv_tap.hip: two half-K workgroups per tile hand their accumulators over through write-through
    slabs and a ticket, the second arriver sums and runs the fused epilogue; api.cpp: pick_conv) at op level: 13 x 13 and 19 x 19
    maps at the batch that gives 64-128 tiles of 128 x 256 (fp16) or 129-256 tiles of 128 x 128 (float32: the only tile whose
    float32 pair instantiation exists).  Asserted: the pair kernel is what runs; the result against the oracle (with a residual, so
    the fused epilogue of the second arriver is the full one); bit-identical results on a second and third launch of the same engine
    (the ticket counters return to zero); and agreement with the whole-K kernel of the same tile (an explicitly forced tile is
    never overridden by the pair split)."""
    B, H, W, cin, cout = shape
    g = new_graph(H, W, cin)
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))            # 1: the pair launch
    g.append(PL.conv2d_bn_act(g[-1].out, cout, 3, 1))            # 2: ... and one with a residual behind it
    g.append(PL.shortcut(g[-1].out, g[1].out))
    x = synth.synthetic_input(B, H, W, cin, seed=77)
    eng = check_graph(g, x, dtype, seed=12)
    infos = eng.kernel_infos()
    names = [ki.name.decode() for ki in infos]
    syms = [ki.symbol.decode() for ki in infos]
    pair = [i for i, n in enumerate(names) if "+pairK" in n]
    assert len(pair) == 2, names
    # (13 x 13, fp16: the image-aligned 128 x 192 tile -- one image per tile, 24 images x 4 cout tiles x 2 halves)
    want_sym = ("conv3x3_tap_kernel<false, 2, 4, 4, 3, 14, 2, 1, true, false, false>" if dtype == "fp16" and H == 13 else
                "conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 2, 1, true, false, false>" if dtype == "fp16" else "conv3x3_tap_kernel<true, 2, 4, 4, 2, 28, 4, 1, true, false, false>")
    assert all(want_sym in syms[i] for i in pair), syms
    a = eng.forward(x).cpu().numpy()
    b = eng.forward(x).cpu().numpy()
    c = eng.forward(x[:B]).cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(a, c), "the pair launch is not repeatable: a ticket counter did not return to zero"
    tile = (22 if H == 13 else 8) if dtype == "fp16" else 11
    whole, eng_w = run_hip(g, synth.darknet_stream(g, seed=12), x, dtype, force_tile=tile)
    assert not any("+pairK" in ki.name.decode() for ki in eng_w.kernel_infos()), "a forced tile must run as it is named"
    assert rel_err(a, whole) <= (2e-3 if dtype == "fp16" else 2e-6)        # same products, K summed in two halves


def test_float32_products_as_nine_bf16_products():
    """conv.hip: conv_igemm_emu_kernel -- a float32 conv on the bf16 matrix cores: every operand split exactly into three bf16 values
    while it is staged, every product as nine bf16 x bf16 MFMA terms accumulated in fp32, small terms first, two-level accumulation
    every 256 k (tf.layers.conv2d in float32, net/layers.py:31-39).  yolo_net_options.f32_products: 0 = the library's rule (whole-K
    launches of the 4-wave kernel with K >= 4608 over >= 256 workgroups: the tiny-YOLOv2 13 x 13 layers), 1 = native float32 MFMA
    everywhere, 2 = wherever the kernel applies.  13 x 13 maps with K = 4608 and 9216, a 1x1, a stride-2 3x3 and a residual, at the
    batch where the 4-wave kernel runs whole K; every layer against the oracle at the float32 tolerance of every other test, in all
    three modes, and the kernels named must be the ones the mode asks for."""
    g = new_graph(13, 13, 512)
    g.append(PL.conv2d_bn_act(g[-1].out, 1024, 3, 1))                # 1  K = 4608: by rule
    g.append(PL.conv2d_bn_act(g[-1].out, 256, 1, 1))                 # 2  K = 1024: only when forced
    g.append(PL.conv2d_bn_act(g[-1].out, 1024, 3, 1))                # 3  K = 2304: only when forced
    g.append(PL.shortcut(g[-1].out, g[1].out))                       # 4
    g.append(PL.conv2d_bn_act(g[-1].out, 1024, 3, 1))                # 5  K = 9216: by rule
    g.append(PL.conv2d_bn_act(g[-1].out, 512, 3, 2))                 # 6
    g.append(PL.max_pool2d(g[-1].out, 2, stride=1))                  # 7
    x = synth.synthetic_input(40, 13, 13, 512, seed=91)
    counts = {}
    for mode in (0, 1, 2):
        eng = check_graph(g, x, "fp32", seed=17, read=(1, 4, 5, 6), tile=0, f32_products=mode)    # tile 0 = the 4-wave kernel (whole K at this batch)
        counts[mode] = sum("conv_igemm_emu" in ki.name.decode() for ki in eng.kernel_infos())
        syms = [ki.symbol.decode() for ki in eng.kernel_infos() if "conv_igemm_emu" in ki.name.decode()]
        assert all(s == "void yolo::conv_igemm_emu_kernel<2, 4, 4, 2>(yolo::ConvParams)" for s in syms), syms
    assert counts[1] == 0 and counts[0] == 2 and counts[2] > counts[0], counts


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("batch", [1, 2])
def test_split_k_small_maps(dtype, batch):
    """13 x 13 / 7 x 9 maps at batch 1-2: a launch is a handful of tiles with K in the thousands, so the K range is split over
    workgroups (conv_tap.hip channel slices, conv.hip K tiles) and the float32 partials are summed by splitk_reduce_kernel (4-wave
    kernel) or by the tile's last arriver inside the launch (tap tile), which then runs the epilogue: bias + leaky, fused residual, reorg and upsample output maps, Cout tails (255, 320), float32 final layer"""
    g = new_graph(13, 13, 64)
    g.append(PL.conv2d_bn_act(g[-1].out, 512, 3, 1))                 # 1  3x3 Cin 64
    g.append(PL.conv2d_bn_act(g[-1].out, 1024, 3, 1))                # 2  3x3 Cin 512: K = 4608
    g.append(PL.conv2d_bn_act(g[-1].out, 512, 1, 1))                 # 3  1x1 Cin 1024
    g.append(PL.conv2d_bn_act(g[-1].out, 1024, 3, 1))                # 4  3x3 + residual
    g.append(PL.shortcut(g[-1].out, g[-3].out))                      # 5
    g.append(PL.conv2d_bn_act(g[-1].out, 320, 3, 2))                 # 6  3x3/2 (4-wave kernel), Cout tail, 7x7
    g.append(PL.conv2d_bn_act(g[-1].out, 256, 1, 1))                 # 7
    g.append(PL.upsample(g[-1].out, 2))                              # 8  fused upsample -> 14x14
    g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))                  # 9
    g.append(PL.reorg(g[-1].out, 2))                                 # 10 fused reorg -> 7x7x256
    g.append(PL.conv2d_bn_act(g[-1].out, 1024, 3, 1))                # 11
    g.append(PL.conv2d_bn_act(g[-1].out, 255, 1, 1, use_batch_normalization=False, activation_fn="linear"))   # 12 float32 out
    x = synth.synthetic_input(batch, 13, 13, 64, seed=31)
    eng = check_graph(g, x, dtype, seed=9, read=(2, 5, 6, 8, 10, 11))
    # round 4: the 3x3 / 1 layers on the tap tile sum their splits INSIDE the launch (the last arriver of a tile adds every split's slab
    # in split order: no reduce launch, and the same bits whoever arrives last)
    names = [ki.name.decode() for ki in eng.kernel_infos()]
    assert sum(",1launch" in n or "+pairK" in n for n in names) >= 2, names
    a = eng.forward(x).cpu().numpy()
    for _ in range(3):
        assert np.array_equal(a, eng.forward(x).cpu().numpy()), "in-launch split-K is not repeatable (summation order or ticket counter)"


# ---- guard-band canaries -----------------------------------------------------------------------------
def _guarded_run(net, w, x, dtype, tile=None, keep_all=True, detect=False, guard=4096):
    """Plan with `guard` never-used bytes behind every tensor, fill the WHOLE workspace with a pattern, run, and require every byte no
    plan region claims as payload to still hold the pattern: the slack + guard behind each tensor / candidate list / counter block /
    scratch slab (yolo_net_workspace_regions).  With keep_all no two tensors share bytes; without it (the fused production plan:
    lifetime-packed arenas) only slack that no other region's payload overlaps can be checked."""
    import torch
    from tensorflow_yolo_amd import _hip
    from tensorflow_yolo_amd.net import engine
    eng = engine.HipNetwork(net, dtype=dtype, max_batch=x.shape[0], keep_all=keep_all, force_tile=tile, guard_bytes=guard)
    eng.load_weights(w)
    ws = eng._workspace
    ws.fill_(0xA5)
    _hip.check(eng.lib.yolo_net_bind_workspace(eng.handle, ws.data_ptr(), ws.numel()), "yolo_net_bind_workspace")   # (zeroes the tickets)
    eng.forward(x)
    if detect:
        eng.detect(x, 0.3, 0.6)
    torch.cuda.synchronize()
    regions = eng.workspace_regions()
    payload = sorted((off, off + used) for _, off, used, _ in regions if used)
    checked = 0
    for name, off, used, region in regions:
        lo, hi = off + used, off + region
        if hi <= lo:
            continue
        # cut out what another region's payload covers (lifetime-packed arenas)
        spans, cur = [], lo
        for a, b in payload:
            if b <= cur or a >= hi:
                continue
            if a > cur:
                spans.append((cur, a))
            cur = max(cur, b)
            if cur >= hi:
                break
        if cur < hi:
            spans.append((cur, hi))
        for a, b in spans:
            bad = (ws[a:b] != 0xA5).nonzero()
            assert bad.numel() == 0, "%s (tile %s, %s): byte %d behind the payload of a %d-byte region was written\n%s" % (
                name, tile, dtype, int(bad[0]) + a - lo, used, eng.describe())
            checked += b - a
    assert checked >= guard, "nothing to check"
    return eng, checked


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
@pytest.mark.parametrize("which", ["v2-416", "v3-160", "tiny-v2-416"])
def test_no_kernel_writes_outside_its_tensor_whole_nets(which, dtype):
    """every kernel of the three network families (reference net/v2.py:18-59, net/v3.py:22-93; tiny-YOLOv2 from the same vocabulary), both
    dtypes, forward + detect: with keep_all (one region per tensor, generic kernels) and as the production plan (fused stem, pools,
    back-to-back 1x1, lifetime-packed arena)"""
    from oracle import cases
    from tensorflow_yolo_amd.net import v2, v3
    names = ["c%d" % i for i in range(80)]
    if which == "v3-160":
        net = v3.create_network(np.reshape(cases.COCO_V3_ANCHORS, [-1, 2]), names, False, input_shape=(160, 160, 3))
        size, batch, nc = 160, 3, 80
    elif which == "v2-416":
        net = v2.create_full_network(np.reshape(cases.COCO_V2_ANCHORS, [-1, 2]), names, False, input_shape=(416, 416, 3))
        size, batch, nc = 416, 2, 80
    else:
        net = v2.create_tiny_network(np.reshape(cases.VOC_TINY_ANCHORS, [-1, 2]), names[:20], False, input_shape=(416, 416, 3))
        size, batch, nc = 416, 2, 20
    w = synth.darknet_stream(net, seed=31, num_classes=nc, obj_bias=-1.0)
    x = synth.synthetic_input(batch, size, size, 3, seed=32)
    total = 0
    for keep_all in (True, False):
        eng, checked = _guarded_run(net, w, x, dtype, keep_all=keep_all, detect=(which == "v3-160"))
        total += checked
    print("%s %s: %d guard / slack bytes intact" % (which, dtype, total))


@pytest.mark.parametrize("tile", list(range(0, 24)))
def test_no_kernel_writes_outside_its_tensor_every_tile(tile):
    """every conv tile id (0 = the 4-wave kernel, 1-7 / 14 / 19 LDS-DMA tiles, 8-13 / 15-18 / 22 tap reuse, 20 / 21 / 23 stride-2 tap reuse),
    forced wherever valid, on graphs whose maps leave position / pixel / cout tails (odd sizes, 255 head channels, 19 x 19 and 13 x 13 maps
    for the image-aligned tiles): fp16, and float32 where the tile has a float32 instantiation"""
    shapes = [(3, 20, 24), (2, 19, 19), (2, 13, 13), (1, 38, 38)]
    total = 0
    for B, H, W in shapes:
        g = new_graph(H, W, 3)
        g.append(PL.conv2d_bn_act(g[-1].out, 32, 3, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 3, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 64, 1, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 3, 1))
        g.append(PL.shortcut(g[-1].out, g[-3].out))
        g.append(PL.conv2d_bn_act(g[-1].out, 256, 3, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 128, 1, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 256, 3, 1))
        g.append(PL.shortcut(g[-1].out, g[-3].out))
        if H % 2 == 0 and W % 2 == 0:
            g.append(PL.conv2d_bn_act(g[-1].out, 512, 3, 2))
            g.append(PL.conv2d_bn_act(g[-1].out, 256, 1, 1))
        g.append(PL.conv2d_bn_act(g[-1].out, 255, 1, 1, use_batch_normalization=False, activation_fn="linear"))
        w = synth.darknet_stream(g, seed=41)
        x = synth.synthetic_input(B, H, W, 3, seed=42)
        for dtype in ("fp16", "fp32"):
            _, checked = _guarded_run(g, w, x, dtype, tile=tile, keep_all=True)
            total += checked
    print("tile %d: %d guard / slack bytes intact" % (tile, total))
