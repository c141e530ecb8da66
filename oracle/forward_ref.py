"""Oracle (test infrastructure): CPU restatement of the reference's TF conv-stack forward.

ARITHMETIC PARITY UNPINNED (see oracle/__init__.py): TensorFlow is absent and the reference
holds no golden logits.  The TOPOLOGY and the Darknet weight order this module walks ARE pinned to the
reference's own builders (tests/golden/topology_*.json, oracle/gen_topology.py).  Every function cites the
reference call site it follows.

All tensors are NHWC like the reference's graph (net/layers.py:108); torch-CPU is
used for the convolution arithmetic only.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import topology

_BN_EPS = 1e-5      # net/layers.py:5
_LEAKY = 0.1        # net/layers.py:6


def parse_darknet_weights(L, flat):
    """Slice a flat float32 Darknet stream into per-conv dicts.

    Order per conv (net/layers.py:53-63): beta, gamma, moving_mean, moving_variance,
    kernel for BN layers; bias, kernel otherwise.  Kernel is stored [out,in,kh,kw]
    (net/base.py:36-40).  Stricter than the reference: asserts the stream is used up
    (net/base.py:44 only prints the counts).
    """
    S = topology.shapes(L)
    flat = np.asarray(flat, dtype=np.float32)
    out = {}
    pos = 0

    def take(n):
        nonlocal pos
        v = flat[pos:pos + n]
        assert v.size == n, "weight stream too short"
        pos += n
        return v

    for i, op in enumerate(L):
        if op[0] != "conv":
            continue
        _, src, f, k, s, bn, act = op
        cin = S[src][2]
        d = {}
        if bn:
            d["beta"], d["gamma"], d["mean"], d["var"] = take(f), take(f), take(f), take(f)
        else:
            d["bias"] = take(f)
        d["kernel_oihw"] = take(f * cin * k * k).reshape(f, cin, k, k)
        out[i] = d
    assert pos == flat.size, "weight stream has %d unread values" % (flat.size - pos)
    return out


def _pad_hw(x, k):
    """net/layers.py:9-14: zero pad (k-1)//2 before, the rest after, on H and W."""
    tot = k - 1
    a = tot // 2
    b = tot - a
    return F.pad(x, (a, b, a, b))           # NCHW: (w_left, w_right, h_top, h_bottom)


def _q16(t):
    """round to fp16 storage and back (emulates the HIP path's fp16 tensors)"""
    return t.to(torch.float16).to(t.dtype)


def _conv_folded_fp16(x, wd, k, s, bn, act, dtype, head):
    """Variant used only to get a TIGHT bound on the fp16 HIP path: same operator semantics as
    _conv, but with the storage roundings the device performs (BN folded in float64 and the folded
    kernel rounded to fp16; bias kept in fp32; output rounded to fp16 unless it is a head conv)."""
    w = torch.from_numpy(np.ascontiguousarray(wd["kernel_oihw"])).to(torch.float64)
    if bn:
        scale = torch.from_numpy(wd["gamma"]).double() / torch.sqrt(torch.from_numpy(wd["var"]).double() + _BN_EPS)
        bias = torch.from_numpy(wd["beta"]).double() - torch.from_numpy(wd["mean"]).double() * scale
        w = w * scale.view(-1, 1, 1, 1)
    else:
        bias = torch.from_numpy(wd["bias"]).double()
    w = w.float().to(torch.float16).to(dtype)
    bias = bias.float().to(dtype).view(1, -1, 1, 1)
    if s > 1:
        y = F.conv2d(_pad_hw(x, k), w, None, stride=s, padding=0)
    else:
        y = F.conv2d(x, w, None, stride=1, padding=(k - 1) // 2)
    y = y + bias
    if act == "leaky":
        y = torch.maximum(_LEAKY * y, y)
    return y            # (`head`: the caller keeps head-conv outputs in float32, everything else is rounded to fp16 there)


def _conv(x, wd, k, s, bn, act, dtype):
    """net/layers.py:17-67.  x: NCHW torch tensor."""
    w = torch.from_numpy(np.ascontiguousarray(wd["kernel_oihw"])).to(dtype)
    if s > 1:
        x = _pad_hw(x, k)                   # :28-29 explicit pad, then VALID
        y = F.conv2d(x, w, None, stride=s, padding=0)
    else:
        y = F.conv2d(x, w, None, stride=1, padding=(k - 1) // 2)    # SAME, odd k
    if bn:                                  # :41-48, inference statistics
        g = torch.from_numpy(wd["gamma"]).to(dtype).view(1, -1, 1, 1)
        b = torch.from_numpy(wd["beta"]).to(dtype).view(1, -1, 1, 1)
        m = torch.from_numpy(wd["mean"]).to(dtype).view(1, -1, 1, 1)
        v = torch.from_numpy(wd["var"]).to(dtype).view(1, -1, 1, 1)
        y = g * (y - m) / torch.sqrt(v + _BN_EPS) + b
    else:
        y = y + torch.from_numpy(wd["bias"]).to(dtype).view(1, -1, 1, 1)     # :37 use_bias
    if act == "leaky":                      # :50-51
        y = torch.maximum(_LEAKY * y, y)
    return y


def _maxpool(x, k, s):
    """net/layers.py:70-81.  s>1: zero pad (0 before, k-1 after) then VALID;
    s==1: TF SAME (window clipped at the edge, padding ignored)."""
    if s > 1:
        x = _pad_hw(x, k)                   # zeros take part in the max (matters for odd H)
        return F.max_pool2d(x, k, s)
    # SAME, stride 1, k=2: pad after with -inf so that padding never wins
    x = F.pad(x, (0, k - 1, 0, k - 1), value=float("-inf"))
    return F.max_pool2d(x, k, 1)


def _reorg(x, s):
    """net/layers.py:90-97: extract_image_patches k=s -> block-major space-to-depth:
    out[n, (di*s+dj)*C + c, i, j] = in[n, c, s*i+di, s*j+dj]."""
    n, c, h, w = x.shape
    x = x.view(n, c, h // s, s, w // s, s)          # n c i di j dj
    x = x.permute(0, 3, 5, 1, 2, 4).contiguous()    # n di dj c i j
    return x.view(n, s * s * c, h // s, w // s)


def forward(L, weights, x_nhwc, dtype=torch.float32, keep=None, threads=None, storage=None):
    """Run the layer list; returns the final output as a NumPy array in the reference's
    layout: v2 [B,h,w,A*(5+C)] (net/v2.py:52-59); v3 [B,sum(h*w*3),5+C] (net/layers.py:119-133).

    keep: optional set of layer indices whose NHWC outputs are returned as a dict too.
    storage: None (the reference's arithmetic) or "fp16": additionally apply the roundings of the
    HIP fp16 path (fp16 input/activations/folded kernels; a fused shortcut adds in fp32 before the
    single rounding) so that path can be bounded tightly.
    """
    if threads:
        torch.set_num_threads(threads)
    S = topology.shapes(L)
    if isinstance(weights, np.ndarray):
        weights = parse_darknet_weights(L, weights)
    outs = [None] * len(L)
    kept = {}
    last_use = {}
    q16 = storage == "fp16"
    # convs whose output feeds ONLY a shortcut are rounded after the add (fused epilogue)
    fused_conv = set()
    head_conv = set()
    if q16:
        cons = {}
        for i, op in enumerate(L):
            srcs_ = [op[1]] if op[0] in ("conv", "maxpool", "reorg", "upsample", "yolo") else \
                ([op[1], op[2]] if op[0] == "shortcut" else (list(op[1]) if op[0] in ("route", "detection") else []))
            for s_ in srcs_:
                cons.setdefault(s_, []).append(i)
        for i, op in enumerate(L):
            if op[0] == "conv":
                c_ = cons.get(i, [])
                if len(c_) == 1 and L[c_[0]][0] == "shortcut":
                    fused_conv.add(i)
                if i == len(L) - 1 or (len(c_) == 1 and L[c_[0]][0] == "yolo"):
                    head_conv.add(i)
    for i, op in enumerate(L):
        srcs = []
        if op[0] in ("conv", "maxpool", "reorg", "upsample", "yolo"):
            srcs = [op[1]]
        elif op[0] == "shortcut":
            srcs = [op[1], op[2]]
        elif op[0] in ("route", "detection"):
            srcs = list(op[1])
        for s_ in srcs:
            last_use[s_] = i
    with torch.no_grad():
        for i, op in enumerate(L):
            k = op[0]
            if k == "input":
                x = torch.from_numpy(np.ascontiguousarray(np.asarray(x_nhwc, dtype=np.float32)))
                y = x.permute(0, 3, 1, 2).contiguous().to(dtype)     # placeholder is float32 (layers.py:108)
                if q16:
                    y = _q16(y)
            elif k == "conv":
                if q16:
                    y = _conv_folded_fp16(outs[op[1]], weights[i], op[3], op[4], op[5], op[6], dtype, i in head_conv)
                    if i not in fused_conv and i not in head_conv:
                        y = _q16(y)
                else:
                    y = _conv(outs[op[1]], weights[i], op[3], op[4], op[5], op[6], dtype)
            elif k == "maxpool":
                y = _maxpool(outs[op[1]], op[2], op[3])
            elif k == "route":                      # layers.py:84-87, concat in list order
                y = outs[op[1][0]] if len(op[1]) == 1 else torch.cat([outs[j] for j in op[1]], dim=1)
            elif k == "reorg":
                y = _reorg(outs[op[1]], op[2])
            elif k == "shortcut":                   # layers.py:100-103, no activation after
                y = outs[op[1]] + outs[op[2]]
                if q16:
                    y = _q16(y)
            elif k == "upsample":                   # layers.py:112-116 nearest, out[i,j]=in[i//s,j//s]
                y = outs[op[1]].repeat_interleave(op[2], dim=2).repeat_interleave(op[2], dim=3)
            elif k == "yolo":                       # layers.py:133 reshape [B,h*w*b,5+C]
                t = outs[op[1]].permute(0, 2, 3, 1).contiguous()
                nb = len(op[2])
                y = t.view(t.shape[0], t.shape[1] * t.shape[2] * nb, t.shape[3] // nb)
            elif k == "detection":                  # layers.py:122 concat on axis 1
                y = torch.cat([outs[j] for j in op[1]], dim=1)
            else:
                raise ValueError(k)
            outs[i] = y
            if keep is not None and i in keep:
                kept[i] = (y.permute(0, 2, 3, 1) if (y.dim() == 4) else y).contiguous().to(torch.float32).numpy()
            for j, lu in list(last_use.items()):    # drop activations nobody reads any more
                if lu == i:
                    outs[j] = None
                    del last_use[j]
    y = outs[-1]
    if y.dim() == 4:
        y = y.permute(0, 2, 3, 1).contiguous()
    res = y.to(torch.float32).numpy()
    return (res, kept) if keep is not None else res


def conv_naive_numpy(x_nhwc, kernel_oihw, stride, pad_before, pad_after):
    """Independent direct convolution in float64 NumPy loops (small shapes only); used to
    cross-check the torch path above (two independent ways)."""
    x = np.asarray(x_nhwc, np.float64)
    w = np.asarray(kernel_oihw, np.float64)
    n, h, wd, c = x.shape
    f, _, kh, kw = w.shape
    xp = np.zeros((n, h + pad_before + pad_after, wd + pad_before + pad_after, c))
    xp[:, pad_before:pad_before + h, pad_before:pad_before + wd, :] = x
    ho = (xp.shape[1] - kh) // stride + 1
    wo = (xp.shape[2] - kw) // stride + 1
    y = np.zeros((n, ho, wo, f))
    for i in range(ho):
        for j in range(wo):
            patch = xp[:, i * stride:i * stride + kh, j * stride:j * stride + kw, :]   # n kh kw c
            y[:, i, j, :] = np.einsum("nhwc,fchw->nf", patch, w)
    return y
