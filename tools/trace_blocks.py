#!/usr/bin/env python3
"""Experiment: per-block phase timeline of the tap-reuse conv (YOLO_CONV_TRACE).  Runs the v3-608 b32 forward once
with tracing, then prints, for each traced layer shape (last launch of it): phase durations, how many workgroups are
in which phase over time, and how the two workgroups of a CU overlap."""
import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
path = "/tmp/yolo_conv_trace.bin"
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    if os.path.exists(path):
        os.remove(path)
    import torch
    from tensorflow_yolo_amd import YoloV3
    from tensorflow_yolo_amd.net import synth
    import bench
    names = ["c%d" % i for i in range(80)]
    model = YoloV3()
    net = YoloV3.create_network(np.reshape(bench.COCO_V3, [-1, 2]), names, False, input_shape=(608, 608, 3))
    w = synth.darknet_stream(net, seed=0, num_classes=80)
    model.build(bench.COCO_V3, names, (608, 608, 3), dtype="fp16", max_batch=32, weights=w)
    eng = model.net.engine
    x = torch.from_numpy(synth.synthetic_input(32, 608, 608, 3, seed=1)).cuda()
    for _ in range(2):
        eng.forward(x)
    torch.cuda.synchronize()
    os.environ["YOLO_CONV_TRACE"] = path
    eng.forward(x)
    torch.cuda.synchronize()
    del os.environ["YOLO_CONV_TRACE"]
raw = np.fromfile(path, dtype=np.uint64)
pos = 0
layers = {}
while pos < len(raw):
    hdr = raw[pos:pos + 8]; pos += 8
    nb = int(hdr[0])
    rec = raw[pos:pos + nb * 8].reshape(nb, 8); pos += nb * 8
    layers[tuple(int(v) for v in hdr[1:8])] = rec
TAP_CFGS = (8, 9, 10, 11, 12, 13, 15)
ONLY = os.environ.get("TRACE_ONLY", "")           # e.g. "k1" (1x1 layers), "s2" (stride-2 layers)
for key, rec in layers.items():
    M, cout, cpt, H, W, cfg, ks = key
    if ONLY == "k1" and ks // 10 != 1: continue
    if ONLY == "s2" and ks % 10 != 2: continue
    if ONLY == "tap" and cfg not in TAP_CFGS: continue
    t = rec[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) / 100.0                      # us (100 MHz)
    hw = rec[:, 4]; xcc = rec[:, 5] & 0xf
    cu = ((hw >> 8) & 0xf).astype(int); se = ((hw >> 13) & 0x7).astype(int); sh = ((hw >> 12) & 1).astype(int)
    cuid = (xcc.astype(int) * 8 + se) * 32 + sh * 16 + cu
    print("== layer %dx%d k%d/s%d cin %d -> %d, tile %d: %d blocks on %d CUs, span %.1f us" % (H, W, ks // 10, ks % 10, cpt * 8, cout, cfg, len(rec), len(set(cuid)), t[:, 3].max()))
    if cfg not in TAP_CFGS and ks:
        f = (rec[:, 6].astype(np.int64) - rec[:, 1].astype(np.int64)) / 100.0
        print("   first K tile landed (after setup) mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (f.mean(), *np.percentile(f, [10, 50, 90])))
    cyc = rec[:, 7].astype(np.float64)
    print("   shader clock during the blocks: %.0f MHz (median of cycles / wall time)" % np.median(cyc / ((rec[:, 3] - rec[:, 0]).astype(np.float64) / 100.0)))
    d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]], 1)
    for i, nm in enumerate(["setup", "prologue+K loop", "epilogue"]):
        print("   %-16s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (nm, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
    # first round (workgroups resident from the start: two per CU) vs back-filled ones; the tail = the part of the span after the
    # last workgroup of the first kind has finished
    first = t[:, 0] < 2.0
    if first.any() and (~first).any():
        kl = t[:, 2] - t[:, 1]
        t_first_done = np.percentile(t[first, 3], 50)
        alone = t[:, 1] > np.percentile(t[first, 3], 90)          # started after (almost) every first-round workgroup had ended
        print("   K loop: first round (%d wgs) mean %.2f us, back-filled (%d) mean %.2f us, started after the first round had ended (%d) mean %.2f us"
              % (first.sum(), kl[first].mean(), (~first).sum(), kl[~first].mean(), alone.sum(), kl[alone].mean() if alone.any() else float("nan")))
        work = kl.sum()
        print("   sum of K-loop time %.0f us over %d CUs = %.1f us per CU at the rates measured; span %.1f us; median end of the first round %.1f us"
              % (work, len(set(cuid)), work / len(set(cuid)) / 2.0, T if False else t[:, 3].max(), t_first_done))
    # occupancy over time
    T = t[:, 3].max()
    grid = np.linspace(0, T, int(os.environ.get("TRACE_BINS", "21")))
    print("   time(us)  in-setup  in-loop  in-epilogue  (workgroups)")
    for g in grid[:-1]:
        a = ((t[:, 0] <= g) & (g < t[:, 1])).sum(); b = ((t[:, 1] <= g) & (g < t[:, 2])).sum(); c = ((t[:, 2] <= g) & (g < t[:, 3])).sum()
        print("   %7.1f  %8d %8d %8d" % (g, a, b, c))
    # per-CU: fraction of time with 2, 1, 0 workgroups in the K loop
    tot = np.zeros(3)
    for c in set(cuid):
        idx = np.where(cuid == c)[0]
        ev = sorted([(t[i, 1], 1) for i in idx] + [(t[i, 2], -1) for i in idx])
        cur = 0; last = 0.0
        for tm, dl in ev:
            tot[min(cur, 2)] += tm - last; last = tm; cur += dl
        tot[0] += T - last
    print("   per-CU time with 0 / 1 / 2 workgroups inside the K loop: %.1f%% / %.1f%% / %.1f%%" % tuple(100 * tot / tot.sum()))
