#!/usr/bin/env python3
"""Experiment: how much does a start offset between the two workgroups of a CU cost the tap kernel?  (profiles/r05_ablation.md section 15)
Experiment build only: the workgroups with an odd threadgroup id start (YOLO_CONV_DBG >> 8) x 0.25 us late; block trace of the 76 x 76 and 38 x 38
3x3 launches of one YOLOv3-608 batch-32 forward per offset."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from tensorflow_yolo_amd.net import synth
kind, size, batch, dtype = bench.WORKLOADS["v3-608-b32-fp16"]
model, w, anchors, ncls = bench.make_model(kind, size, batch, dtype, streams=1)
eng = model.net.engine
x = torch.from_numpy(synth.synthetic_input(batch, size, size, 3, seed=1)).cuda()
for _ in range(2): eng.forward(x)
torch.cuda.synchronize()
path = "/tmp/offset_trace.bin"
for units in [0, 1, 2, 4, 8, 20, 40, 80]:
    if os.path.exists(path): os.remove(path)
    os.environ["YOLO_CONV_TRACE"] = path
    os.environ["YOLO_CONV_DBG"] = str((units << 8) | int(os.environ.get("PROBE_DBG_BITS", "0")))
    eng.forward(x); torch.cuda.synchronize()
    del os.environ["YOLO_CONV_TRACE"]
    raw = np.fromfile(path, dtype=np.uint64); pos = 0; seen = set()
    while pos < len(raw):
        hdr = raw[pos:pos + 8]; pos += 8
        nb = int(hdr[0]); rec = raw[pos:pos + nb * 8].reshape(nb, 8); pos += nb * 8
        M, cout, cpt, H, W, cfg, ks = (int(v) for v in hdr[1:8])
        if cfg != 8 or ks != 31 or (H, cout) in seen or H not in (76, 38): continue
        seen.add((H, cout))
        t = rec[:, :4].astype(np.int64); t = (t - t[:, 0].min()) / 100.0
        odd = ((rec[:, 4] >> 16) & 1).astype(bool)
        first = t[:, 0] < 3.0
        d = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]], 1)
        print('   phases (setup, prologue+K, epilogue) mean:', d[first].mean(0).round(2), 'first round;', d[~first].mean(0).round(2) if (~first).any() else '', 'later')
        kl = t[:, 2] - t[:, 1]
        print("offset %5.2f us  %dx%d: span %6.1f us | first round K loop: even tg %6.2f us (%d wgs), odd tg %6.2f us (%d) | later rounds %6.2f us" % (
            units * 0.25, H, W, t[:, 3].max(), kl[first & ~odd].mean(), (first & ~odd).sum(), kl[first & odd].mean(), (first & odd).sum(), kl[~first].mean()))
