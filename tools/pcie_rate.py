#!/usr/bin/env python3
"""PCIe-inclusive rate of the product API: Yolo.predict() on HOST batches -- float32 / float64 NumPy arrays as the
reference's test loop holds them (net/base.py:153) -- including the host cast, the H2D copy, the records' D2H copy and the
BoundingBox lists; next to the device-resident detect() rate bench.py reports.  YOLOv3-608 batch 32 fp16."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tensorflow_yolo_amd.net import synth

model, w, anchors, ncls = bench.make_model("v3", 608, 32, "fp16")
x32 = synth.synthetic_input(32, 608, 608, 3, seed=5)
x64 = x32.astype(np.float64)
xp = torch.from_numpy(x32).pin_memory()
xd = torch.from_numpy(x32).cuda()
for name, x in (("device-resident float32 (bench.py's step)", xd), ("host float32, pinned", xp), ("host float32, pageable", x32), ("host float64 (the reference's dtype)", x64)):
    for _ in range(2):
        model.predict(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        boxes = model.predict(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-44s %7.2f ms/batch  %8.1f images/s   (%d boxes)" % (name, dt * 1e3, 32 / dt, sum(len(b) for b in boxes)))
