"""diagnostic: v3-608 detect at b=2 and b=32 (fp16): logits sanity + counts vs oracle decode"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import decode_ref
import bench
for batch in (2, 32):
    model, w, anchors, ncls = bench.make_model("v3", 608, batch, "fp16")
    eng = model.net.engine
    from tensorflow_yolo_amd.net import synth
    x = torch.from_numpy(synth.synthetic_input(batch, 608, 608, 3, seed=1000)).cuda()
    logits = eng.forward(x)
    l = logits.cpu().numpy()
    obj = l[..., 4]
    print("batch", batch, "logits nan", int(np.isnan(l).sum()), "inf", int(np.isinf(l).sum()), "max|l| %.2f" % np.nanmax(np.abs(l)),
          "obj mean %.3f std %.3f max %.3f" % (obj.mean(), obj.std(), obj.max()), "rows obj>0 per image", (obj > 0).sum(axis=1)[:4])
    boxes, counts, status = eng.detect(x, 0.5, 0.6)
    print("  detect counts", counts.cpu().numpy()[:8], "status", status.cpu().numpy()[:8])
    sc = [(y.h, y.w, y.anchors) for y in model.net[-1].yolos]
    want = decode_ref.find_bounding_boxes_v3(l[:1], 0.5, 0.6, sc)
    print("  oracle on GPU logits, image 0:", len(want[0]))
    # second call on the same input, and on the alternate input
    boxes, counts, status = eng.detect(x, 0.5, 0.6)
    print("  detect again counts", counts.cpu().numpy()[:8])
    del model, eng
