import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
from oracle import decode_ref
from tensorflow_yolo_amd.net import synth, engine as yengine
kind, size, batch, dtype = bench.WORKLOADS["v3-608-b32-fp32"]
model, w, anchors, ncls = bench.make_model(kind, size, batch, dtype, streams=1)
eng = model.net.engine
x = torch.from_numpy(synth.synthetic_input(batch, size, size, 3, seed=1000)).cuda()
logits = eng.forward(x).cpu().numpy()
recs, _ = yengine.records_to_host(*eng.detect(x, 0.5, 0.6))
sc = decode_ref.v3_scales(anchors, (size, size))
want = decode_ref.find_bounding_boxes_v3(logits, 0.5, 0.6, sc)
for i in range(batch):
    g = recs[i]; wv = [b.astuple() for b in want[i]]
    if len(g) != len(wv): print("image", i, "count", len(g), len(wv))
    for k, (a, b) in enumerate(zip(g, wv)):
        if int(a[4]) != int(b[4]) or abs(a[5]-b[5]) > 2e-6 or max(abs(a[j]-b[j]) / max(1.0, 10*abs(b[j])) for j in range(4)) > 2e-5:
            print("image", i, "box", k, "hip", a, "oracle", b)
            # find the row: match by prob among logits
            rows = logits[i]
            p = 1/(1+np.exp(-rows[:,4].astype(np.float32)))
            r = int(np.argmin(np.abs(p - b[5])))
            cl = rows[r,5:]
            top = np.argsort(-cl)[:3]
            s32 = (1/(1+np.exp(-cl.astype(np.float32)))).astype(np.float32)
            print("   row", r, "top class logits", [(int(t), float(cl[t]), float(s32[t])) for t in top])
