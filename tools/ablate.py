"""experiment: per-kernel forward times of v3-608 b32 fp16 under YOLO_CONV_DBG ablations (results are wrong by design)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from collections import defaultdict
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
for _ in range(2): eng.forward_timed(x)
ms = np.zeros(eng.num_kernels)
for _ in range(5): ms += eng.forward_timed(x)
ms /= 5
g = defaultdict(lambda: [0, 0.0])
for k, ki in enumerate(eng.kernel_infos()):
    key = "%s k%d s%d %d->%d @%d" % (ki.name.decode(), ki.ksize, ki.stride, ki.cin, ki.cout, ki.out_h)
    g[key][0] += 1; g[key][1] += ms[k]
print("DBG=%s total %.3f ms" % (os.environ.get("YOLO_CONV_DBG", "0"), ms.sum()))
for key, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:8]:
    print("   %-70s n=%2d %.3f ms" % (key, v[0], v[1]))
