import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import to_oracle
from oracle import forward_ref
from tensorflow_yolo_amd import YoloV3
from tensorflow_yolo_amd.net import synth
import bench
names = ["c%d" % i for i in range(80)]
net = YoloV3.create_network(np.reshape(bench.COCO_V3, [-1, 2]), names, False, input_shape=(608, 608, 3))
w = synth.darknet_stream(net, seed=0, num_classes=80)
L = to_oracle(net); Wd = forward_ref.parse_darknet_weights(L, w)
print("cpu count", os.cpu_count(), "torch default threads", torch.get_num_threads())
for thr in (16, 32, 64, 128):
    for chunk in (2, 8):
        torch.set_num_threads(thr)
        x = synth.synthetic_input(chunk, 608, 608, 3, seed=1)
        forward_ref.forward(L, Wd, x)
        t0 = time.perf_counter(); forward_ref.forward(L, Wd, x); dt = time.perf_counter() - t0
        print("threads %3d chunk %d: %.2f img/s" % (thr, chunk, chunk / dt), flush=True)
