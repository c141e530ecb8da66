"""Run by tests/test_sanitizer.py under the ASan + UBSan build of the library (no GPU): plans of the three network families, both dtypes, and
the whole host side of yolo_net_load_weights -- BN fold, repack into the kernels' [Cout_pad][K] chunk layout (plan.cpp: pack_weights) -- up to the
one call that needs a device (the copy: YOLO_ERR_HIP here); a stream one value short must be refused (YOLO_ERR_WEIGHTS, stricter than the
reference's net/base.py:44)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tensorflow_yolo_amd import _hip
from tensorflow_yolo_amd.net import engine, synth, v2, v3
names = ["c%d" % i for i in range(80)]
for kind, dtype in (("v3", "fp16"), ("v3", "fp32"), ("v2", "fp16"), ("tiny", "fp32")):
    if kind == "v3":
        net = v3.create_network(np.reshape([10,13,16,30,33,23,30,61,62,45,59,119,116,90,156,198,373,326], [-1, 2]), names, False, input_shape=(608, 608, 3))
    elif kind == "v2":
        net = v2.create_full_network(np.reshape([0.57273,0.677385,1.87446,2.06253,3.33843,5.47434,7.88282,3.52778,9.77052,9.16828], [-1, 2]), names, False, input_shape=(416, 416, 3))
    else:
        net = v2.create_tiny_network(np.reshape([1.08,1.19,3.42,4.41,6.63,11.38,9.42,5.11,16.62,10.52], [-1, 2]), names[:20], False, input_shape=(416, 416, 3))
    plan = engine.Plan(net, dtype=dtype, max_batch=8)
    w = synth.darknet_stream(net, seed=1, num_classes=80 if kind != "tiny" else 20)
    rc = plan.lib.yolo_net_load_weights(plan.handle, w.ctypes.data, w.size, C.c_void_p(0x10000000), plan.weights_bytes)
    print(kind, dtype, "load_weights rc", rc, plan.lib.yolo_last_error().decode()[:80])
    assert rc == 3, rc          # YOLO_ERR_HIP: everything up to the device copy ran
    rc = plan.lib.yolo_net_load_weights(plan.handle, w.ctypes.data, w.size - 1, C.c_void_p(0x10000000), plan.weights_bytes)
    print("  short stream rc", rc)
    assert rc == 4, rc          # YOLO_ERR_WEIGHTS
    plan.close()
print("pack probe OK")
