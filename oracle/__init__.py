"""CPU oracle for the YOLO v2/v3 TEST-mode hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy / torch-CPU) of the algorithm the
reference wns349/tensorflow-yolo runs on its inference path.  It exists so the
HIP path can be checked against it.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``tensorflow-yolo_amd/``) never does and fails loudly when its HIP extension
is missing.

Pinning status
--------------
* decode + NMS half (``decode_ref.py``) -- PINNED: checked against the
  reference's own NumPy functions (``net/v2.py:83-119``, ``net/v3.py:109-151``,
  ``net/base.py:171-209``) imported in the build container; the outputs of the
  reference itself are committed as ``tests/golden/decode_*.npz`` together with
  the generating script ``oracle/gen_golden.py``.
* conv-stack half (``forward_ref.py``) -- PARITY UNPINNED: the arithmetic lives
  in TensorFlow 1.x (``requirements.txt:5,7``: tensorflow>=1.10.1 /
  tensorflow_gpu>=1.9.0, not vendored, not installable here) and the reference
  holds no tests, golden logits or weights for it.  The restatement follows the
  call sites ``net/layers.py:9-134`` and is cross-checked two independent ways
  (torch conv2d in fp64 vs a naive NumPy direct convolution) in
  ``tests/test_oracle_forward.py``.
* image preprocessing (``preprocess_ref.py``) -- PARITY UNPINNED:
  the arithmetic lives in OpenCV (``requirements.txt``: opencv-python, absent);
  restates OpenCV's published 8-bit INTER_LINEAR fixed-point algorithm and is
  property-checked in ``tests/test_oracle_golden.py``.
"""
