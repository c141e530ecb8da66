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
* network topology + Darknet weight order (``topology.py``, the layer lists ``forward_ref.py`` walks) -- PINNED: the
  reference's own graph builders (``net/v2.py:11-60``, ``net/v3.py:9-94`` over ``net/layers.py``) were executed in
  the build container under a RECORDING ``tensorflow`` module (``oracle/gen_topology.py``); what they built -- layer
  classes, TF op arguments, source layers by ``.out`` identity, shapes, ``variable_names`` -- is committed as
  ``tests/golden/topology_{v2_416,v3_416,v3_608}.json`` and ``tests/test_oracle_forward.py`` asserts
  product == oracle == fixture.
* conv-stack ARITHMETIC (``forward_ref.py``) -- PARITY UNPINNED: the arithmetic lives
  in TensorFlow 1.x (``requirements.txt:5,7``: tensorflow>=1.10.1 /
  tensorflow_gpu>=1.9.0, not vendored, not installable here) and the reference
  holds no tests, golden logits or weights for it.  The restatement follows the
  call sites ``net/layers.py:9-134`` and is cross-checked two independent ways
  (torch conv2d in fp64 vs a naive NumPy direct convolution) in
  ``tests/test_oracle_forward.py``.
* the parity gate (``parity.py``) -- checker built on the two halves above: max |logit error|, post-NMS box sets and
  the margin rule of SURVEY 7.3 #3; used by tests/, ``bench.py`` (outside the timed region) and nothing else.
* image preprocessing (``preprocess_ref.py``) -- PARITY UNPINNED:
  the arithmetic lives in OpenCV (``requirements.txt``: opencv-python, absent);
  restates OpenCV's published 8-bit INTER_LINEAR fixed-point algorithm and is
  property-checked in ``tests/test_oracle_golden.py``.
"""
