"""MI355X-native YOLO v2/v3 TEST-mode inference (drop-in for wns349/tensorflow-yolo's hot path).

Python host code calls hand-written gfx950 HIP kernels through a ctypes C ABI
(`include/yolo_hip.h`, `libyolo_hip.so`); torch-ROCm tensors only hold device memory.
Importable as `tensorflow_yolo_amd` (the directory is named `tensorflow-yolo_amd`; the
shim `tensorflow_yolo_amd.py` at the repository root maps one onto the other).
"""
from . import _hip  # noqa: F401
from .net.base import BoundingBox  # noqa: F401
from .net.yolo import Yolo, YoloV2, YoloV2Tiny, YoloV3  # noqa: F401

__all__ = ["Yolo", "YoloV2", "YoloV2Tiny", "YoloV3", "BoundingBox"]
