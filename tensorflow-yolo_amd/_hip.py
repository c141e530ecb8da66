"""ctypes binding of libyolo_hip.so (C ABI: include/yolo_hip.h).

The HIP library IS the product path: there is no CPU fallback.  Loading fails loudly
(ImportError with the build hint) when the shared object is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (YOLO_HIP_LIB: another build of the SAME library -- the sanitizer build of tests/test_sanitizer.py, A/B builds of tools/; no fallback of any kind)
LIB_PATH = os.environ.get("YOLO_HIP_LIB") or os.path.join(_HERE, "libyolo_hip.so")

ABI_VERSION = 5

# enum yolo_op
OP_INPUT, OP_CONV, OP_MAXPOOL, OP_ROUTE, OP_REORG, OP_SHORTCUT, OP_UPSAMPLE, OP_YOLO, OP_DETECTION = range(9)
DTYPE_F32, DTYPE_F16 = 0, 1
NMS_AGNOSTIC, NMS_PER_CLASS = 0, 1
MAX_SRC, MAX_ANCHORS, MAX_SCALES = 4, 8, 4
# Records per image every Python entry point asks for unless told otherwise.  The reference's lists are unbounded
# (net/base.py:195-209); 1024 covers every YOLOv2 head (845 rows) and exceeding it raises (engine.check_status).
DEFAULT_MAX_BOXES = 1024


class LayerDesc(C.Structure):
    _fields_ = [("op", C.c_int32), ("n_src", C.c_int32), ("src", C.c_int32 * MAX_SRC),
                ("filters", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32),
                ("batch_norm", C.c_int32), ("leaky", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
                ("n_anchors", C.c_int32), ("anchors", C.c_double * (2 * MAX_ANCHORS))]


class NetOptions(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("max_batch", C.c_int32), ("keep_all", C.c_int32),
                ("cand_capacity", C.c_int32), ("max_boxes", C.c_int32), ("streams", C.c_int32), ("force_tile", C.c_int32), ("guard_bytes", C.c_int32), ("f32_products", C.c_int32)]


class Box(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float),
                ("prob", C.c_float), ("class_idx", C.c_int32)]


class HeadDesc(C.Structure):
    _fields_ = [("version", C.c_int32), ("n_classes", C.c_int32), ("n_scales", C.c_int32),
                ("h", C.c_int32 * MAX_SCALES), ("w", C.c_int32 * MAX_SCALES), ("n_anchors", C.c_int32 * MAX_SCALES),
                ("anchors", (C.c_double * (2 * MAX_ANCHORS)) * MAX_SCALES)]


class WsRegion(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("offset", C.c_uint64), ("used_bytes", C.c_uint64), ("region_bytes", C.c_uint64)]


class KernelInfo(C.Structure):
    _fields_ = [("kind", C.c_int32), ("layer", C.c_int32), ("variant", C.c_int32), ("ksize", C.c_int32),
                ("stride", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("flops", C.c_double), ("bytes", C.c_double), ("weight_bytes", C.c_double), ("name", C.c_char * 64),
                ("symbol", C.c_char * 160)]


# name -> (restype, argtypes); every symbol include/yolo_hip.h declares
SIGNATURES = {
    "yolo_hip_abi_version": (C.c_int, []),
    "yolo_last_error": (C.c_char_p, []),
    "yolo_net_create": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.POINTER(NetOptions), C.POINTER(C.c_void_p)]),
    "yolo_net_destroy": (None, [C.c_void_p]),
    "yolo_net_weight_count": (C.c_size_t, [C.c_void_p]),
    "yolo_net_weights_bytes": (C.c_size_t, [C.c_void_p]),
    "yolo_net_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "yolo_net_output_count": (C.c_size_t, [C.c_void_p]),
    "yolo_net_flops_per_image": (C.c_double, [C.c_void_p]),
    "yolo_net_head_desc": (C.c_int, [C.c_void_p, C.POINTER(HeadDesc)]),
    "yolo_net_set_head": (C.c_int, [C.c_void_p, C.POINTER(HeadDesc)]),
    "yolo_net_num_kernels": (C.c_int, [C.c_void_p]),
    "yolo_net_num_streams": (C.c_int, [C.c_void_p]),
    "yolo_net_describe": (C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "yolo_net_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "yolo_net_bind_workspace": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "yolo_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "yolo_net_detect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "yolo_net_autotune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_net_tune_streams": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "yolo_net_set_streams": (C.c_int, [C.c_void_p, C.c_int]),
    "yolo_net_workspace_regions": (C.c_int, [C.c_void_p, C.POINTER(WsRegion), C.c_int]),
    "yolo_net_kernel_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(KernelInfo)]),
    "yolo_net_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "yolo_net_read_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "yolo_decode_scratch_bytes": (C.c_size_t, [C.POINTER(HeadDesc), C.c_int, C.c_int]),
    "yolo_decode_nms": (C.c_int, [C.POINTER(HeadDesc), C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "yolo_preprocess_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "yolo_nms_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int,
                                C.c_void_p, C.POINTER(C.c_int32)]),
}

_lib = None


class YoloHipError(RuntimeError):
    """A libyolo_hip call returned a non-zero status (message from yolo_last_error)."""


def lib():
    """Load (once) and return the shared library with typed entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libyolo_hip.so not found at %s: the HIP extension is the only compute path of this package "
            "(no CPU fallback). Build it with `python __graft_entry__.py` or `make -C tensorflow-yolo_amd/csrc`."
            % LIB_PATH)
    # torch first: libyolo_hip.so links the system libamdhip64 while torch-ROCm carries its own copy; whichever is loaded first
    # serves both, and a process that loaded the system copy before torch ends up with two HIP runtimes of which ours sees no
    # device ("no ROCm-capable device is detected" from hipMemcpy).  torch is this package's device-memory plumbing anyway.
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = handle.yolo_hip_abi_version()
    if v != ABI_VERSION:
        raise ImportError("libyolo_hip.so ABI version %d, binding expects %d: rebuild the library" % (v, ABI_VERSION))
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().yolo_last_error()
        raise YoloHipError("%s failed (status %d): %s" % (what or "libyolo_hip call", rc, (msg or b"").decode()))
