"""Runtime object behind a layer list: lowers it through the C ABI to a fused kernel plan and
runs it.  Stands where the reference has `tf.Session` + `sess.run(net[-1].out, {net[0].out: x})`
(reference net/yolo.py:67-83).

torch is used for exactly three things: device memory (uint8/float32 tensors whose
`.data_ptr()` goes to the C ABI), the current HIP stream, and host<->device copies.
"""
import ctypes as C

import numpy as np

from .. import _hip
from . import dist as ydist
from . import layers as L


def number_layers(net):
    for i, layer in enumerate(net):
        layer.index = i
    return net


def to_descs(net):
    """Layer objects -> the C ABI's yolo_layer_desc array (one per list element)."""
    number_layers(net)
    descs = (_hip.LayerDesc * len(net))()
    for i, layer in enumerate(net):
        d = descs[i]
        d.op = layer.op
        d.n_src = len(layer.inputs)
        if d.n_src > _hip.MAX_SRC:
            raise ValueError("layer %d has more than %d inputs" % (i, _hip.MAX_SRC))
        for k, src in enumerate(layer.inputs):
            if src.index is None or src.index >= i or net[src.index] is not src:
                raise ValueError("layer %d consumes a layer that is not earlier in the same list" % i)
            d.src[k] = src.index
        layer.fill_desc(d)
    return descs


def head_desc_v2(h, w, anchors, num_classes):
    """YOLOv2 keeps its anchors outside the graph (reference net/v2.py:83-85): one scale."""
    anchors = np.reshape(np.asarray(anchors, dtype=np.float64), [-1, 2])
    hd = _hip.HeadDesc()
    hd.version, hd.n_classes, hd.n_scales = 2, int(num_classes), 1
    hd.h[0], hd.w[0], hd.n_anchors[0] = int(h), int(w), len(anchors)
    for i, (aw, ah) in enumerate(anchors):
        hd.anchors[0][2 * i], hd.anchors[0][2 * i + 1] = float(aw), float(ah)
    return hd


def head_desc_v3(yolos):
    """From the yolo layers of a detection layer (reference net/v3.py:145-149)."""
    hd = _hip.HeadDesc()
    hd.version, hd.n_classes, hd.n_scales = 3, int(yolos[0].no_c), len(yolos)
    for s, y in enumerate(yolos):
        hd.h[s], hd.w[s], hd.n_anchors[s] = y.h, y.w, y.b
        for i, (aw, ah) in enumerate(y.anchors):
            hd.anchors[s][2 * i], hd.anchors[s][2 * i + 1] = float(aw), float(ah)
    return hd


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP backend needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback in this package")
    return torch


class Plan(object):
    """Device-free part: create the native net object, query sizes, print the plan.
    Works without a GPU (used by the CPU test-suite to check planning)."""

    def __init__(self, net, dtype="fp16", max_batch=1, keep_all=False, cand_capacity=4096, max_boxes=_hip.DEFAULT_MAX_BOXES, streams=0,
                 force_tile=None, guard_bytes=0, f32_products=0):
        self.lib = _hip.lib()
        self.layers = list(net)
        known = {"fp16": _hip.DTYPE_F16, "f16": _hip.DTYPE_F16, "half": _hip.DTYPE_F16,
                 "fp32": _hip.DTYPE_F32, "f32": _hip.DTYPE_F32, "float": _hip.DTYPE_F32}
        if str(dtype).lower() not in known:
            raise ValueError("dtype must be fp16 or fp32, got %r" % (dtype,))
        self.dtype = known[str(dtype).lower()]
        self.max_batch = int(max_batch)
        self.max_boxes = int(max_boxes)
        self.cand_capacity = int(cand_capacity)
        opt = _hip.NetOptions(dtype=self.dtype, max_batch=self.max_batch, keep_all=int(bool(keep_all)),
                              cand_capacity=self.cand_capacity, max_boxes=self.max_boxes, streams=int(streams),
                              force_tile=0 if force_tile is None else int(force_tile) + 1, guard_bytes=int(guard_bytes),
                              f32_products=int(f32_products))
        descs = to_descs(self.layers)
        handle = C.c_void_p()
        _hip.check(self.lib.yolo_net_create(descs, len(self.layers), C.byref(opt), C.byref(handle)), "yolo_net_create")
        self.handle = handle
        self.weight_count = self.lib.yolo_net_weight_count(handle)
        self.weights_bytes = self.lib.yolo_net_weights_bytes(handle)
        self.workspace_bytes = self.lib.yolo_net_workspace_bytes(handle)
        self.output_count = self.lib.yolo_net_output_count(handle)
        self.flops_per_image = self.lib.yolo_net_flops_per_image(handle)
        self.num_kernels = self.lib.yolo_net_num_kernels(handle)
        self._auto_streams = int(streams) <= 0                          # streams = 0: the library's rule, re-measured on the device
        self._streams_tuned = False                                     # (HipNetwork: at the first full batch)
        self.input_hwc = self.layers[0].out.hwc
        last = self.layers[-1]
        if isinstance(last, L.detection_layer):
            self.output_shape = (last.out.hwc[0], last.out.hwc[2])              # [rows, 5+C]
            self.set_head(head_desc_v3(last.yolos))
        else:
            self.output_shape = last.out.hwc                                    # [h, w, c]

    @property
    def num_streams(self):
        """parts / HIP streams a full batch currently runs as (yolo_net_options.streams; 0 = the library's rule, which a
        HipNetwork re-measures on the device at its first full batch: yolo_net_tune_streams)"""
        return self.lib.yolo_net_num_streams(self.handle)

    def workspace_regions(self):
        """[(name, offset, used_bytes, region_bytes)] of the planned workspace (yolo_net_workspace_regions: diagnostic / canary tests)."""
        n = self.lib.yolo_net_workspace_regions(self.handle, None, 0)
        arr = (_hip.WsRegion * n)()
        self.lib.yolo_net_workspace_regions(self.handle, arr, n)
        return [(r.name.decode(), int(r.offset), int(r.used_bytes), int(r.region_bytes)) for r in arr]

    def set_head(self, hd):
        _hip.check(self.lib.yolo_net_set_head(self.handle, C.byref(hd)), "yolo_net_set_head")
        self.head = hd

    def describe(self):
        n = self.lib.yolo_net_describe(self.handle, None, 0)
        buf = C.create_string_buffer(n)
        self.lib.yolo_net_describe(self.handle, buf, n)
        return buf.value.decode()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.yolo_net_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipNetwork(Plan):
    """Plan + device memory: forward() / detect() on the current torch HIP stream."""

    def __init__(self, net, dtype="fp16", max_batch=1, device=None, **kw):
        super(HipNetwork, self).__init__(net, dtype=dtype, max_batch=max_batch, **kw)
        torch = _torch()
        self.torch = torch
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        with torch.cuda.device(self.device):
            self._weights = torch.empty(max(self.weights_bytes, 256), dtype=torch.uint8, device=self.device)
            self._workspace = torch.zeros(max(self.workspace_bytes, 256), dtype=torch.uint8, device=self.device)
            # ONE record buffer [counts | status | boxes] (net/dist.py): what a multi-GPU run all-gathers as it is
            self.records = torch.zeros(ydist.record_words(self.max_batch, self.max_boxes), dtype=torch.int32, device=self.device)
            self._boxes, self._counts, self._status = ydist.split_records(self.records, self.max_batch, self.max_boxes)
            _hip.check(self.lib.yolo_net_bind_workspace(self.handle, self._workspace.data_ptr(), self._workspace.numel()),
                       "yolo_net_bind_workspace")
        self.weights_loaded = False

    # -- weights ------------------------------------------------------------------------------
    def load_weights(self, flat):
        """flat: float32 body of a Darknet .weights file, layer-list order (reference net/base.py:26-46)."""
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        with self.torch.cuda.device(self.device):
            _hip.check(self.lib.yolo_net_load_weights(self.handle, flat.ctypes.data, flat.size,
                                                      self._weights.data_ptr(), self._weights.numel()),
                       "yolo_net_load_weights")
        self.weights_loaded = True

    # -- data ---------------------------------------------------------------------------------
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def to_device(self, x_batch):
        """Host NumPy [B,H,W,C] (float64 in the reference, net/base.py:153) or a torch tensor ->
        float32 device tensor, the dtype the reference's placeholder casts to (net/layers.py:108)."""
        torch = self.torch
        if isinstance(x_batch, torch.Tensor):
            x = x_batch.to(device=self.device, dtype=torch.float32)
        else:
            x = torch.from_numpy(np.ascontiguousarray(x_batch, dtype=np.float32)).to(self.device)
        x = x.contiguous()
        h, w, c = self.input_hwc
        if x.dim() != 4 or tuple(x.shape[1:]) != (h, w, c):
            raise ValueError("expected input [B,%d,%d,%d], got %s" % (h, w, c, tuple(x.shape)))
        if not 1 <= x.shape[0] <= self.max_batch:
            raise ValueError("batch %d outside 1..%d (max_batch)" % (x.shape[0], self.max_batch))
        return x

    def _tune_streams(self, x):
        """streams = 0 and the rule said two halves: time one pass against two halves on THIS device, once, with the first batch
        that is large enough, and keep the faster (yolo_net_tune_streams: the gain is board-dependent, +4 % to -1.5 %)."""
        if self._streams_tuned or not self._auto_streams or not self.weights_loaded:
            return
        if self.lib.yolo_net_num_streams(self.handle) < 2 or 2 * x.shape[0] <= self.max_batch:
            self._streams_tuned = self.lib.yolo_net_num_streams(self.handle) < 2
            return
        with self.torch.cuda.device(self.device):
            _hip.check(self.lib.yolo_net_tune_streams(self.handle, x.data_ptr(), x.shape[0], self._stream()), "yolo_net_tune_streams")
        self._streams_tuned = True

    def agree_streams(self, x, group=None):
        """Ranks of a sharded batch must run the SAME plan: each rank measures for itself where it can, then rank
        0's answer is broadcast and taken by everyone -- one pass and two halves differ in fp16 summation order, and the job's step
        time is the slowest rank's.  A collective: every rank calls it once (net/dist.py does, at its first sharded detect)."""
        torch = self.torch
        if x is not None and x.shape[0] > 0:
            self._tune_streams(self.to_device(x))
        import torch.distributed as dist
        from . import dist as ydist
        want = ydist.broadcast_rank0_int(self.num_streams, group, self.device if dist.get_backend(group) == "nccl" else "cpu")
        if want != self.num_streams:
            _hip.check(self.lib.yolo_net_set_streams(self.handle, want), "yolo_net_set_streams")
        self._streams_tuned = True
        return want

    def forward(self, x, out=None):
        """Head logits (float32, device) in the reference's layout: v2 [B,h,w,A*(5+C)], v3 [B,rows,5+C]."""
        torch = self.torch
        x = self.to_device(x)
        self._tune_streams(x)
        b = x.shape[0]
        if out is None:
            out = torch.empty((b,) + tuple(self.output_shape), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.yolo_net_forward(self.handle, x.data_ptr(), b, out.data_ptr(), self._stream()),
                       "yolo_net_forward")
        return out

    def detect(self, x, threshold, iou_threshold, nms_mode=_hip.NMS_AGNOSTIC):
        """forward + decode + NMS, one enqueue.  Returns device tensors (boxes [B,K,6] as
        (x,y,w,h,prob,class-as-int32-bits), counts [B], status [B]); no host sync."""
        torch = self.torch
        x = self.to_device(x)
        self._tune_streams(x)
        b = x.shape[0]
        with torch.cuda.device(self.device):
            _hip.check(self.lib.yolo_net_detect(self.handle, x.data_ptr(), b, float(threshold), float(iou_threshold),
                                                int(nms_mode), self._boxes.data_ptr(), self._counts.data_ptr(),
                                                self._status.data_ptr(), self._stream()), "yolo_net_detect")
        return self._boxes[:b], self._counts[:b], self._status[:b]

    def autotune(self, x):
        """Pick the fastest conv tile per layer by timing them on the device (optional, synchronous)."""
        x = self.to_device(x)
        with self.torch.cuda.device(self.device):
            _hip.check(self.lib.yolo_net_autotune(self.handle, x.data_ptr(), x.shape[0], self._stream()), "yolo_net_autotune")

    def kernel_infos(self):
        out = []
        for k in range(self.num_kernels):
            ki = _hip.KernelInfo()
            _hip.check(self.lib.yolo_net_kernel_info(self.handle, k, C.byref(ki)), "yolo_net_kernel_info")
            out.append(ki)
        return out

    def forward_timed(self, x, out=None):
        """Instrumented forward: device milliseconds of every kernel (hipEvents on the launch stream)."""
        torch = self.torch
        x = self.to_device(x)
        b = x.shape[0]
        if out is None:
            out = torch.empty((b,) + tuple(self.output_shape), dtype=torch.float32, device=self.device)
        ms = np.zeros(self.num_kernels, dtype=np.float32)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.yolo_net_forward_timed(self.handle, x.data_ptr(), b, out.data_ptr(), self._stream(),
                                                       ms.ctypes.data), "yolo_net_forward_timed")
        return ms

    def read_layer(self, index, batch):
        """Dense float32 NHWC copy of one layer's output (needs keep_all=True)."""
        h, w, c = self.layers[index].out.hwc
        host = np.empty((batch, h, w, c), dtype=np.float32)
        _hip.check(self.lib.yolo_net_read_layer(self.handle, index, batch, host.ctypes.data, host.size), "yolo_net_read_layer")
        return host


def check_status(status, allow_truncation=False):
    """The reference's lists are unbounded (net/base.py:195-209); the record buffers are not.  Both overflow bits mean the
    result would differ from the reference's, so both raise: bit 0 = more candidates than cand_capacity passed the
    threshold, bit 1 = more NMS survivors than max_boxes."""
    status = np.asarray(status).reshape(-1)
    if (status & 1).any():
        raise _hip.YoloHipError("candidate capacity exceeded for image(s) %s: raise cand_capacity or the threshold"
                                % np.nonzero(status & 1)[0].tolist())
    if (status & 2).any() and not allow_truncation:
        raise _hip.YoloHipError("more than max_boxes boxes survive NMS for image(s) %s: the list would be truncated "
                                "(the reference has no cap): raise max_boxes" % np.nonzero(status & 2)[0].tolist())


def records_to_host(boxes, counts, status, allow_truncation=False):
    """Device records -> per-image list of (x, y, w, h, class_idx, prob) tuples (synchronises).
    Raises if a capacity was exceeded (the result would not be the reference's): check_status."""
    boxes = boxes.cpu().numpy()
    counts = counts.cpu().numpy()
    status = status.cpu().numpy()
    check_status(status, allow_truncation)
    out = []
    cls = boxes[..., 5].view(np.int32)
    for i in range(boxes.shape[0]):
        n = int(counts[i])
        b = boxes[i, :n].astype(np.float64).tolist()        # float32 -> the same Python floats as float(np.float32)
        c = cls[i, :n].tolist()
        out.append([(r[0], r[1], r[2], r[3], c[k], r[4]) for k, r in enumerate(b)])
    return out, status


def decode_nms(head, logits, threshold, iou_threshold, nms_mode=_hip.NMS_AGNOSTIC, cand_capacity=4096,
               max_boxes=_hip.DEFAULT_MAX_BOXES, allow_truncation=False):
    """Standalone decode + NMS of a head tensor already on (or copied to) the device:
    the drop-in for find_bounding_boxes (reference net/v2.py:83-90, net/v3.py:140-151)."""
    torch = _torch()
    lib = _hip.lib()
    if not isinstance(logits, torch.Tensor):
        logits = torch.from_numpy(np.ascontiguousarray(logits, dtype=np.float32)).cuda()
    logits = logits.contiguous().float()
    b = logits.shape[0]
    dev = logits.device
    with torch.cuda.device(dev):
        nbytes = lib.yolo_decode_scratch_bytes(C.byref(head), b, cand_capacity)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        boxes = torch.empty((b, max_boxes, 6), dtype=torch.float32, device=dev)
        counts = torch.zeros((2, b), dtype=torch.int32, device=dev)
        _hip.check(lib.yolo_decode_nms(C.byref(head), logits.data_ptr(), b, float(threshold), float(iou_threshold),
                                       int(nms_mode), cand_capacity, max_boxes, scratch.data_ptr(), nbytes,
                                       boxes.data_ptr(), counts[0].data_ptr(), counts[1].data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream), "yolo_decode_nms")
        return records_to_host(boxes, counts[0], counts[1], allow_truncation)
