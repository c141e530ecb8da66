"""Data-parallel sharding of the hot path over the GPUs of one node (NEW functionality: the
reference is single-process).

Images are independent units (reference net/v2.py:87-89, net/v3.py:142-150 loop per image; NMS
never crosses images), so the batch dimension shards with no data-path collective at all.  The one
exchange is ONE all-gather per batch of the FINAL fixed-size record buffer: the engine
writes counts, status words and box records of its shard into one contiguous device buffer

    int32 words:  [ counts[B] | status[B] | boxes[B][K] x {f32 x, y, w, h, prob; i32 class} ]

(B = the engine's max_batch, K = max_boxes; ~200 KB per rank at B = 32, K = 256: latency bound on
xGMI) and that buffer goes through `all_gather_into_tensor` as it is -- no packing pass, one RCCL
call.  One process per GPU; `torch.distributed` backend "nccl" (= RCCL on ROCm) on the GPU box,
"gloo" in the CPU tests, which drive this same code on CPU tensors.
"""
import numpy as np


def shard_range(n_images, rank, world_size):
    """Contiguous shard [lo, hi) of a global batch: rank r gets images r*ceil(n/w) ...."""
    per = -(-n_images // world_size)
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def record_words(max_batch, max_boxes):
    """int32 words of one rank's record buffer."""
    return 2 * max_batch + 6 * max_boxes * max_batch


def split_records(flat, max_batch, max_boxes):
    """Views into record buffers: flat [..., record_words] int32 ->
    (boxes [..., B, K, 6] float32 (class index = int32 bits of field 5), counts [..., B], status [..., B])."""
    import torch
    B, K = max_batch, max_boxes
    lead = tuple(flat.shape[:-1])
    counts = flat[..., :B]
    status = flat[..., B:2 * B]
    boxes = flat[..., 2 * B:].view(torch.float32).reshape(lead + (B, K, 6))
    return boxes, counts, status


def world(group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def broadcast_rank0_int(value, group=None, device="cpu"):
    """Every rank passes its own value, every rank gets rank 0's (a collective; `device`: where the one-word tensor lives --
    the GPU under RCCL, the CPU under gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int32, device=device)
    dist.broadcast(t, src=0, group=group)
    return int(t.item())


def gather_records(flat, group=None, always=False):
    """The path's only collective: all-gather of the per-rank record buffer.
    flat: [record_words] int32 (device or CPU).  Returns [world, record_words]; rank order == image order.
    `always`: issue the collective even in a one-rank group (the GPU test uses it to put this exact call through RCCL)."""
    import torch
    import torch.distributed as dist
    _, w = world(group)
    if w == 1 and not (always and dist.is_available() and dist.is_initialized()):
        return flat.reshape(1, -1)
    out = torch.empty(w * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat.contiguous().reshape(-1), group=group)
    return out.view(w, flat.numel())


def detect_sharded(engine, x_local, threshold, iou_threshold, nms_mode=0, group=None):
    """One step of the sharded hot path on this rank: forward + decode + NMS of the local shard (one C call,
    `engine.detect`) and the all-gather of the record buffer.  x_local may be None / empty when the global batch
    leaves this rank without images (its counts are then zero).  `engine` needs `.records` (int32
    [record_words]), `.max_batch`, `.max_boxes` and `.detect()`; the gloo tests pass a CPU stand-in.
    Returns (boxes [W, B, K, 6], counts [W, B], status [W, B]) as views of the gathered buffer."""
    n_local = 0 if x_local is None else int(x_local.shape[0])
    B = engine.max_batch
    _, w = world(group)
    if w > 1 and not getattr(engine, "_streams_agreed", False):
        # once per engine, on every rank (a collective): all ranks run the plan rank 0 measured -- one pass or two half batches --
        # instead of each rank's own tuner answer (engine.agree_streams; stand-in engines of the CPU tests have none)
        agree = getattr(engine, "agree_streams", None)
        if agree is not None:
            agree(x_local if n_local else None, group)
        engine._streams_agreed = True
    if n_local:
        engine.detect(x_local, threshold, iou_threshold, nms_mode)
    if n_local < B:         # records of images this rank did not run: count 0, status 0
        engine.records[n_local:B].zero_()
        engine.records[B + n_local:2 * B].zero_()
    return split_records(gather_records(engine.records, group), B, engine.max_boxes)


def records_to_lists(boxes, counts, n_images=None):
    """[..., B, K, 6] / [..., B] tensors -> list of [(x, y, w, h, class_idx, prob)] per image, rank-major
    (= global image order when every rank but the last holds B images); `n_images` drops the padding."""
    b = boxes.cpu().numpy()
    c = counts.cpu().numpy().reshape(-1)
    b = b.reshape((-1,) + b.shape[-2:])
    cls = np.ascontiguousarray(b[..., 5]).view(np.int32)
    n = b.shape[0] if n_images is None else n_images
    return [[(float(b[i, k, 0]), float(b[i, k, 1]), float(b[i, k, 2]), float(b[i, k, 3]), int(cls[i, k]), float(b[i, k, 4]))
             for k in range(int(c[i]))] for i in range(n)]
