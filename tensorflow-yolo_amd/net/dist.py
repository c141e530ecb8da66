"""Data-parallel sharding of the hot path over the GPUs of one node (NEW functionality: the
reference is single-process).

Images are independent units (reference net/v2.py:87-89, net/v3.py:142-150 loop per image; NMS
never crosses images), so the batch dimension shards with no data-path collective at all.  The one
exchange is an all-gather of the FINAL fixed-size box records (a few hundred KB per rank: latency
bound on xGMI, one RCCL call per batch).  One process per GPU; `torch.distributed` backend "nccl"
(= RCCL on ROCm) on the GPU box, "gloo" in the CPU tests.
"""
import numpy as np


def shard_range(n_images, rank, world_size):
    """Contiguous shard [lo, hi) of a global batch: rank r gets images r*ceil(n/w) ...."""
    per = -(-n_images // world_size)
    lo = min(n_images, rank * per)
    return lo, min(n_images, lo + per)


def gather_records(boxes, counts, status=None, group=None):
    """All-gather the per-rank record buffers; result order = rank order = image order.

    boxes [B_local, K, 6] float32, counts [B_local] int32 (device or CPU tensors; every rank must
    pass the same B_local -- pad the last shard).  Returns (boxes [W*B_local, K, 6], counts, status)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return boxes, counts, status
    w = dist.get_world_size(group)
    boxes = boxes.contiguous()
    counts = counts.contiguous()
    gb = torch.empty((w * boxes.shape[0],) + tuple(boxes.shape[1:]), dtype=boxes.dtype, device=boxes.device)
    gc = torch.empty((w * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(gb, boxes, group=group)
    dist.all_gather_into_tensor(gc, counts, group=group)
    gs = None
    if status is not None:
        status = status.contiguous()
        gs = torch.empty((w * status.shape[0],), dtype=status.dtype, device=status.device)
        dist.all_gather_into_tensor(gs, status, group=group)
    return gb, gc, gs


def records_to_lists(boxes, counts):
    """[N,K,6] / [N] tensors -> list[N] of [(x, y, w, h, class_idx, prob)]."""
    b = boxes.cpu().numpy()
    c = counts.cpu().numpy()
    cls = np.ascontiguousarray(b[..., 5]).view(np.int32)
    return [[(float(b[i, k, 0]), float(b[i, k, 1]), float(b[i, k, 2]), float(b[i, k, 3]), int(cls[i, k]), float(b[i, k, 4]))
             for k in range(int(c[i]))] for i in range(b.shape[0])]
