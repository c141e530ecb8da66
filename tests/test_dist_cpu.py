"""CPU: the N>1 path (shard images over ranks, all-gather the box records) with world_size 2 on gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT  # noqa: F401
from tensorflow_yolo_amd.net import dist as ydist


def test_shard_range_covers_batch_in_rank_order():
    for n, w in ((256, 8), (10, 4), (3, 8), (32, 1)):
        spans = [ydist.shard_range(n, r, w) for r in range(w)]
        flat = [i for lo, hi in spans for i in range(lo, hi)]
        assert flat == list(range(n))
    assert ydist.shard_range(256, 3, 8) == (96, 128)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, K = 3, 5
        rng = np.random.RandomState(100 + rank)
        boxes = torch.from_numpy(rng.rand(B, K, 6).astype(np.float32))
        boxes[..., 5] = torch.from_numpy(rng.randint(0, 80, (B, K)).astype(np.int32)).view(torch.float32)
        counts = torch.tensor([rank + 1, 0, K], dtype=torch.int32)
        status = torch.zeros(B, dtype=torch.int32)
        gb, gc, gs = ydist.gather_records(boxes, counts, status)
        q.put((rank, gb.numpy().copy(), gc.numpy().copy(), boxes.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_gather_records_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    local = [g[3] for g in got]
    for rank, gb, gc, _ in got:
        assert gb.shape == (6, 5, 6) and np.array_equal(gb[:3], local[0]) and np.array_equal(gb[3:], local[1])   # rank order == image order
        assert gc.tolist() == [1, 0, 5, 2, 0, 5]
    lists = ydist.records_to_lists(torch.from_numpy(got[0][1]), torch.from_numpy(got[0][2]))
    assert [len(l) for l in lists] == [1, 0, 5, 2, 0, 5]
    assert isinstance(lists[0][0][4], int) and 0 <= lists[0][0][4] < 80


def test_single_process_is_identity():
    b, c = torch.zeros(2, 4, 6), torch.zeros(2, dtype=torch.int32)
    gb, gc, gs = ydist.gather_records(b, c)
    assert gb is b and gc is c and gs is None
