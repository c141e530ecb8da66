"""CPU: the N>1 path with world_size 2 on gloo -- images shard over ranks, ONE all-gather of the fixed-size
record buffer.  The ranks run the very functions the GPU path runs (`net/dist.py: detect_sharded`, which is
`bench.py`'s step, and `Yolo.predict`'s sharded branch) on CPU tensors; only the engine is a stand-in that
fills the record buffer from the input instead of launching kernels (no oracle involved: this is plumbing)."""
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT  # noqa: F401
from tensorflow_yolo_amd import YoloV3
from tensorflow_yolo_amd.net import dist as ydist


def test_shard_range_covers_batch_in_rank_order():
    for n, w in ((256, 8), (10, 4), (3, 8), (32, 1)):
        spans = [ydist.shard_range(n, r, w) for r in range(w)]
        flat = [i for lo, hi in spans for i in range(lo, hi)]
        assert flat == list(range(n))
    assert ydist.shard_range(256, 3, 8) == (96, 128)


def test_record_buffer_layout_is_one_contiguous_block():
    B, K = 3, 5
    flat = torch.arange(ydist.record_words(B, K), dtype=torch.int32)
    boxes, counts, status = ydist.split_records(flat, B, K)
    assert counts.tolist() == [0, 1, 2] and status.tolist() == [3, 4, 5]
    assert boxes.shape == (B, K, 6) and boxes.dtype == torch.float32
    assert boxes.data_ptr() == flat.data_ptr() + 4 * 2 * B                 # views, no copies
    assert boxes.view(torch.int32).reshape(-1).tolist() == list(range(2 * B, ydist.record_words(B, K)))


class StandInEngine(object):
    """What detect_sharded needs from net/engine.py: HipNetwork, on CPU tensors: `.records`, `.max_batch`, `.max_boxes`,
    `.detect()`.  Image i with mean value m yields round(m * 8) boxes whose fields encode (image tag, k)."""
    weights_loaded = True

    def __init__(self, max_batch, max_boxes):
        self.max_batch, self.max_boxes = max_batch, max_boxes
        self.records = torch.full((ydist.record_words(max_batch, max_boxes),), 12345, dtype=torch.int32)   # stale garbage
        self._boxes, self._counts, self._status = ydist.split_records(self.records, max_batch, max_boxes)
        self.calls = 0
        self.agreed = []

    def agree_streams(self, x, group=None):
        """HipNetwork.agree_streams with the measurement replaced by a rank-dependent answer: rank 0 'measures' one pass, the
        others two halves; everybody must end up with rank 0's."""
        mine = 1 if dist.get_rank() == 0 else 2
        self.agreed.append(ydist.broadcast_rank0_int(mine, group))

    def detect(self, x, threshold, iou_threshold, nms_mode=0):
        self.calls += 1
        b = x.shape[0]
        for i in range(b):
            tag = float(x[i].mean())
            n = int(round(tag * 8))
            self._counts[i] = n
            self._status[i] = 0
            for k in range(n):
                self._boxes[i, k, :5] = torch.tensor([tag, k, 0.5, 0.25, 1.0 - 0.01 * k])
                self._boxes[i, k, 5:] = torch.tensor([k + 7], dtype=torch.int32).view(torch.float32)
        return self._boxes[:b], self._counts[:b], self._status[:b]


def _expected(tags):
    return [[(np.float32(t), float(k), 0.5, 0.25, k + 7, float(np.float32(1.0 - 0.01 * k))) for k in range(int(round(t * 8)))] for t in tags]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


TAGS = [0.125, 0.375, 0.0, 0.625, 0.25]       # global batch of 5 images over 2 ranks: shards of 3 and 2 (one padded slot)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.stack([torch.full((4, 4, 3), t) for t in TAGS])
        eng = StandInEngine(max_batch=3, max_boxes=6)
        # (1) the bench step: detect on the local shard + the one all-gather
        lo, hi = ydist.shard_range(len(TAGS), rank, world)
        boxes, counts, status = ydist.detect_sharded(eng, x[lo:hi], 0.5, 0.6)
        lists = ydist.records_to_lists(boxes, counts)
        # (2) the product API: every rank passes the global batch, gets the global result
        model = YoloV3()
        model.net = type("Net", (list,), {})()
        model.net.engine = eng
        got = model.predict(x, 0.5, 0.6)
        # (3) a global batch smaller than the world: rank 1 has no image and still takes part in the exchange
        got1 = model.predict(x[:1], 0.5, 0.6)
        q.put((rank, tuple(boxes.shape), counts.tolist(), status.tolist(), lists,
               [[(b.x, b.y, b.w, b.h, b.class_idx, b.prob) for b in img] for img in got],
               [[(b.x, b.y, b.w, b.h, b.class_idx, b.prob) for b in img] for img in got1], eng.calls, eng.agreed))
    finally:
        dist.destroy_process_group()


def test_sharded_step_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=60) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = _expected(TAGS)
    for rank, shape, counts, status, lists, pred, pred1, calls, agreed in got:
        assert agreed == [1]        # one agreement per engine (three sharded calls), rank 0's answer on every rank
        assert shape == (2, 3, 6, 6)                                    # [world, B, K, 6]: rank order == image order
        assert counts == [[1, 3, 0], [5, 2, 0]] and status == [[0, 0, 0], [0, 0, 0]]     # the padded slot of rank 1 reads count 0
        flat = [lists[0], lists[1], lists[2], lists[3], lists[4]]       # slots rank-major; slot 5 is the pad
        for img, w in zip(flat, want):
            assert [tuple(np.float32(v) for v in b) for b in img] == [tuple(np.float32(v) for v in b) for b in w]
        assert lists[5] == []
        assert len(pred) == 5
        for img, w in zip(pred, want):                                  # Yolo.predict: same global list on every rank
            assert [(np.float32(b[0]), b[1], b[4]) for b in img] == [(np.float32(v[0]), v[1], v[4]) for v in w]
        assert len(pred1) == 1 and len(pred1[0]) == 1
        assert calls == (3 if rank == 0 else 2)                         # rank 1 ran no kernels for the 1-image batch


def _bench_worker(rank, world, port, q):
    """bench.py's own timed region and contract line under two gloo ranks (the stand-in engine in place of the HIP one)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, ROOT)
        import bench
        eng = StandInEngine(max_batch=3, max_boxes=6)
        xs = [torch.stack([torch.full((4, 4, 3), t) for t in TAGS[:3]]), torch.stack([torch.full((4, 4, 3), t) for t in TAGS[2:5]])]
        seen = []

        def step(i):
            seen.append(i)
            return ydist.detect_sharded(eng, xs[i & 1], 0.5, 0.6)

        elapsed, (boxes, counts, status) = bench.timed_region(step, 4, 2, world, lambda: None, torch.device("cpu"))
        line = bench.contract_fields(rank, world, 4, 2, 3, elapsed, "fp16")
        q.put((rank, elapsed, seen, tuple(boxes.shape), line))
    finally:
        dist.destroy_process_group()


def test_bench_timed_region_and_contract_line_world2_gloo():
    """The N > 1 half of the bench contract on CPU: W untimed + exactly K timed steps per rank, ONE time for the job (the MAX over
    ranks, equal on both), rank 0 alone gets a line -- with n_gpus = 2 and the whole-job images/sec -- and no other rank does
    (so `parity` / `cpu_baseline`, which bench.py hangs on that line, exist on rank 0 only)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (r0, e0, seen0, shape0, line0), (r1, e1, seen1, shape1, line1) = got
    assert e0 == e1 > 0                                  # max over ranks: one number for the job
    assert seen0 == seen1 == [0, 1, 0, 1, 2, 3]          # 2 warm-up + exactly 4 timed steps on every rank
    assert shape0 == shape1 == (2, 3, 6, 6)              # every step ended in the all-gather of both shards
    assert line1 is None                                 # rank 1 prints nothing: no line, hence no parity / cpu_baseline of its own
    assert line0["n_gpus"] == 2 and line0["steps"] == 4 and line0["warmup"] == 2 and line0["scaling"] == "weak"
    assert line0["value"] == round(4 * 3 * 2 / e0, 2) and line0["unit"] == "images/sec" and line0["vs_baseline"] is None
    assert "parity" not in line0 and "cpu_baseline" not in line0      # (added by main() on rank 0 only, after the timed region)


def test_single_process_is_identity():
    eng = StandInEngine(max_batch=2, max_boxes=4)
    x = torch.stack([torch.full((2, 2, 3), 0.25), torch.full((2, 2, 3), 0.5)])
    boxes, counts, status = ydist.detect_sharded(eng, x, 0.5, 0.6)
    assert boxes.shape == (1, 2, 4, 6) and counts.tolist() == [[2, 4]]
    assert boxes.data_ptr() == eng._boxes.data_ptr()                    # no gather, no copy


def test_bench_self_launches_ranks_without_a_gpu():
    """`python bench.py --gpus 2` as the driver calls it (no WORLD_SIZE): it must start two ranks itself, and in this
    GPU-less container both must end with the 'no GPU' error -- not with a usage error before any rank exists."""
    import pytest
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU (on the GPU box this would launch a real 2-GPU benchmark)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode not in (0, 2), r.stderr[-2000:]
    # (the launcher tears the other rank down as soon as one has failed, so only one of the two messages is guaranteed)
    assert "rank 0 needs GPU 0" in r.stderr or "rank 1 needs GPU 1" in r.stderr, r.stderr[-2000:]


# ---- the shipped launcher under a multi-process launcher (`torchrun launcher.py ...` must shard by itself) -------------
def _launcher_worker(rank, world, port, ini, q):
    """What `torchrun --nproc-per-node 2 launcher.py --config ... --mode test` gives each process: the rendezvous variables in
    the environment and nothing else.  launcher.main must join the group itself (gloo here: no GPU), Yolo.test must preprocess
    and run only the rank's shard, rank 0 alone writes the files."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    import io
    from contextlib import redirect_stdout
    from tensorflow_yolo_amd import launcher
    from tensorflow_yolo_amd.net import base, yolo
    seen = {"engine": None, "preprocessed": []}

    def build(self, anchors, class_names, input_shape=(416, 416, 3), dtype="fp32", max_batch=1, **kw):
        self.net = type("Net", (list,), {})()
        self.net.engine = seen["engine"] = StandInEngine(max_batch=max_batch, max_boxes=kw.get("max_boxes", 8))
        seen["kw"] = dict(kw, max_batch=max_batch)
        return self.net

    real_pre = base.preprocess_image

    def counting_pre(path, shape):
        seen["preprocessed"].append(os.path.basename(path))
        return real_pre(path, shape)

    yolo.Yolo.build = build
    yolo.YoloV2.load_weights = staticmethod(lambda net, path: None)
    base.preprocess_image = counting_pre
    out = io.StringIO()
    with redirect_stdout(out):
        launcher.main(["--config", ini, "--mode", "test"])
    q.put((rank, out.getvalue(), seen["preprocessed"], seen["engine"].calls, seen["kw"], dist.is_initialized()))


def test_launcher_main_shards_under_torchrun_environment(tmp_path):
    from PIL import Image
    img_dir, out_dir = tmp_path / "img", tmp_path / "out"
    img_dir.mkdir(); out_dir.mkdir()
    tags = {"a": 0.125, "b": 0.375, "c": 0.0, "d": 0.625, "e": 0.25}          # mean value -> number of stand-in boxes (x 8)
    for name, t in tags.items():
        Image.new("RGB", (40, 30), (int(round(t * 255)),) * 3).save(str(img_dir / (name + ".png")))
    names = ["c%d" % i for i in range(16)]
    ini = tmp_path / "y.ini"
    ini.write_text("[COMMON]\nversion = v2\ninput_h = 32\ninput_w = 32\ninput_c = 3\n[TEST]\nimage_dir = %s\nout_dir = %s\nbatch_size = 4\n"
                   "threshold = 0.5\niou_threshold = 0.6\nanchors = [1, 1, 2, 2]\nclass_names = %r\ncheckpoint_path =\n"
                   "pretrained_weights_path = none.weights\npreprocess = pillow\nmax_boxes = 8\ncand_capacity = 8192\n" % (img_dir, out_dir, names))
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_launcher_worker, args=(r, world, port, str(ini), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    order = sorted(os.listdir(str(img_dir)))            # (load_image_paths order is the directory order; compare as sets per batch)
    r0, r1 = got
    assert not r0[5] and not r1[5]                                          # launcher.main destroyed the group it created
    assert r0[4] == {"max_boxes": 8, "cand_capacity": 8192, "max_batch": 2}      # optional cap keys reach build(); 4 images / 2 ranks
    # every image is preprocessed by exactly one rank (batches of 4 + 1: shards 2 + 2, then 1 + 0)
    assert sorted(r0[2] + r1[2]) == order and len(r0[2]) == 3 and len(r1[2]) == 2
    assert r0[3] == 2 and r1[3] == 1                                        # kernels ran for the local shard only
    lines = [l for l in r0[1].splitlines() if "Found" in l]
    assert len(lines) == 5 and r0[1].strip().endswith("Done")
    for name, t in tags.items():
        assert any(l.startswith("%s: Found %d objects." % (name, int(round(int(round(t * 255)) / 255.0 * 8)))) for l in lines), (name, lines)
        assert os.path.exists(str(out_dir / (name + "_out.png")))
    assert "Found" not in r1[1] and "Done" not in r1[1]                     # rank 1 draws / writes nothing
