#!/usr/bin/env python3
"""Headline benchmark: images/sec of the YOLO TEST-mode hot path (conv forward + head decode + NMS)
on N MI355X GPUs of one node, one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload v3-608-b32-fp16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one synthetic batch resident in HBM: float32 NHWC input ->
libyolo_hip detect (75 fused convs + decode + NMS) [-> RCCL all-gather of the fixed-size box
records when N > 1].  Images shard over ranks (weak scaling: every rank runs the full per-GPU batch).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run ... bench.py` started before this process touches the GPU; rank 0's JSON line is the
child's stdout, the exit code is the child's).

Rank 0 prints ONE JSON line with the contract fields plus
  "parity":       the metric's second half, "post-NMS box-set match vs CPU ref": EVERY image of the first timed batch through the
                  TIMED engine (same plan, same batch size) and through the CPU fp32 oracle pipeline OUTSIDE the timed region --
                  max |logit - oracle logit| against a bound the HIP path cannot influence and the post-NMS box sets under the
                  gate of oracle/parity.py
  "roofline":     dominant kernel family (the implicit-GEMM conv tile with most device time): algorithmic FLOPs of its launches
                  / their device time measured with hipEvents on the launch stream (instrumented steps
                  run right after the timed region; the events add bubbles so they never time `value`)
  "cpu_baseline": the CPU oracle (torch-CPU restatement of the reference's TF path + NumPy decode/NMS,
                  kind "port") timed on the host cores on the same batch: forward and decode + NMS apart.
  "one_stream" / "two_streams": (N = 1) the batch as one pass on one stream and as two half batches on two streams, ALWAYS both: the mode
                  `value` was timed in (`config.streams`; the library's rule + its on-device tuner decide) carries the timed region itself,
                  the other mode the same K steps right before and right after it (A / timed / A) -- so that lines from different boxes,
                  on which the tuner may decide differently, stay comparable.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (model, input, per-GPU batch, dtype)            BASELINE.json configs[2] / [1] / [4] / [0]
    "v3-608-b32-fp16": ("v3", 608, 32, "fp16"),
    "v2-416-b16-fp16": ("v2", 416, 16, "fp16"),
    "tiny-v2-voc-416-b64-fp32": ("v2-tiny", 416, 64, "fp32"),
    "v2-416-b1-fp32": ("v2", 416, 1, "fp32"),
    "v3-416-b32-fp16": ("v3", 416, 32, "fp16"),
    "v3-608-b8-fp16": ("v3", 608, 8, "fp16"),       # smaller batches of the headline net (tile-choice sanity, latency)
    "v3-608-b1-fp16": ("v3", 608, 1, "fp16"),
    "v3-608-b16-fp16": ("v3", 608, 16, "fp16"),     # points for the one-stream / two-stream rule (DESIGN.md "Kernel boundaries")
    "v3-416-b16-fp16": ("v3", 416, 16, "fp16"),
    "v2-416-b32-fp16": ("v2", 416, 32, "fp16"),
    "v2-416-b64-fp16": ("v2", 416, 64, "fp16"),
    # the float32 plan -- the carrier of north_star's "1e-4 on logits + identical boxes" -- at the headline batches
    "v3-608-b32-fp32": ("v3", 608, 32, "fp32"),
    "v2-416-b16-fp32": ("v2", 416, 16, "fp32"),
}
PEAK = {"fp16": 2500.0, "fp32": 157.3}     # dense MFMA TFLOP/s, MI355X_MICROARCH.md "Chip-level parameters"
COCO_V2 = [0.57273, 0.677385, 1.87446, 2.06253, 3.33843, 5.47434, 7.88282, 3.52778, 9.77052, 9.16828]
VOC_TINY = [1.08, 1.19, 3.42, 4.41, 6.63, 11.38, 9.42, 5.11, 16.62, 10.52]
COCO_V3 = [10, 13, 16, 30, 33, 23, 30, 61, 62, 45, 59, 119, 116, 90, 156, 198, 373, 326]


def make_model(kind, size, batch, dtype, seed=0, streams=0, max_boxes=256, **engine_kw):
    from tensorflow_yolo_amd import YoloV2, YoloV2Tiny, YoloV3
    from tensorflow_yolo_amd.net import synth
    cls, anchors, ncls = {"v3": (YoloV3, COCO_V3, 80), "v2": (YoloV2, COCO_V2, 80), "v2-tiny": (YoloV2Tiny, VOC_TINY, 20)}[kind]
    names = ["c%d" % i for i in range(ncls)]
    model = cls()
    net = cls.create_network(np.reshape(anchors, [-1, 2]), names, False, input_shape=(size, size, 3))
    hg, frac = synth.HEAD_DEFAULTS[kind]
    w = synth.darknet_stream(net, seed=seed, num_classes=ncls, head_gain=hg, obj_bias=0.0)
    model.build(anchors, names, (size, size, 3), dtype=dtype, max_batch=batch, weights=w, streams=streams, max_boxes=max_boxes, **engine_kw)
    # data-dependent objectness prior (uses the product's own forward): a realistic handful of candidates
    # (at the workload's own batch: a rocprofv3 trace of this command then holds launches of that batch only)
    w = synth.calibrate_model(model, synth.synthetic_input(batch, size, size, 3, seed=999), frac)
    return model, w, anchors, ncls


def timed_steps(eng, xs, steps, warmup, threshold, iou_threshold):
    """W untimed + K timed steps of one engine on this process's GPU (no process group): the one-stream comparison legs."""
    import torch
    from tensorflow_yolo_amd.net import dist as ydist
    for i in range(warmup):
        ydist.detect_sharded(eng, xs[i & 1], threshold, iou_threshold)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        _, _, status = ydist.detect_sharded(eng, xs[i & 1], threshold, iou_threshold)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if status.cpu().numpy().any():
        raise RuntimeError("comparison leg: record capacity exceeded")
    return dt


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# the REFERENCE's own decode + NMS functions (net/v2.py:83-119, net/v3.py:109-151, net/base.py:195-209),
# imported and timed in the survey session on that container's host (Intel Xeon @ 2.10 GHz, 8 vCPU, one Python thread,
# NumPy 2.2.6, synthetic randn heads).  Quoted, not re-measured: the reference does not travel to the GPU box.
SURVEY_REFERENCE_DECODE_NMS_MS = {"v2-416": 7.1, "v3-416": 70.4, "v3-608": 168.6}


def cpu_baseline(kind, size, w, anchors, ncls, x, dtype, budget_s=24.0, time_it=True):
    """Oracle forward + decode + NMS on the host cores over the images `x` (the bench's own first batch), never the thing
    shipped.  Returns (cpu_baseline dict or None, oracle fp32 logits of x, e_ref) -- the logits and e_ref (what fp16 storage
    alone does to them, per the oracle: parity.py) feed the parity check of the timed plan."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import to_oracle
    from oracle import decode_ref, forward_ref
    from tensorflow_yolo_amd import YoloV2, YoloV2Tiny, YoloV3
    cls = {"v3": YoloV3, "v2": YoloV2, "v2-tiny": YoloV2Tiny}[kind]
    names = ["c%d" % i for i in range(ncls)]
    net = cls.create_network(np.reshape(anchors, [-1, 2]), names, False, input_shape=(size, size, 3))
    L = to_oracle(net)
    Wd = forward_ref.parse_darknet_weights(L, w)
    # MKL-DNN convs on the 128-core GPU host peak at ~32 threads (16: 3.3, 32: 4.0, 64: 2.2, 128: 0.96 images/s,
    # tools/cpu_threads_probe.py); more threads only add contention, so the baseline uses its best setting.
    default_threads = torch.get_num_threads()
    cores = max(1, min(32, default_threads))
    torch.set_num_threads(cores)
    n_img = x.shape[0]
    chunk = 2 if size >= 416 else 4
    sc = decode_ref.v3_scales(anchors, (size, size)) if kind == "v3" else None

    def decode(logits, full_scan=False):
        if kind == "v3":
            return decode_ref.find_bounding_boxes_v3(logits, 0.5, 0.6, sc, full_scan=full_scan)
        return decode_ref.find_bounding_boxes_v2(logits, 0.5, 0.6, anchors, ncls, full_scan=full_scan)

    forward_ref.forward(L, Wd, x[:min(chunk, n_img)])          # warm-up (thread pool, allocator)
    ref = []
    t_fwd = t_dec = 0.0
    n_timed = 0
    t_all = time.perf_counter()
    for i in range(0, n_img, chunk):                            # first pass: every image once (its logits are the parity reference)
        t0 = time.perf_counter()
        lg = forward_ref.forward(L, Wd, x[i:i + chunk])
        t1 = time.perf_counter()
        decode(lg)
        t2 = time.perf_counter()
        ref.append(lg)
        t_fwd += t1 - t0; t_dec += t2 - t1; n_timed += lg.shape[0]
    while time_it and time.perf_counter() - t_all < 10.0:       # small workloads: keep going until ~10 s of CPU work are on the clock
        for i in range(0, n_img, chunk):
            t0 = time.perf_counter()
            lg = forward_ref.forward(L, Wd, x[i:i + chunk])
            t1 = time.perf_counter()
            decode(lg)
            t2 = time.perf_counter()
            t_fwd += t1 - t0; t_dec += t2 - t1; n_timed += lg.shape[0]
            if time.perf_counter() - t_all > budget_s:
                break
    ref_logits = np.concatenate(ref)
    e_ref = None
    if dtype == "fp16":     # (not part of the timed baseline: the oracle with the HIP path's storage roundings, for the parity bands)
        e_ref = 0.0
        for i in range(0, n_img, chunk):
            l16 = forward_ref.forward(L, Wd, x[i:i + chunk], storage="fp16")
            e_ref = max(e_ref, float(np.max(np.abs(l16.astype(np.float64) - ref_logits[i:i + chunk]))))
    base = None
    if time_it:
        torch.set_num_threads(1)
        # decode + NMS the way the reference runs it: the Python scan over EVERY cell (net/v3.py:113-136), one thread
        k = min(2, n_img)
        t0 = time.perf_counter()
        decode(ref_logits[:k], full_scan=True)
        full_ms = (time.perf_counter() - t0) / k * 1e3
        key = "%s-%d" % ("v2" if kind != "v3" else "v3", size)
        base = {"value": round(n_timed / (t_fwd + t_dec), 3), "unit": "images/sec", "cores": int(cores), "kind": "port",
                "forward_img_s": round(n_timed / t_fwd, 3),
                "decode_nms_ms_per_image": round(full_ms, 2),
                "decode_nms_prefiltered_ms_per_image": round(t_dec / n_timed * 1e3, 3),
                "reference_code_decode_nms_ms_per_image": SURVEY_REFERENCE_DECODE_NMS_MS.get(key if kind != "v2-tiny" else ""),
                "reference_code_source": "SURVEY.md section 6: the reference's own net/v2.py:83-119 / net/v3.py:109-151 + net/base.py:195-209, "
                                         "timed in the survey container (Xeon 2.1 GHz, 1 thread, NumPy 2.2.6); not re-measurable on the GPU box",
                "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
                "sample": "%d images %dx%d (the timed batch, %d distinct), CPU oracle: torch-CPU fp32 restatement of the reference's TF path on %d "
                          "threads (forward_img_s) + NumPy decode/NMS restatement (value = both; decode_nms_ms_per_image = the full per-cell "
                          "Python scan as the reference runs it, 1 thread, %d image(s); *_prefiltered = the oracle's vectorised pre-filter "
                          "form used inside value), not TensorFlow, %.1f s" % (n_timed, size, size, n_img, cores, k, t_fwd + t_dec)}
    torch.set_num_threads(default_threads)
    return base, ref_logits, e_ref


def parity_report(eng, kind, size, anchors, ncls, x, ref_logits, threshold, iou_threshold, dtype, e_ref):
    """The TIMED plan (same engine, same batch size, every image of the batch distinct) vs the CPU fp32 oracle pipeline
    (oracle/parity.py); outside the timed region."""
    from oracle import decode_ref, parity
    from tensorflow_yolo_amd.net import engine as yengine
    got_logits = eng.forward(x).cpu().numpy()
    recs, _ = yengine.records_to_host(*eng.detect(x, threshold, iou_threshold))
    bound = dict(e_ref=e_ref) if dtype == "fp16" else dict(abs_bound=1e-4)
    if kind == "v3":
        rep = parity.check(ref_logits, got_logits, recs, 3, threshold, iou_threshold, scales=decode_ref.v3_scales(anchors, (size, size)), **bound)
    else:
        rep = parity.check(ref_logits, got_logits, recs, 2, threshold, iou_threshold, anchors=anchors, num_classes=ncls, **bound)
    rep["plan"] = "the timed engine: batch %d, %d stream(s)" % (x.shape[0], eng.num_streams)
    return rep


def timed_region(step, steps, warmup, world, device_sync, reduce_device):
    """The bench contract's timed region: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier +
    device synchronise on both sides; the MAX over ranks is the job's time.  Returns (elapsed seconds, result of the last step).
    (A function of its own so that the world-2 gloo test drives the very code the GPU run times: tests/test_dist_cpu.py.)"""
    import torch
    import torch.distributed as dist

    def fence():
        if world > 1:
            dist.barrier()
        device_sync()

    for i in range(warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, last


def contract_fields(rank, world, steps, warmup, batch, elapsed, dtype):
    """The contract's fields of the ONE line rank 0 prints (None on every other rank): whole-job images/sec over all ranks."""
    if rank != 0:
        return None
    return {"metric": "images/sec at 1/2/4/8 MI355X + post-NMS box-set match vs CPU ref",
            "value": round(steps * batch * world / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if dtype == "fp16" else "f32", "data": "synthetic"}


def launch_workers(n):
    """`python bench.py --gpus N` as the driver calls it: start the N ranks as a child torch.distributed.run job.  This
    process has not touched the GPU (torch is not even imported yet), nothing is exec'd; the child's exit code is ours."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="v3-608-b32-fp16", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity check of the timed plan against the CPU oracle")
    ap.add_argument("--force-tile", type=int, default=None, help="tuning hook: yolo_net_options.force_tile (one conv tile id wherever valid)")
    ap.add_argument("--max-boxes", type=int, default=256, help="box records per image (SURVEY 8e: K_max = 256 -> 196.7 KB per rank)")
    ap.add_argument("--autotune", action="store_true", help="time every conv tile per layer on the device first (default: built-in rules)")
    ap.add_argument("--streams", type=int, default=0, help="yolo_net_options.streams: 0 = the library's rule (two half batches on two HIP "
                    "streams where that was measured faster, e.g. the headline workload; config.streams says what ran), 1 = one pass, "
                    "2..4 = that many parts.  The per-kernel roofline figures are always those of whole-batch launches run alone (streams = 1)")
    ap.add_argument("--no-one-stream-leg", action="store_true", help="skip the one-stream comparison legs around the timed region (N = 1, "
                    "only when the timed engine runs on more than one stream)")
    ap.add_argument("--f32-products", type=int, default=0, help="yolo_net_options.f32_products (float32 nets): 0 the library's rule, 1 native float32 MFMA everywhere, 2 nine bf16 products wherever possible")
    ap.add_argument("--threshold", type=float, default=0.5)
    ap.add_argument("--iou-threshold", type=float, default=0.6)
    ap.add_argument("--dump-kernels", default=None, help="write the per-kernel timing table (JSON) here")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:        # before anything touches the GPU
        sys.exit(launch_workers(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d needs GPU %d, torch sees %d device(s): no GPU, no benchmark (there is no CPU path)"
              % (rank, local_rank, torch.cuda.device_count() if torch.cuda.is_available() else 0), file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    kind, size, batch, dtype = WORKLOADS[args.workload]
    model, w, anchors, ncls = make_model(kind, size, batch, dtype, streams=args.streams, max_boxes=args.max_boxes, force_tile=args.force_tile,
                                         f32_products=args.f32_products)
    eng = model.net.engine
    from tensorflow_yolo_amd.net import synth
    # two different resident input batches, alternated, so no step re-reads the previous step's input
    xs_host = [synth.synthetic_input(batch, size, size, 3, seed=1000 + 17 * rank + i) for i in range(2)]
    xs = [torch.from_numpy(x).to(dev) for x in xs_host]
    if args.autotune:
        eng.autotune(xs[0])     # per-layer conv tile choice, timed on this device (outside the timed region)
    from tensorflow_yolo_amd.net import dist as ydist

    # The roofline figures price a kernel by the duration of a whole-batch launch that has the chip to itself: when the timed engine
    # runs the batch as parts on several streams, a second engine of the same weights with streams = 1 serves the instrumented
    # passes (rank 0) -- and, at N = 1, one-stream comparison legs of the same K steps BEFORE and AFTER the timed region.
    def other_engine(streams):
        m1 = type(model)()
        m1.build(anchors, ["c%d" % i for i in range(ncls)], (size, size, 3), dtype=dtype, max_batch=batch, weights=w, streams=streams,
                 max_boxes=args.max_boxes, force_tile=args.force_tile, f32_products=args.f32_products)
        return m1.net.engine

    def one_stream_engine():
        return other_engine(1)

    eng1 = eng
    if world == 1 and eng.num_streams > 1:          # (N > 1: rank 0 builds it after the timed region, nothing beside the ranks' steps)
        eng1 = one_stream_engine()
    # the OTHER way of running the batch, as comparison legs around the timed region (N = 1): one pass when `value` runs two halves, two
    # halves (an explicit choice: half-size arenas) when `value` runs one pass -- whatever the tuner decided on this box, both numbers are
    # in the line.  A batch of one image has no halves.
    eng_other = None
    if world == 1 and not args.no_one_stream_leg:
        if eng.num_streams > 1:
            eng_other = eng1
        elif batch >= 2:
            eng_other = other_engine(2)
            if eng_other.num_streams != 2:
                eng_other = None
    one_stream = None
    legs = eng_other is not None
    if legs:
        one_stream = [timed_steps(eng_other, xs, args.steps, args.warmup, args.threshold, args.iou_threshold)]

    def step(i):
        # forward + decode + NMS of this rank's images (one C call) and, for N > 1, the path's only exchange: ONE all-gather
        # of the fixed-size record buffer (rank order == image order) -- the same function the gloo CPU tests drive
        return ydist.detect_sharded(eng, xs[i & 1], args.threshold, args.iou_threshold)

    elapsed, (boxes, counts, status) = timed_region(step, args.steps, args.warmup, world, torch.cuda.synchronize, dev)
    st = status.cpu().numpy()
    nboxes = counts.cpu().numpy()
    if st.any() and not os.environ.get("YOLO_BENCH_WRONG_RESULTS_OK"):      # (tools/ timing experiments with intentionally wrong kernels only)
        raise RuntimeError("candidate / box-record capacity exceeded during the benchmark: result would not match the reference")
    if legs:
        one_stream.append(timed_steps(eng_other, xs, args.steps, 0, args.threshold, args.iou_threshold))

    out = None
    if rank == 0:
        if eng1 is eng and eng.num_streams > 1:
            eng1 = one_stream_engine()
        total_images = args.steps * batch * world
        # ---- roofline of the dominant kernel family, instrumented steps (hipEvents on the launch stream), one-stream engine
        infos = eng1.kernel_infos()
        reps = max(3, min(10, args.steps))
        ms = np.zeros(eng1.num_kernels, dtype=np.float64)
        for r in range(reps):
            ms += eng1.forward_timed(xs[r & 1])
        ms /= reps
        fam = {}
        for k, ki in enumerate(infos):
            f = fam.setdefault(ki.name.decode(), {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            f["launches"] += 1
            f["ms"] += float(ms[k])
            f["flops"] += ki.flops * batch
            f["bytes"] += ki.bytes * batch + ki.weight_bytes
        dom = max(fam, key=lambda n: fam[n]["ms"])
        d = fam[dom]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        dom_symbol = next(ki.symbol.decode() for ki in infos if ki.name.decode() == dom)
        roof = {"bound": "mfma", "kernel": dom, "kernel_symbol": dom_symbol, "achieved": round(achieved, 2), "peak": PEAK[dtype], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK[dtype], 4), "traffic": None,
                "launches_per_step": d["launches"], "avg_launch_ms": round(d["ms"] / d["launches"], 5),
                "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
                "algorithmic_mb_per_launch": round(d["bytes"] / d["launches"] / 1e6, 3),
                "forward_ms_sum_of_kernels": round(float(ms.sum()), 4),
                "whole_forward_tflops": round(eng1.flops_per_image * batch / (float(ms.sum()) * 1e-3) / 1e12, 2),
                "measured_on": "whole-batch launches, one stream, each kernel alone on the chip (instrumented passes of %s; the same launches "
                               "`bench.py --streams 1` times and profiles/*kernel_stats*.csv hold)" %
                               ("the timed engine" if eng1 is eng else "a streams = 1 engine of the same weights")}
        pk = os.path.join(ROOT, "profiles", "r04_peaks.json")      # measured ceilings of this chip family (tools/probes/peak_probe.hip)
        if os.path.exists(pk):
            try:
                pj = json.load(open(pk))
                roof["sustained_peak"] = pj["mfma_f16_tflops_sustained"] if dtype == "fp16" else pj.get("mfma_f32_tflops_sustained")
                roof["sustained_peak_note"] = pj.get("note")
            except Exception:
                pass
        tj = os.path.join(ROOT, "profiles", "traffic.json")      # HBM bytes per launch from rocprofv3 --pmc passes, if recorded
        if os.path.exists(tj):
            try:
                tw = json.load(open(tj)).get(args.workload, {})
                rec = tw.get(dom_symbol) or tw.get(dom)
                if rec:     # recorded under rocprofv3 --pmc for this kernel family (may predate a retune)
                    roof["traffic"] = rec["hbm_bytes_per_launch"]
                    roof["traffic_detail"] = dict(rec, source="profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)")
            except Exception:
                pass
        if args.dump_kernels:
            rows = [{"kernel": k, "name": ki.name.decode(), "symbol": ki.symbol.decode(), "layer": ki.layer, "k": ki.ksize, "s": ki.stride, "cin": ki.cin,
                     "cout": ki.cout, "out_hw": [ki.out_h, ki.out_w], "ms": round(float(ms[k]), 5),
                     "tflops": round(ki.flops * batch / (ms[k] * 1e-3) / 1e12, 2) if ms[k] > 0 else 0.0,
                     "gbps": round((ki.bytes * batch + ki.weight_bytes) / (ms[k] * 1e-3) / 1e9, 1) if ms[k] > 0 else 0.0}
                    for k, ki in enumerate(infos)]
            with open(args.dump_kernels, "w") as f:
                json.dump({"workload": args.workload, "families": fam, "kernels": rows}, f, indent=1)
        out = contract_fields(rank, world, args.steps, args.warmup, batch, elapsed, dtype)
        out.update({
            "config": {"workload": args.workload, "network": kind, "input": [size, size, 3], "batch_per_gpu": batch,
                       "global_batch": batch * world, "weights": "seeded synthetic Darknet stream (random-init)",
                       "threshold": args.threshold, "iou_threshold": args.iou_threshold, "streams": int(eng.num_streams),
                       "streams_rule": "explicit --streams %d" % args.streams if args.streams > 0 else
                                       "yolo_net_options.streams = 0: the library's rule%s (DESIGN.md, Kernel boundaries)" %
                                       (", then one pass vs two halves timed on this device at the first batch (yolo_net_tune_streams)"
                                        if getattr(eng, "_streams_tuned", False) else ""),
                       "sharding": "images over ranks; all-gather of box records only" if world > 1 else "single GPU",
                       "boxes_per_image_last_step": round(float(nboxes.mean()), 1),
                       "forward_gflop_per_image": round(eng.flops_per_image / 1e9, 3),
                       **({"float32_products": "yolo_net_options.f32_products = %d: %d of %d launches multiply as nine bf16 x bf16 products per float32 product "
                                                "(exact partial products, float32 accumulation; 1 = native float32 MFMA everywhere)"
                                                % (args.f32_products, sum("conv_igemm_emu" in ki.name.decode() for ki in infos), len(infos))}
                          if dtype == "fp32" else {})},
            "roofline": roof,
            "forward_frac_of_mfma_peak": round(eng.flops_per_image * total_images / elapsed / 1e12 / (PEAK[dtype] * world), 4),
        })
        if world == 1:
            mine = "one_stream" if eng.num_streams == 1 else "two_streams"
            out[mine] = {"streams": int(eng.num_streams), "value": out["value"], "unit": "images/sec", "ms_per_step": out["ms_per_step"],
                         "role": "the timed region (`value`)"}
        if legs:
            dt1 = 0.5 * (one_stream[0] + one_stream[1])
            other = "two_streams" if eng.num_streams == 1 else "one_stream"
            out[other] = {"streams": int(eng_other.num_streams), "value": round(args.steps * batch / dt1, 2), "unit": "images/sec",
                          "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                          "ms_per_step_before_after": [round(one_stream[0] / args.steps * 1e3, 4), round(one_stream[1] / args.steps * 1e3, 4)],
                          "role": "comparison legs: same build, weights, inputs and K steps, timed right before and right after the timed "
                                  "region (A / timed / A); `value` above is the %d-stream run" % eng.num_streams}
        out["cpu_baseline"] = out["parity"] = None
        # the CPU baseline is timed on rank 0 at N = 1 only (bench contract); the parity check (rank 0's TIMED engine, every image of
        # its first batch, outside the timed region, before the process group goes away) runs at every N
        time_cpu = world == 1 and not args.no_cpu_baseline
        if time_cpu or not args.no_parity:
            base, ref_logits, e_ref = cpu_baseline(kind, size, w, anchors, ncls, xs_host[0], dtype, time_it=time_cpu)
            out["cpu_baseline"] = base
            if not args.no_parity:
                out["parity"] = parity_report(eng, kind, size, anchors, ncls, xs[0], ref_logits, args.threshold, args.iou_threshold, dtype, e_ref)
        if world > 1:
            out["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": 0, "kind": "port",
                                   "sample": "not timed at N > 1: the CPU baseline is a property of the host, see the N = 1 line"}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
