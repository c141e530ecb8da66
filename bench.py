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
  "parity":       the metric's second half, "post-NMS box-set match vs CPU ref": two images through the HIP path and
                  through the CPU fp32 oracle pipeline OUTSIDE the timed region -- max |logit - oracle logit| and the
                  post-NMS box sets compared under the margin rule of oracle/parity.py
  "roofline":     dominant kernel family (the implicit-GEMM conv tile with most device time): algorithmic FLOPs of its launches
                  / their device time measured with hipEvents on the launch stream (instrumented steps
                  run right after the timed region; the events add bubbles so they never time `value`)
  "cpu_baseline": the CPU oracle (torch-CPU restatement of the reference's TF path + NumPy decode/NMS,
                  kind "port") timed on the host cores on a bounded sample of the same workload.
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


def two_stream_leg(kind, size, batch, dtype, w, xs, args):
    """The same K steps with the batch run as two independent halves on two HIP streams (yolo_net_options.streams = 2,
    DESIGN.md "Kernel boundaries"): the kernels' tails and launch boundaries of one half overlap the other half's kernels.
    Reported BESIDE `value` (which stays the one-stream number the per-launch `roofline` figures belong to)."""
    import torch
    from tensorflow_yolo_amd import YoloV2, YoloV2Tiny, YoloV3
    from tensorflow_yolo_amd.net import dist as ydist
    cls, anchors, ncls = {"v3": (YoloV3, COCO_V3, 80), "v2": (YoloV2, COCO_V2, 80), "v2-tiny": (YoloV2Tiny, VOC_TINY, 20)}[kind]
    model = cls()
    model.build(anchors, ["c%d" % i for i in range(ncls)], (size, size, 3), dtype=dtype, max_batch=batch, weights=w, streams=2,
                max_boxes=args.max_boxes)
    eng = model.net.engine
    for i in range(args.warmup):
        ydist.detect_sharded(eng, xs[i & 1], args.threshold, args.iou_threshold)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        _, _, status = ydist.detect_sharded(eng, xs[i & 1], args.threshold, args.iou_threshold)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if status.cpu().numpy().any():
        raise RuntimeError("two-stream leg: record capacity exceeded")
    return {"streams": 2, "value": round(args.steps * batch / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / args.steps * 1e3, 4),
            "note": "same build, same inputs and K steps, batch as two halves on two HIP streams (opt-in mode; `value` is the one-stream run)"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(kind, size, w, anchors, ncls, budget_s=20.0, time_it=True):
    """Oracle forward + decode + NMS on the host cores, bounded sample (never the thing shipped).
    Returns (cpu_baseline dict or None, x [2 images], oracle fp32 logits of x) -- the logits feed the parity check."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import to_oracle
    from oracle import decode_ref, forward_ref
    from tensorflow_yolo_amd import YoloV2, YoloV2Tiny, YoloV3
    from tensorflow_yolo_amd.net import synth
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
    chunk = 2
    x = synth.synthetic_input(chunk, size, size, 3, seed=123)

    def one():
        logits = forward_ref.forward(L, Wd, x)
        if kind == "v3":
            sc = decode_ref.v3_scales(anchors, (size, size))
            decode_ref.find_bounding_boxes_v3(logits, 0.5, 0.6, sc)
        else:
            decode_ref.find_bounding_boxes_v2(logits, 0.5, 0.6, anchors, ncls)
        return logits

    ref_logits = one()                      # warm-up (thread pool, allocator); its logits are the parity reference
    if not time_it:
        torch.set_num_threads(default_threads)
        return None, x, ref_logits
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += chunk
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 64:
            break
    torch.set_num_threads(default_threads)
    return ({"value": round(n / dt, 3), "unit": "images/sec", "cores": int(cores), "kind": "port",
             "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
             "sample": "%d images %dx%d, CPU oracle (torch-CPU fp32 restatement of the reference's TF path + NumPy decode/NMS, "
                       "not TensorFlow), %.1f s" % (n, size, size, dt)}, x, ref_logits)


def parity_report(model, kind, size, anchors, ncls, x, ref_logits, threshold, iou_threshold):
    """HIP path vs the CPU fp32 oracle pipeline on the images `x` (oracle/parity.py); outside the timed region."""
    from oracle import decode_ref, parity
    from tensorflow_yolo_amd.net import engine as yengine
    eng = model.net.engine
    got_logits = eng.forward(x).cpu().numpy()
    recs, _ = yengine.records_to_host(*eng.detect(x, threshold, iou_threshold))
    if kind == "v3":
        return parity.check(ref_logits, got_logits, recs, 3, threshold, iou_threshold, scales=decode_ref.v3_scales(anchors, (size, size)))
    return parity.check(ref_logits, got_logits, recs, 2, threshold, iou_threshold, anchors=anchors, num_classes=ncls)


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
    ap.add_argument("--no-parity", action="store_true", help="skip the 2-image parity check against the CPU oracle")
    ap.add_argument("--force-tile", type=int, default=None, help="tuning hook: yolo_net_options.force_tile (one conv tile id wherever valid)")
    ap.add_argument("--max-boxes", type=int, default=256, help="box records per image (SURVEY 8e: K_max = 256 -> 196.7 KB per rank)")
    ap.add_argument("--autotune", action="store_true", help="time every conv tile per layer on the device first (default: built-in rules)")
    ap.add_argument("--streams", type=int, default=0, help="run every batch as this many independent parts on as many HIP streams "
                    "(overlaps the kernels' tails; the per-kernel roofline figures then describe one part's launches run alone)")
    ap.add_argument("--no-two-stream-leg", action="store_true", help="skip the extra timed leg with --streams 2 (N = 1 only)")
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
    model, w, anchors, ncls = make_model(kind, size, batch, dtype, streams=args.streams, max_boxes=args.max_boxes, force_tile=args.force_tile)
    eng = model.net.engine
    from tensorflow_yolo_amd.net import synth
    # two different resident input batches, alternated, so no step re-reads the previous step's input
    xs = [torch.from_numpy(synth.synthetic_input(batch, size, size, 3, seed=1000 + 17 * rank + i)).to(dev) for i in range(2)]
    if args.autotune:
        eng.autotune(xs[0])     # per-layer conv tile choice, timed on this device (outside the timed region)
    from tensorflow_yolo_amd.net import dist as ydist

    def step(i):
        # forward + decode + NMS of this rank's images (one C call) and, for N > 1, the path's only exchange: ONE all-gather
        # of the fixed-size record buffer (rank order == image order) -- the same function the gloo CPU tests drive
        return ydist.detect_sharded(eng, xs[i & 1], args.threshold, args.iou_threshold)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        boxes, counts, status = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = status.cpu().numpy()
    nboxes = counts.cpu().numpy()
    if st.any():
        raise RuntimeError("candidate / box-record capacity exceeded during the benchmark: result would not match the reference")

    out = None
    if rank == 0:
        total_images = args.steps * batch * world
        value = total_images / elapsed
        # ---- roofline of the dominant kernel family, instrumented steps (hipEvents on the launch stream)
        infos = eng.kernel_infos()
        reps = max(3, min(10, args.steps))
        ms = np.zeros(eng.num_kernels, dtype=np.float64)
        for r in range(reps):
            ms += eng.forward_timed(xs[r & 1])
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
                "whole_forward_tflops": round(eng.flops_per_image * batch / (float(ms.sum()) * 1e-3) / 1e12, 2)}
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
        out = {
            "metric": "images/sec at 1/2/4/8 MI355X + post-NMS box-set match vs CPU ref",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16" if dtype == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": args.workload, "network": kind, "input": [size, size, 3], "batch_per_gpu": batch,
                       "global_batch": batch * world, "weights": "seeded synthetic Darknet stream (random-init)",
                       "threshold": args.threshold, "iou_threshold": args.iou_threshold, "streams": max(1, args.streams),
                       "sharding": "images over ranks; all-gather of box records only" if world > 1 else "single GPU",
                       "boxes_per_image_last_step": round(float(nboxes.mean()), 1),
                       "forward_gflop_per_image": round(eng.flops_per_image / 1e9, 3)},
            "roofline": roof,
            "forward_frac_of_mfma_peak": round(eng.flops_per_image * total_images / elapsed / 1e12 / (PEAK[dtype] * world), 4),
        }
        if world == 1 and args.streams == 0 and batch >= 2 and not args.no_two_stream_leg:
            out["two_streams"] = two_stream_leg(kind, size, batch, dtype, w, xs, args)
        out["cpu_baseline"] = out["parity"] = None
        # the CPU baseline is timed on rank 0 at N = 1 only (bench contract); the parity check (rank 0's engine, two images,
        # outside the timed region, before the process group goes away) runs at every N
        time_cpu = world == 1 and not args.no_cpu_baseline
        if time_cpu or not args.no_parity:
            base, xb, ref_logits = cpu_baseline(kind, size, w, anchors, ncls, time_it=time_cpu)
            out["cpu_baseline"] = base
            if not args.no_parity:
                nb = min(batch, xb.shape[0])        # (a batch-1 workload checks one image)
                out["parity"] = parity_report(model, kind, size, anchors, ncls, xb[:nb], ref_logits[:nb], args.threshold, args.iou_threshold)
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
