"""Chained launches (csrc/conv_chain.hip) against the separate launches of the same build: head logits of one YOLOv3-608 batch, run in two
processes (the switch YOLO_NO_CHAIN is read once per process).  python tools/chain_probe.py [batch]"""
import os, subprocess, sys
import numpy as np

def child(tag):
    sys.path.insert(0, os.getcwd())
    import torch, bench
    from tensorflow_yolo_amd.net import synth
    batch = int(os.environ.get("PROBE_BATCH", "32"))
    kind, size, _, dtype = bench.WORKLOADS["v3-608-b32-fp16"]
    model, w, anchors, ncls = bench.make_model(kind, size, batch, dtype, streams=int(os.environ.get("PROBE_STREAMS", "1")))
    eng = model.net.engine
    x = torch.from_numpy(synth.synthetic_input(batch, size, size, 3, seed=1000)).cuda()
    out = eng.forward(x).cpu().numpy()
    out2 = eng.forward(x).cpu().numpy()
    np.save("/tmp/chain_probe_%s.npy" % tag, out)
    print(tag, "second forward identical:", bool(np.array_equal(out, out2)), "finite:", bool(np.isfinite(out).all()))
    for name, off, used, size in eng.workspace_regions():
        if "chained" in name:
            ctrl = eng._workspace[off:off + 8192].view(torch.int32).cpu().numpy()
            print(tag, "chain region at", off, "heads", ctrl[0:256:32], "exited", ctrl[256], "timeout", ctrl[257], "nonzero done", int(np.count_nonzero(ctrl[288:])))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("chain", "sep"):
        child(sys.argv[1])
        sys.exit(0)
    os.makedirs("gpurun_out", exist_ok=True)
    for tag in ("sep", "chain"):
        env = dict(os.environ)
        if tag == "sep": env["YOLO_NO_CHAIN"] = "1"
        r = subprocess.run([sys.executable, __file__, tag], env=env, timeout=280)
        if r.returncode: sys.exit("child %s failed" % tag)
    a = np.load("/tmp/chain_probe_sep.npy"); b = np.load("/tmp/chain_probe_chain.npy")
    print("shape", a.shape, "equal", bool(np.array_equal(a, b)), "max abs diff", float(np.max(np.abs(a - b))))
    rows = [(0, 1083), (1083, 5415), (5415, 22743)]
    for i, (lo, hi) in enumerate(rows):
        d = np.abs(a[:, lo:hi] - b[:, lo:hi]).max(axis=(1, 2))
        print("scale", i, "per-image max diff", np.array2string(d, precision=3))
