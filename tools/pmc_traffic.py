#!/usr/bin/env python3
"""rocprofv3 --pmc passes (tools/gpu_pmc.sh) -> profiles/traffic.json: HBM bytes per launch of every kernel
symbol (the key bench.py looks up: roofline.kernel_symbol) (FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM': on gfx950 it reports half the bytes of a
wide coalesced stream; WRITE_SIZE as is; both in KiB)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "v3-608-b32-fp16"
out_path = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"


def family(name):
    """key = the kernel's name as rocprofv3 prints it = yolo_kernel_info.symbol = bench.py's roofline.kernel_symbol"""
    return name if "yolo::" in name else None


acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
    if not rows:
        continue
    # keep only the launches of the LAST forward pass of the run (between the last two nms_kernel dispatches): the
    # earlier ones include the small batch-2 calibration net
    order = sorted({int(r["Dispatch_Id"]): r["Kernel_Name"] for r in rows}.items())
    nms = [d for d, n in order if "nms_kernel" in n]
    lo, hi = (nms[-2], nms[-1]) if len(nms) >= 2 else (-1, 1 << 62)
    for r in rows:
        fam = family(r["Kernel_Name"])
        if fam and lo < int(r["Dispatch_Id"]) < hi:
            acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for fam, c in acc.items():
    fetch = 2.0 * 1024 * sum(c.get("FETCH_SIZE", [0])) / max(1, len(c.get("FETCH_SIZE", [0])))
    write = 1024.0 * sum(c.get("WRITE_SIZE", [0])) / max(1, len(c.get("WRITE_SIZE", [0])))
    res[fam] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes_per_launch": round(fetch), "write_bytes_per_launch": round(write),
                "launches_sampled": len(c.get("FETCH_SIZE", []))}
prev = json.load(open(out_path)) if os.path.exists(out_path) else {}
prev[workload] = res
prev["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), averaged over the launches of each "
                 "kernel family inside one forward pass of the named workload; FETCH_SIZE x2 per the gfx950 correction")
json.dump(prev, open(out_path, "w"), indent=1, sort_keys=True)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
    print("%-44s fetch %8.1f MB  write %8.1f MB  (n=%d)" % (k, v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["launches_sampled"]))
