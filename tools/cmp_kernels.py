#!/usr/bin/env python3
"""Per-layer comparison of bench.py --dump-kernels files: tools/cmp_kernels.py A.json B.json [C.json ...]
(rows = kernels of the forward pass, columns = ms per file; grouped totals per (k, s, out_hw) at the end)."""
import json
import sys
files = sys.argv[1:]
D = [json.load(open(f)) for f in files]
rows = D[0]["kernels"]
print("%-4s %-44s %-12s " % ("k", "name", "shape") + " ".join("%10s" % f.split("/")[-1][-10:] for f in files))
groups = {}
for i, r in enumerate(rows):
    ms = [d["kernels"][i]["ms"] if i < len(d["kernels"]) else float("nan") for d in D]
    if not any(ms):
        continue
    key = "%dx%d/%d %d->%d @%d" % (r["k"], r["k"], r["s"], r["cin"], r["cout"], r["out_hw"][0])
    g = groups.setdefault(key, [0] * (len(files) + 1))
    g[0] += 1
    for j, m in enumerate(ms):
        g[j + 1] += m
    print("%-4d %-44s %-12s " % (i, D[0]["kernels"][i]["name"][:44], key) + " ".join("%10.4f" % m for m in ms))
print()
for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%-26s x%-3d " % (key, g[0]) + " ".join("%10.4f" % m for m in g[1:]))
print("%-31s " % "sum" + " ".join("%10.4f" % sum(g[j + 1] for g in groups.values()) for j in range(len(files))))
