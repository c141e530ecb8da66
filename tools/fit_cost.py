#!/usr/bin/env python3
"""Table of per-launch times (us) per layer shape and tile from tools/gpu_tile_sweep.sh dumps (cost-model calibration)."""
import glob, json, os, re, sys
from collections import defaultdict
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep"
tab = defaultdict(dict)   # (wl, shape) -> tile name -> us per launch
cnt = {}
for f in sorted(glob.glob(os.path.join(root, "*_t*.json"))):
    wl = os.path.basename(f).rsplit("_t", 1)[0]
    g = defaultdict(lambda: [0, 0.0])
    for k in json.load(open(f))["kernels"]:
        key = (wl, k["k"], k["s"], k["cin"], k["cout"], k["out_hw"][0], k["name"].split("<")[-1].rstrip(">").replace("f16,", ""))
        g[key][0] += 1; g[key][1] += k["ms"]
    for key, (n, ms) in g.items():
        shape = key[:6]
        t = ms / n * 1e3
        tab[shape][key[6]] = min(t, tab[shape].get(key[6], 1e9))
        cnt[shape] = n
tiles = sorted({t for v in tab.values() for t in v})
short = {t: t.replace("K32,S3,x2", "x2").replace(",K64", "k64").replace("tap9", "tap") for t in tiles}
print("%-40s" % "layer" + "".join("%15s" % short[t][-14:] for t in tiles))
for shape in sorted(tab, key=lambda s: (s[0], -s[1], s[2], -s[5])):
    if len(tab[shape]) < 2:
        continue
    best = min(tab[shape].values())
    print("%-16s k%d s%d %4d->%4d @%-3d n=%2d" % (shape + (cnt[shape],)) + "".join(
        ("%14.1f%s" % (tab[shape][t], "*" if tab[shape][t] == best else " ")) if t in tab[shape] else "%15s" % "-" for t in tiles))
