#!/usr/bin/env python3
"""Compare per-layer-shape times between --dump-kernels files: python tools/cmp_tiles.py a.json b.json ..."""
import json, sys
from collections import defaultdict
cols = []
for f in sys.argv[1:]:
    g = defaultdict(lambda: [0, 0.0, ""])
    for k in json.load(open(f))["kernels"]:
        key = (k["k"], k["s"], k["cin"], k["cout"], tuple(k["out_hw"]))
        g[key][0] += 1; g[key][1] += k["ms"]; g[key][2] = k["name"].split("<")[-1].rstrip(">")
    cols.append(g)
keys = sorted(cols[0], key=lambda k: -cols[0][k][1])
print("%-34s" % "layer shape" + "".join("%24s" % f.split("kernels_")[-1][:22] for f in sys.argv[1:]))
for key in keys:
    print("k%d s%d %4d->%4d @%-9s n=%2d " % (key[0], key[1], key[2], key[3], "x".join(map(str, key[4])), cols[0][key][0])
          + "".join("%9.3f %-14s" % (c[key][1], c[key][2][-14:]) for c in cols))
print("%-34s" % "total" + "".join("%9.3f %-14s" % (sum(v[1] for v in c.values()), "") for c in cols))
