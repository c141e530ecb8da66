#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage, demangled).
    python tools/resusage.py tensorflow-yolo_amd/csrc/conv_tap.hip [extra hipcc flags]"""
import os, re, subprocess, sys
src = os.path.abspath(sys.argv[1])
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + inc, "-c", src, "-o", "/tmp/resusage.o",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:], capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(": ", 1)[1]
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for (k, r), d in zip(rows.items(), dem):
    print("%-88s VGPR %4s AGPR %3s spill %3s scratch %4s SGPR %3s occ %s LDS %s" % (d[:88], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("TotalSGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
