#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/gpu_pmc.sh) per (kernel, grid): mean counters + derived ratios."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_r01"
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "yolo" not in name:
            continue
        short = re.sub(r"^_ZN4yolo\d+", "", name)
        short = re.sub(r"EvNS_\d+\w+Params?E$", "", short)
        key = (short[:60], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for key, c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    m["_us"] = sum(dur[key]) / len(dur[key])
    m["_n"] = len(next(iter(c.values())))
    rows.append((key, m))
rows.sort(key=lambda kv: -kv[1]["_us"] * kv[1]["_n"])
out = []
for key, m in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    d = {"kernel": key[0], "grid": key[1], "wg": key[2], "n": m["_n"], "us": round(m["_us"], 1),
         "wait_any/wave_cyc": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
         "wait_inst/wave_cyc": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
         "active_inst/wave_cyc": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
         "mfma_busy_cyc": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "busy_cyc": m.get("SQ_BUSY_CYCLES"),
         "lds_conflict/lds_active": round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1)), 3),
         "FETCH_MB(x2 corrected)": round(2 * m.get("FETCH_SIZE", 0) / 1024, 1), "WRITE_MB": round(m.get("WRITE_SIZE", 0) / 1024, 1),
         "l2_hit": round(m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)), 3),
         "gui_active": m.get("GRBM_GUI_ACTIVE"),
         "insts": {k[9:]: int(m[k]) for k in m if k.startswith("SQ_INSTS_")},
         "wait_inst_lds/wave_cyc": round(m.get("SQ_WAIT_INST_LDS", 0) / wc, 3)}
    out.append(d)
    print(json.dumps(d))
