#!/bin/bash
# one full bench.py line (parity + cpu baseline + roofline.traffic from profiles/traffic.json) per BASELINE.json workload -> gpurun_out/<tag>_all_workloads.jsonl
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG="${1:-r03}"
: > gpurun_out/${TAG}_all_workloads.jsonl
for wl in ${2:-v3-608-b32-fp16 v2-416-b16-fp16 tiny-v2-voc-416-b64-fp32 v2-416-b1-fp32 v3-416-b32-fp16 v3-608-b8-fp16 v3-608-b1-fp16}; do
  timeout -k 10 400 python bench.py --workload $wl --steps 20 --warmup 5 --dump-kernels gpurun_out/${TAG}_kernels_$wl.json > gpurun_out/${TAG}_bench_$wl.log 2>&1
  rc=$?
  tail -1 gpurun_out/${TAG}_bench_$wl.log >> gpurun_out/${TAG}_all_workloads.jsonl
  echo "$wl rc=$rc $(tail -1 gpurun_out/${TAG}_bench_$wl.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["forward_frac_of_mfma_peak"], d["roofline"]["frac"], d["roofline"]["traffic"], d["parity"]["boxes_unexplained"], d["parity"]["max_abs_logit_err"])' 2>/dev/null)"
  if [ $rc -ge 124 ]; then echo "timeout: stopping"; exit $rc; fi
done
