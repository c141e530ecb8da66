#!/bin/bash
# rocprofv3 evidence for every BASELINE.json workload: kernel-trace stats + PMC passes per workload, runs without the
# batch-2 parity pass (--no-parity) so only the workload's own launches are profiled.  $1 = round tag (r03), $2 = workloads
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG="${1:-r03}"
for wl in ${2:-v3-608-b32-fp16 v2-416-b16-fp16 tiny-v2-voc-416-b64-fp32}; do
  echo "=== $wl: kernel trace"
  bash tools/gpu_prof.sh ${TAG}_prof_$wl --workload $wl --no-parity > gpurun_out/${TAG}_prof_$wl.log 2>&1 || { echo "prof failed"; tail -5 gpurun_out/${TAG}_prof_$wl.log; }
  tail -1 gpurun_out/${TAG}_prof_$wl.log | cut -c1-120
  echo "=== $wl: pmc"
  bash tools/gpu_pmc.sh ${TAG}_pmc_$wl --workload $wl --no-parity > gpurun_out/${TAG}_pmc_$wl.log 2>&1
  rc=$?
  grep "rc=" gpurun_out/${TAG}_pmc_$wl.log
  if [ $rc -ge 124 ]; then echo "pmc timed out: stopping"; exit $rc; fi
  # keep only what the summaries need (the raw CSVs are large)
  # profiles/traffic.json travels with the snapshot (gpurun_out/ does not): update it in place, then copy it out for the merge back
  python3 tools/pmc_traffic.py gpurun_out/${TAG}_pmc_$wl $wl profiles/traffic.json > gpurun_out/${TAG}_traffic_$wl.txt 2>&1
  cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
  python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_$wl 24 > gpurun_out/${TAG}_pmc_summary_$wl.jsonl 2>&1
  f=$(find gpurun_out/${TAG}_prof_$wl -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_$wl.csv
  cp gpurun_out/${TAG}_prof_$wl/bench_under_rocprof.log gpurun_out/${TAG}_bench_under_rocprof_$wl.log 2>/dev/null
  rm -rf gpurun_out/${TAG}_pmc_$wl gpurun_out/${TAG}_prof_$wl
done
ls -la gpurun_out | grep ${TAG}_ | head -40
