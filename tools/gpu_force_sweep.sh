#!/bin/bash
# per-layer times with one conv tile id forced wherever valid (product build: bench.py --force-tile); $1 = workloads, $2 = tile ids (d = default rules)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/sweep
for wl in ${1:-v3-608-b32-fp16}; do
  for tile in ${2:-d 1 2 3 4 5 6 7 14}; do
    FT=""; [ "$tile" != "d" ] && FT="--force-tile $tile"
    timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-one-stream-leg --no-parity $FT --dump-kernels gpurun_out/sweep/kernels_${wl}_t$tile.json > gpurun_out/sweep/${wl}_t$tile.log 2>&1
    rc=$?
    echo "$wl tile=$tile rc=$rc $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/sweep/${wl}_t$tile.log)"
    if [ $rc -ge 124 ]; then echo "timeout: stopping"; exit $rc; fi
  done
done
