#!/bin/bash
# per-layer times of every conv tile on the fp16 workloads (input of the cost-model calibration)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/sweep
for wl in ${1:-v3-608-b32-fp16 v3-416-b32-fp16 v2-416-b16-fp16}; do
  for tile in ${2:-0 1 2 3 4 5 6 8 9 10 11 12 13}; do
    YOLO_CONV_TILE=$tile timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-one-stream-leg --dump-kernels gpurun_out/sweep/${wl}_t$tile.json > gpurun_out/sweep/${wl}_t$tile.log 2>&1
    rc=$?
    echo "$wl tile=$tile rc=$rc $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/sweep/${wl}_t$tile.log)"
    if [ $rc -ge 124 ]; then echo "timeout: stopping"; exit $rc; fi
  done
done
