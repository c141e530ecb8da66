#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out
for wl in ${1:-v2-416-b16-fp16 tiny-v2-voc-416-b64-fp32 v2-416-b1-fp32 v3-416-b32-fp16}; do
  for at in ${2:-none --autotune}; do
  [ "$at" = "none" ] && at=""
  tag=$wl${at:+_at}
  timeout -k 10 300 python bench.py --workload $wl --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-one-stream-leg --no-parity $at --dump-kernels gpurun_out/kernels_$tag.json > gpurun_out/bench_$tag.log 2>&1
  echo "$tag rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/bench_$tag.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_$tag.log) $(grep -o '"forward_frac_of_mfma_peak": [0-9.]*' gpurun_out/bench_$tag.log)"
  done
done
