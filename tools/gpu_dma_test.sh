#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for tile in "" 7 5 4; do
  export YOLO_CONV_TILE="$tile"
  [ -z "$tile" ] && unset YOLO_CONV_TILE
  timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py -m gpu -q -p no:cacheprovider -k "fp16" > gpurun_out/pytest_tile_${tile:-auto}.log 2>&1
  rc=$?
  echo "tile=${tile:-auto} rc=$rc: $(tail -n 1 gpurun_out/pytest_tile_${tile:-auto}.log)"
  if [ $rc -ge 124 ]; then echo "timeout/killed: stopping"; exit $rc; fi
done
unset YOLO_CONV_TILE
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-kernels gpurun_out/kernels_dma.json > gpurun_out/bench_dma.log 2>&1
echo "bench rc=$?"; tail -n 2 gpurun_out/bench_dma.log | cut -c1-600
