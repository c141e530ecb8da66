#!/bin/bash
# Interleaved A/B of two builds of libyolo_hip.so on ONE box (rule: never rank builds across boxes, +-5 % box to box).
#   tools/gpu_ab.sh <alt .so in tree> [rounds] [bench args...]      A = tree build, B = alt
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ALT="$1"; ROUNDS="${2:-3}"; shift; shift || true
LIB=tensorflow-yolo_amd/libyolo_hip.so
mkdir -p gpurun_out
cp $LIB /tmp/lib_a.so
trap 'cp /tmp/lib_a.so $LIB' EXIT
for rep in $(seq 1 $ROUNDS); do
  for v in A B; do
    if [ $v = A ]; then cp /tmp/lib_a.so $LIB; else cp "$ALT" $LIB; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-stream-leg --no-parity --dump-kernels gpurun_out/ab_${v}_$rep.json "$@" > gpurun_out/ab_${v}_$rep.log 2>&1
    rc=$?
    echo "$v round $rep rc=$rc $(tail -1 gpurun_out/ab_${v}_$rep.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["forward_ms_sum_of_kernels"])' 2>/dev/null)"
    if [ $rc -ge 124 ]; then cp /tmp/lib_a.so $LIB; exit $rc; fi
  done
done
cp /tmp/lib_a.so $LIB
