#!/bin/bash
# Interleaved comparison of several ARMS on ONE box (never rank builds across boxes: +-5 % box to box).
#   tools/gpu_arms.sh ROUNDS "name|lib.so or -|ENV=1 ENV2=x|extra bench args" ["name2|..."] -- [bench.py args for every arm]
# An arm = a library file in the tree (- = the tree build) + environment switches read by the library.  Per arm and round one
# bench.py run (20 steps, per-kernel dump gpurun_out/arms_<name>_<round>.json); prints img/s, ms/step and the sum of kernel times.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROUNDS="$1"; shift
ARMS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done
[ $# -gt 0 ] && shift
LIB=tensorflow-yolo_amd/libyolo_hip.so
mkdir -p gpurun_out
cp $LIB /tmp/lib_tree.so
trap 'cp /tmp/lib_tree.so $LIB' EXIT
for rep in $(seq 1 $ROUNDS); do
  for arm in "${ARMS[@]}"; do
    IFS='|' read -r name lib envs extra <<< "$arm"
    if [ "$lib" = "-" ] || [ -z "$lib" ]; then cp /tmp/lib_tree.so $LIB; else cp "$lib" $LIB; fi
    env $envs timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-one-stream-leg --no-parity --dump-kernels gpurun_out/arms_${name}_$rep.json $extra "$@" > gpurun_out/arms_${name}_$rep.log 2>&1
    rc=$?
    echo "$name round $rep rc=$rc $(tail -1 gpurun_out/arms_${name}_$rep.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["forward_ms_sum_of_kernels"])' 2>/dev/null)"
    if [ $rc -ge 124 ]; then exit $rc; fi
  done
done
