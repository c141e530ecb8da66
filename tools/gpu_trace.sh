#!/bin/bash
# block trace (tools/trace_blocks.py) with the experiment build of the library; $1 = TRACE_ONLY filter (k1 | s2 | tap | "")
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
LIB=tensorflow-yolo_amd/libyolo_hip.so
EXP=tensorflow-yolo_amd/libyolo_hip_exp.so
[ -f $EXP ] || { echo "build the experiment library first (make EXPERIMENT=1 OBJDIR=... OUT=$EXP)"; exit 1; }
mkdir -p gpurun_out
cp $LIB /tmp/lib_prod.so
trap 'cp /tmp/lib_prod.so $LIB' EXIT      # the product build comes back whatever ends the script
cp $EXP $LIB
TRACE_ONLY="${1:-}" TRACE_BINS="${2:-21}" timeout -k 10 300 python tools/trace_blocks.py > gpurun_out/trace_${1:-all}.txt 2>&1
rc=$?
cp /tmp/lib_prod.so $LIB
echo "trace rc=$rc"; tail -5 gpurun_out/trace_${1:-all}.txt
exit $rc
