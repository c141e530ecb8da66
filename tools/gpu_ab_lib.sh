#!/bin/bash
# A/B of two builds of libyolo_hip.so on one box: $1 = alternative library file (in tree), rest = workloads
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ALT="$1"; shift
LIB=tensorflow-yolo_amd/libyolo_hip.so
cp $LIB /tmp/lib_a.so
for rep in 1 2; do
  cp /tmp/lib_a.so $LIB; echo "== A (tree build)"; bash tools/gpu_workloads.sh "${1:-v3-608-b32-fp16}" none
  cp "$ALT" $LIB;        echo "== B ($ALT)";       bash tools/gpu_workloads.sh "${1:-v3-608-b32-fp16}" none
done
cp /tmp/lib_a.so $LIB
