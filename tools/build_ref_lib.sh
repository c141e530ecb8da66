#!/bin/bash
# Builds libyolo_hip.so of a git revision into build_ab/<name>.so (for interleaved A/B runs on one box: tools/gpu_arms.sh).
#   tools/build_ref_lib.sh [rev = HEAD] [name = lib_prev]
set -eu
cd "$(dirname "$0")/.."
REV="${1:-HEAD}"; NAME="${2:-lib_prev}"
WT=/tmp/yolo_ref_wt
rm -rf $WT; git worktree prune
git worktree add -f $WT "$REV" > /dev/null 2>&1
make -C $WT/tensorflow-yolo_amd/csrc -j6 > /dev/null 2>&1
mkdir -p build_ab
cp $WT/tensorflow-yolo_amd/libyolo_hip.so build_ab/$NAME.so
git worktree remove --force $WT
ls -la build_ab/$NAME.so
