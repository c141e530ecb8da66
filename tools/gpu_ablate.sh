#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONUNBUFFERED=1
for dbg in ${1:-0 8}; do
  for tile in ${2:-""}; do
  echo "== dbg=$dbg tile=${tile:-auto}"
  YOLO_CONV_TILE=$tile YOLO_CONV_DBG=$dbg timeout -k 10 200 python tools/ablate.py 2>&1 | grep "DBG\|@76 \|@38 \|k3 s1 512->1024\|@152"
  done
done
