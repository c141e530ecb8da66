#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONUNBUFFERED=1
for dbg in 0 4 1 2 3 7 5 6; do
  YOLO_CONV_DBG=$dbg timeout -k 10 200 python tools/ablate.py 2>&1 | grep "DBG\|@76 \|@38 \|k3 s1 512->1024"
done
