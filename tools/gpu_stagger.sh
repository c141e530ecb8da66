#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export PYTHONUNBUFFERED=1
for f in 0 0.3 0.5 0.8; do
  echo "== stagger $f"
  YOLO_CONV_STAGGER=$f timeout -k 10 200 python tools/ablate.py 2>&1 | grep -v "Weights ready\|amdgpu.ids" | head -5
done
