#!/bin/bash
# Runs on the GPU box (via gpurun): GPU test-suite, then a short bench; logs under gpurun_out/.
# A step that times out or dies with a signal stops the script (no further GPU work after a hang).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
TESTS="${1:-tests}"
STEPS="${2:-10}"
timeout -k 10 1000 python -m pytest $TESTS -m gpu -q -s -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -n 40 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then echo "pytest timed out or was killed: stopping"; exit $rc; fi
timeout -k 10 600 python bench.py --steps $STEPS --warmup 3 --dump-kernels gpurun_out/kernels.json > gpurun_out/bench.log 2>&1
brc=$?
echo "bench rc=$brc"
tail -n 5 gpurun_out/bench.log
exit $(( rc != 0 ? rc : brc ))
