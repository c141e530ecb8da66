#!/bin/bash
# rocprofv3 PMC passes over a short bench run (separate passes, kernel-trace only: no sys/hip trace).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG="${1:-pmc}"; shift || true
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/pass$i" -o p -- python3 "$ROOTDIR/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-one-stream-leg --streams 1 "$@" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc ($ctrs)"
  if [ $rc -ge 124 ]; then echo "timeout: stopping"; exit $rc; fi
done
cd "$ROOTDIR"
find "$OUT" -name "*.csv" | head; du -sh "$OUT"
