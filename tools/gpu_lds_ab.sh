#!/bin/bash
# LDS counters of one kernel for the tree build (A) and an alternative build (B) of libyolo_hip.so, same box.  $1 = alt .so, $2 = kernel name substring
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ALT="$1"; KERN="${2:-stem_v3}"
ROOTDIR=$(pwd)
LIB=tensorflow-yolo_amd/libyolo_hip.so
mkdir -p gpurun_out/ldsab
cp $LIB /tmp/lib_a.so
trap 'cp /tmp/lib_a.so "$ROOTDIR/$LIB"' EXIT      # the tree build comes back whatever ends the script
(rocprofv3 --list-avail 2>/dev/null || rocprofv3 -L 2>/dev/null) | grep -i "lds" | head -60 > gpurun_out/ldsab/avail.txt
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  if [ $v = A ]; then cp /tmp/lib_a.so $ROOTDIR/$LIB; else cp "$ROOTDIR/$ALT" $ROOTDIR/$LIB; fi
  i=0
  # (round 5: $3 / $4 = other counter sets for the two passes, e.g. the wave-state counters: "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY ...")
  P1="${3:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM}"
  P2="${4:-SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_STALL}"
  for ctrs in "$P1" "$P2"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/ldsab_$v/pass$i -o p -- python3 "$ROOTDIR/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-one-stream-leg --streams 1 --no-parity > $ROOTDIR/gpurun_out/ldsab/${v}_pass$i.log 2>&1
    echo "$v pass $i rc=$?"
  done
  python3 - "$v" "$KERN" <<'PY' > $ROOTDIR/gpurun_out/ldsab/$v.txt
import csv, glob, sys
from collections import defaultdict
v, kern = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob("/tmp/ldsab_%s/pass*/*counter_collection.csv" % v):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, sum(acc[k]) / len(acc[k]), len(acc[k]))
PY
  cat $ROOTDIR/gpurun_out/ldsab/$v.txt
done
cp /tmp/lib_a.so $ROOTDIR/$LIB
