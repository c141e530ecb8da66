#!/bin/bash
# rocprofv3 kernel-trace + stats of a short bench run (on the GPU box, via gpurun).  $1 = tag, rest = bench args
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG="${1:-prof}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1
ROOTDIR=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOTDIR/$OUT" -o trace -- python3 "$ROOTDIR/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-one-stream-leg --streams 1 "$@" > "$ROOTDIR/$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof rc=$?"
cd "$ROOTDIR"
find "$OUT" -name "*kernel_stats*.csv" | head -3
f=$(find "$OUT" -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -20 "$f"
# the bench line alone, as JSON (the log also holds rocprofv3's own messages): what profiles/*_bench_under_rocprof.json is copied from
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json"
python3 -c "import json,sys; json.load(open('$OUT/bench_under_rocprof.json')); print('bench line: valid JSON')"
tail -2 "$OUT/bench_under_rocprof.log"
