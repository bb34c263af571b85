#!/bin/bash
# tools/prof_any.sh <tag> <python script> [args]: rocprofv3 kernel stats of an arbitrary script
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1 < /dev/null
F=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv"
find "$OUT/trace" -name "*kernel_trace.csv" -delete 2>/dev/null
tail -3 "$OUT/run.log"
