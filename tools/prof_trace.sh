#!/bin/bash
# tools/prof_trace.sh <tag> <python script>: rocprofv3 kernel trace of a short script, raw trace kept (gpurun_out/<tag>/kernel_trace.csv)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1 < /dev/null
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_trace.csv" && ls -la "$OUT/kernel_trace.csv"
rm -rf "$OUT/trace"
tail -2 "$OUT/run.log"
