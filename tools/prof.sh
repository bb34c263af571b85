#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 kernel-trace + stats of bench.py into gpurun_out/<tag>/
# (run through gpurun; every step is time-bounded and detached from stdin).
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python "$ROOT/bench.py" --no-cpu-baseline --no-latency --no-side-legs "$@" > "$OUT/bench.log" 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
F=$(find "$OUT/trace" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then cp "$F" "$OUT/kernel_stats.csv"; head -45 "$OUT/kernel_stats.csv"; else echo "no kernel_stats.csv"; tail -20 "$OUT/bench.log"; fi
# keep the merged payload small: drop the raw traces
find "$OUT/trace" -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
tail -2 "$OUT/bench.log"
