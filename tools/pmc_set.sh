#!/bin/bash
# tools/pmc_set.sh <tag> "<counters>" <python script> [args]: one rocprofv3 PMC pass with the given counters; per-kernel means
set -u
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d "$OUT/sq" -- python "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1 < /dev/null
f=$(find "$OUT/sq" -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$f" ] && python "$ROOT/tools/pmc_agg.py" "$f" > "$OUT/pmc.csv" && head -2 "$OUT/pmc.csv" | cut -c1-300 || tail -5 "$OUT/run.log"
rm -rf "$OUT/sq"
