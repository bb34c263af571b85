#!/bin/bash
# tools/pmc_any.sh <tag> <python script> [args]: one rocprofv3 SQ PMC pass (--kernel-trace only, as the pool requires) of an
# arbitrary script; per-kernel means in gpurun_out/<tag>/pmc_sq.csv
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    --output-format csv -d "$OUT/sq" -- python "$ROOT/$1" "${@:2}" > "$OUT/run.log" 2>&1 < /dev/null
f=$(find "$OUT/sq" -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$f" ] && python "$ROOT/tools/pmc_agg.py" "$f" > "$OUT/pmc_sq.csv" && head -12 "$OUT/pmc_sq.csv" | cut -c1-200
rm -rf "$OUT/sq"
