#!/bin/bash
# tools/pmc.sh <tag> -- rocprofv3 PMC passes (own runs, --kernel-trace only, as the pool requires) of a short
# bench.py run; writes gpurun_out/<tag>/pmc_<pass>.csv (per-dispatch counter rows, aggregated by tools/pmc_agg.py).
set -u
TAG=${1:-pmc}; shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python "$ROOT/bench.py" --no-cpu-baseline --no-latency --no-side-legs --steps 3 --warmup 1 --engines 1 $EXTRA > "$OUT/$name.log" 2>&1 < /dev/null
  echo "pass $name rc=$?"
  local f=$(find "$OUT/$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$ROOT/tools/pmc_agg.py" "$f" > "$OUT/pmc_$name.csv"; head -14 "$OUT/pmc_$name.csv" | cut -c1-220; else echo "no counter csv"; tail -5 "$OUT/$name.log"; fi
  rm -rf "$OUT/$name"
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
