#!/bin/bash
# randomised parity sweeps on the final build (fewer cases than round 3's 24 + 12: the oracle's CPU time is box time)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4_sweeps
mkdir -p "$OUT"
cd "$ROOT"
timeout 500 python tools/train_parity_sweep.py ${1:-8} 500 > "$OUT/train_sweep.log" 2>&1; tail -4 "$OUT/train_sweep.log"
timeout 700 python tools/parity_sweep.py ${2:-10} 3000 > "$OUT/sweep.log" 2>&1; tail -4 "$OUT/sweep.log"
