#!/bin/bash
# parity evidence of the round: engine + training GPU tests, randomised sweeps, the per-stage error log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_parity
mkdir -p "$OUT"
cd "$ROOT"
rm -f gpurun_out/parity_stage_errors.jsonl
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_training_gpu.py tests/test_spline_backward_gpu.py -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -15 "$OUT/pytest.log"
cp gpurun_out/parity_stage_errors.jsonl "$OUT/" 2>/dev/null
timeout 1500 python tools/train_parity_sweep.py 12 500 > "$OUT/train_sweep.log" 2>&1; tail -14 "$OUT/train_sweep.log"
timeout 1500 python tools/parity_sweep.py 24 3000 > "$OUT/sweep.log" 2>&1; tail -6 "$OUT/sweep.log"
