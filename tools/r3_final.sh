#!/bin/bash
# end-of-round evidence: full GPU test-suite, default bench line, kernel stats of the driver's command, PMC traffic
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_final
mkdir -p "$OUT"
cd "$ROOT"
rm -f gpurun_out/parity_stage_errors.jsonl
timeout 2400 python -m pytest tests/ -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
tail -4 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_stage_errors.jsonl "$OUT/" 2>/dev/null
timeout 1200 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 1500 "$OUT/bench_default.json"; tail -3 "$OUT/bench_default.err"
