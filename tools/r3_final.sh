#!/bin/bash
# end-of-round evidence in ONE box (so that the bench line can carry the PMC traffic of the build it runs on): PMC passes ->
# profiles/r3_traffic.json (stamped), kernel stats of the three bench commands + the S-edges probe, full GPU test-suite,
# default bench line.  Everything lands in gpurun_out/ (merged back); tools/r3_collect.sh copies it into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_final
mkdir -p "$OUT"
cd "$ROOT"
bash tools/pmc.sh r3_pmc_ev --events-only > "$OUT/pmc_ev.log" 2>&1
bash tools/pmc.sh r3_pmc_img --no-events-only-leg > "$OUT/pmc_img.log" 2>&1
cd "$ROOT" && python tools/make_traffic_json.py r3_pmc_img r3_pmc_ev r3 && cp profiles/r3_traffic.json "$OUT/"
bash tools/prof.sh r3_img_e1 --engines 1 --no-events-only-leg > "$OUT/prof_img_e1.log" 2>&1
bash tools/prof.sh r3_ev_e1 --events-only --engines 1 > "$OUT/prof_ev_e1.log" 2>&1
bash tools/prof.sh r3_default > "$OUT/prof_default.log" 2>&1
bash tools/prof_any.sh r3_edges tools/stage_probe.py edges:8:100000 > "$OUT/prof_edges.log" 2>&1
cd "$ROOT"
rm -f gpurun_out/parity_stage_errors.jsonl
timeout 1500 python -m pytest tests/ -q -m gpu > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1
cp gpurun_out/parity_stage_errors.jsonl "$OUT/" 2>/dev/null
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 600 "$OUT/bench_default.json"; tail -2 "$OUT/bench_default.err"
timeout 200 python tools/train_probe.py 8 50000 10 > "$OUT/train_probe.json" 2>&1; tail -1 "$OUT/train_probe.json"
timeout 200 python tools/tail_probe.py > "$OUT/tail_probe.jsonl" 2>&1
