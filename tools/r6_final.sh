#!/bin/bash
# end-of-round evidence in ONE box (so that the bench line can carry the PMC traffic of the build it runs on): PMC passes ->
# profiles/r6_traffic.json (stamped), kernel stats of the bench commands + the S-edges probe + a B = 1 window timeline,
# full GPU test-suite, default bench line.  Everything lands in gpurun_out/ (merged back); tools/r6_collect.sh copies it
# into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_final
mkdir -p "$OUT"
cd "$ROOT"
bash tools/pmc.sh r6_pmc_ev --events-only > "$OUT/pmc_ev.log" 2>&1
bash tools/pmc.sh r6_pmc_img --no-events-only-leg > "$OUT/pmc_img.log" 2>&1
cd "$ROOT" && python tools/make_traffic_json.py r6_pmc_img r6_pmc_ev r6 && cp profiles/r6_traffic.json "$OUT/"
bash tools/prof.sh r6_img_e1 --engines 1 --no-events-only-leg > "$OUT/prof_img_e1.log" 2>&1
bash tools/prof.sh r6_ev_e1 --events-only --engines 1 > "$OUT/prof_ev_e1.log" 2>&1
bash tools/prof.sh r6_default > "$OUT/prof_default.log" 2>&1
bash tools/prof_any.sh r6_edges tools/stage_probe.py edges:8:100000 > "$OUT/prof_edges.log" 2>&1
bash tools/prof_trace.sh r6_tl_b1 tools/latency_trace.py 1 25000 6 > "$OUT/trace_b1.log" 2>&1
python tools/latency_timeline.py gpurun_out/r6_tl_b1/kernel_trace.csv > "$OUT/timeline_b1_25k.txt" 2>&1; rm -f gpurun_out/r6_tl_b1/kernel_trace.csv
bash tools/prof_trace.sh r6_tl_b8 tools/latency_trace.py 8 100000 6 > "$OUT/trace_b8.log" 2>&1
python tools/latency_timeline.py gpurun_out/r6_tl_b8/kernel_trace.csv > "$OUT/timeline_b8_100k.txt" 2>&1; rm -f gpurun_out/r6_tl_b8/kernel_trace.csv
cd "$ROOT"
rm -f gpurun_out/parity_stage_errors.jsonl
export DAGR_PARITY_LOG=$OUT/parity_stage_errors.jsonl
( time timeout 1500 python -m pytest tests/ -q -m gpu ) > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1
unset DAGR_PARITY_LOG
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 400 "$OUT/bench_default.json"; tail -2 "$OUT/bench_default.err"
timeout 200 python tools/pool_probe.py uniform:8:100000:0 uniform:8:100000:1 edges:8:100000:0 edges:8:100000:1 uniform:1:25000:0 2>&1 | grep spec > "$OUT/pool_probe.jsonl"
timeout 200 python tools/img_branch_probe.py 2>&1 | grep lt_residual > "$OUT/img_branch_probe.jsonl"
timeout 200 python tools/train_probe.py 8 50000 10 > "$OUT/train_probe.json" 2>&1; tail -1 "$OUT/train_probe.json" | cut -c1-300
timeout 200 python tools/train_syncs.py 2>/dev/null | head -24 > "$OUT/train_syncs.txt"; head -3 "$OUT/train_syncs.txt"
timeout 200 python tools/tail_probe.py uniform:8:100000 uniform:1:25000 2>&1 | grep spec > "$OUT/tail_probe.jsonl"
ASYNC=1 timeout 300 python tools/lat_probe.py 25000 > "$OUT/lat_probe.json" 2>/dev/null; tail -c 700 "$OUT/lat_probe.json"

IMG=1 ASYNC=0 timeout 400 python tools/lat_probe.py 25000 > "$OUT/lat_probe_img.json" 2>/dev/null; tail -c 500 "$OUT/lat_probe_img.json"
# the graph build alone: digests of the event-ordered edge_index + codes must equal rounds 4 / 5 (profiles/r5_graph_probe_r5.jsonl)
SPECS="uniform:8:100000 edges:8:100000 edges:8:200000 edges:1:400000 uniform:1:25000 edges:1:25000 uniform:8:400000 uniform:1:200000"
PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/graph_probe_r6.jsonl" 2>/dev/null
cat "$OUT/graph_probe_r6.jsonl" | cut -c1-140
timeout 300 python tools/config_probe.py > "$OUT/config_probe.jsonl" 2>/dev/null; cat "$OUT/config_probe.jsonl" | cut -c1-200
