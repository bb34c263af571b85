#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c9
mkdir -p "$OUT"
cd "$ROOT"
R4=$ROOT/dagr_amd/lib/libdagr_hip_r4.so
( timeout 900 python -m pytest -q -m gpu tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_async_update_gpu.py ) > "$OUT/pytest_graph.log" 2>&1
tail -3 "$OUT/pytest_graph.log"
SPECS="uniform:8:100000 edges:8:100000 edges:8:200000 edges:1:400000 uniform:1:25000 edges:1:25000 uniform:8:400000 uniform:1:200000 uniform:2:3000"
DAGR_HIP_LIB=$R4 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_r4.jsonl" 2>/dev/null
cmp() { python - "$OUT/probe_r4.jsonl" "$1" "$2" <<'PY'
import json, sys
a = [json.loads(l) for l in open(sys.argv[1])]
b = [json.loads(l) for l in open(sys.argv[2])]
print(sys.argv[3], " ".join(f'{x["spec"]}: {x["build_us"]:.0f}->{y["build_us"]:.0f}{"" if x["digest"] == y["digest"] else " DIFF!"}' for x, y in zip(a, b)))
PY
}
PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_default.jsonl" 2>/dev/null; cmp "$OUT/probe_default.jsonl" "policy (capacity = the window)"
DAGR_TIME_BUCKETS=1 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_nb1.jsonl" 2>/dev/null; cmp "$OUT/probe_nb1.jsonl" "one bucket everywhere"
DAGR_TIME_BUCKETS=5 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_nb5.jsonl" 2>/dev/null; cmp "$OUT/probe_nb5.jsonl" "five buckets everywhere"
DAGR_TIME_BUCKETS=1 DAGR_ROWS_VARIANT=46 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_nb1_46.jsonl" 2>/dev/null; cmp "$OUT/probe_nb1_46.jsonl" "one bucket, variant 46"
DAGR_TIME_BUCKETS=5 DAGR_ROWS_VARIANT=35 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_nb5_35.jsonl" 2>/dev/null; cmp "$OUT/probe_nb5_35.jsonl" "five buckets, variant 35"
timeout 900 python -m pytest -q -m gpu tests/test_engine_gpu.py -k "small or tiny or vga_uniform_b1 or max_neighbors or b1_dense" 2>&1 | tail -3
timeout 600 python bench.py --events-only --no-cpu-baseline --no-latency --steps 20 --warmup 5 > "$OUT/bench_ev_new.json" 2>"$OUT/bench_ev_new.err"
python - "$OUT/bench_ev_new.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("events-only ms_per_step", d["ms_per_step"], "value", d["value"], {k: v["ms"] for k, v in d["stages"].items()})
PY
