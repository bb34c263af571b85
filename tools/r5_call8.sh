#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c8
mkdir -p "$OUT"
cd "$ROOT"
R4=$ROOT/dagr_amd/lib/libdagr_hip_r4.so
( timeout 900 python -m pytest -q -m gpu tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_async_update_gpu.py ) > "$OUT/pytest_graph.log" 2>&1
tail -3 "$OUT/pytest_graph.log"
SPECS="uniform:8:100000 edges:8:100000 edges:8:200000 edges:1:400000 uniform:1:25000 edges:1:25000 edges:1:100000 uniform:8:400000"
DAGR_HIP_LIB=$R4 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_r4.jsonl" 2>/dev/null
for cfg in "5 0 128" "5 0 192" "5 0 96" "4 12500 128"; do
  set -- $cfg
  DAGR_TIME_BUCKETS=$1 DAGR_BUCKET_US=$2 DAGR_DEFER_CAP=$3 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_$1_$3.jsonl" 2>/dev/null
  python - "$OUT/probe_r4.jsonl" "$OUT/probe_$1_$3.jsonl" "nb=$1 cap=$3" <<'PY'
import json, sys
a = [json.loads(l) for l in open(sys.argv[1])]
b = [json.loads(l) for l in open(sys.argv[2])]
print(sys.argv[3], " ".join(f'{x["spec"]}: {x["build_us"]:.0f}->{y["build_us"]:.0f}{"" if x["digest"] == y["digest"] else " DIFF!"}' for x, y in zip(a, b)))
PY
done
echo "== kernel stats S-edges"
for cfg in "5 0 128"; do
  set -- $cfg
  DAGR_TIME_BUCKETS=$1 DAGR_BUCKET_US=$2 DAGR_DEFER_CAP=$3 bash tools/prof_any.sh r5c8_prof_$1_$3 tools/graph_probe.py edges:8:100000 > /dev/null 2>&1
  echo "-- nb=$1 cap=$3"; python - "$ROOT/gpurun_out/r5c8_prof_$1_$3/kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    name = re.sub(r"\(anonymous namespace\)::|void |dagr::", "", r["Name"]).split("(")[0]
    print(f'{name[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f}')
PY
done
DAGR_DEFER_CAP=128 timeout 600 python bench.py --events-only --no-cpu-baseline --no-latency --steps 20 --warmup 5 > "$OUT/bench_ev_new.json" 2>"$OUT/bench_ev_new.err"
wc -l "$OUT/bench_ev_new.json"; python - "$OUT/bench_ev_new.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("ms_per_step", d["ms_per_step"], "value", d["value"], d["gather"])
PY
