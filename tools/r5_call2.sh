#!/bin/bash
# round 5, GPU call 2: the time-bucketed index.  (1) bit-exactness: the graph / property / asynchronous-update suites + digests
# of the event-ordered edge_index against the round-4 library at full size, (2) build time A/B, (3) knob sweeps, (4) kernel stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c2
mkdir -p "$OUT"
cd "$ROOT"
R4=$ROOT/dagr_amd/lib/libdagr_hip_r4.so
SPECS="uniform:8:100000 edges:8:100000 uniform:1:25000 edges:1:25000 uniform:8:400000 edges:8:200000 uniform:1:400000 edges:1:400000 uniform:2:3000"
( time timeout 900 python -m pytest -q -m gpu tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_async_update_gpu.py \
    tests/test_queue_compat_gpu.py ) > "$OUT/pytest_graph.log" 2>&1
tail -25 "$OUT/pytest_graph.log"
echo "== digests + build time: round-4 library"
DAGR_HIP_LIB=$R4 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_r4.jsonl" 2> "$OUT/probe_r4.err"; cat "$OUT/probe_r4.jsonl"
echo "== digests + build time: new"
PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_new.jsonl" 2> "$OUT/probe_new.err"; cat "$OUT/probe_new.jsonl"; tail -3 "$OUT/probe_new.err"
echo "== rows variants"
for v in 45 46 26 36 47; do
  echo "variant $v: $(DAGR_ROWS_VARIANT=$v timeout 300 python tools/graph_probe.py uniform:8:100000 edges:8:100000 uniform:1:25000 2>/dev/null | tr '\n' ' ')"
done 2>&1 | tee "$OUT/variants.txt"
echo "== time buckets"
for cfg in "1 0" "3 16700" "4 12500" "5 0" "6 0" "8 0"; do
  set -- $cfg
  echo "nb=$1 wb=$2: $(DAGR_TIME_BUCKETS=$1 DAGR_BUCKET_US=$2 PROBE_CHECK=1 timeout 300 python tools/graph_probe.py uniform:8:100000 edges:8:100000 uniform:1:25000 2>/dev/null | tr '\n' ' ')"
done 2>&1 | tee "$OUT/buckets.txt"
echo "== kernel stats"
for spec in uniform:8:100000 edges:8:100000; do
  tag=${spec//:/_}
  DAGR_HIP_LIB=$R4 bash tools/prof_any.sh r5c2_prof_r4_$tag tools/graph_probe.py $spec > /dev/null 2>&1
  bash tools/prof_any.sh r5c2_prof_new_$tag tools/graph_probe.py $spec > /dev/null 2>&1
  for t in r5c2_prof_r4_$tag r5c2_prof_new_$tag; do echo "-- $t"; python - "$ROOT/gpurun_out/$t/kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    name = re.sub(r"\(anonymous namespace\)::|void |dagr::", "", r["Name"]).split("(")[0]
    print(f'{name[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f}')
PY
  done
done 2>&1 | tee "$OUT/kernel_stats.txt"
