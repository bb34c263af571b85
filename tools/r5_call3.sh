#!/bin/bash
# round 5, GPU call 3: rows kernel v2 (expand list, DPP scans / rank counting): exactness (digests vs the round-4 library),
# variants, deferral cap sweep, kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG:-r5c3}
mkdir -p "$OUT"
cd "$ROOT"
R4=$ROOT/dagr_amd/lib/libdagr_hip_r4.so
SPECS="uniform:8:100000 edges:8:100000 uniform:1:25000 edges:1:25000 uniform:8:400000 edges:8:200000 uniform:1:400000 edges:1:400000 uniform:2:3000"
( time timeout 900 python -m pytest -q -m gpu tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_async_update_gpu.py \
    tests/test_queue_compat_gpu.py ) > "$OUT/pytest_graph.log" 2>&1
tail -5 "$OUT/pytest_graph.log"
DAGR_HIP_LIB=$R4 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_r4.jsonl" 2> "$OUT/probe_r4.err"
PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_new.jsonl" 2> "$OUT/probe_new.err"; tail -3 "$OUT/probe_new.err"
python - "$OUT/probe_r4.jsonl" "$OUT/probe_new.jsonl" <<'PY'
import json, sys
a = [json.loads(l) for l in open(sys.argv[1])]
b = [json.loads(l) for l in open(sys.argv[2])]
for x, y in zip(a, b):
    print(f'{x["spec"]:20s} r4 {x["build_us"]:8.1f} us  new {y["build_us"]:8.1f} us  digest {"SAME" if x["digest"] == y["digest"] and x["edges"] == y["edges"] else "DIFFERENT"}')
PY
echo "== rows variants"
for v in 45 46 47 44 35 36 25 26; do
  echo "variant $v: $(DAGR_ROWS_VARIANT=$v timeout 300 python tools/graph_probe.py uniform:8:100000 edges:8:100000 uniform:1:25000 2>/dev/null | python -c 'import sys,json; print(" ".join(str(json.loads(l)["build_us"]) for l in sys.stdin))')"
done 2>&1 | tee "$OUT/variants.txt"
echo "== deferral cap"
for c in 48 64 96 128 192 256 320; do
  echo "cap $c: $(DAGR_DEFER_CAP=$c timeout 300 python tools/graph_probe.py uniform:8:100000 edges:8:100000 edges:8:200000 edges:1:100000 uniform:8:400000 2>/dev/null | python -c 'import sys,json; print(" ".join(str(json.loads(l)["build_us"]) for l in sys.stdin))')"
done 2>&1 | tee "$OUT/defer_cap.txt"
echo "== time buckets"
for cfg in "1 0" "3 16700" "4 12500" "5 0" "6 0"; do
  set -- $cfg
  echo "nb=$1 wb=$2: $(DAGR_TIME_BUCKETS=$1 DAGR_BUCKET_US=$2 timeout 300 python tools/graph_probe.py uniform:8:100000 edges:8:100000 uniform:1:25000 2>/dev/null | python -c 'import sys,json; print(" ".join(str(json.loads(l)["build_us"]) for l in sys.stdin))')"
done 2>&1 | tee "$OUT/buckets.txt"
echo "== kernel stats"
for spec in uniform:8:100000 edges:8:100000; do
  tag=${spec//:/_}
  bash tools/prof_any.sh ${TAG:-r5c3}_prof_$tag tools/graph_probe.py $spec > /dev/null 2>&1
  echo "-- $tag"; python - "$ROOT/gpurun_out/${TAG:-r5c3}_prof_$tag/kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    name = re.sub(r"\(anonymous namespace\)::|void |dagr::", "", r["Name"]).split("(")[0]
    print(f'{name[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f}')
PY
done 2>&1 | tee "$OUT/kernel_stats.txt"
