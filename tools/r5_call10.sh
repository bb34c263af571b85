#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for nb in 1 5; do
for a in 192 448; do
  for spec in uniform:8:100000 edges:8:100000; do
  rm -rf gpurun_out/r5c10_t
  DAGR_DEFER_CAP=128 DAGR_TIME_BUCKETS=$nb DAGR_ABLATE=$a bash tools/prof_any.sh r5c10_t tools/graph_probe.py $spec > /dev/null 2>&1
  python - gpurun_out/r5c10_t/kernel_stats.csv $nb $a $spec <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "search_rows" in r["Name"]: print("nb", sys.argv[2], "alt", sys.argv[3], sys.argv[4], "rows avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
done; done; done
