#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
SPEC=${1:-uniform:8:100000}
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
for a in ${ABL:-0 1 9 17}; do
  rm -rf gpurun_out/r5abl_t$a
  DAGR_ABLATE=$a bash tools/prof_any.sh r5abl_t$a tools/graph_probe.py $SPEC > /dev/null 2>&1
  python - gpurun_out/r5abl_t$a/kernel_stats.csv $a <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "search_rows" in r["Name"]: print("ablate", sys.argv[2], "time avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
  DAGR_ABLATE=$a PROBE_REPS=5 bash tools/pmc_set.sh r5abl_$a "$CNT" tools/graph_probe.py $SPEC > /dev/null 2>&1
  echo "    $(grep search_rows gpurun_out/r5abl_$a/pmc.csv | cut -c1-200)"
done
head -1 gpurun_out/r5abl_0/pmc.csv
