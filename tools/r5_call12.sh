#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
for nb in 5 1; do
for a in 192 193 194 196; do
  DAGR_TIME_BUCKETS=$nb DAGR_ABLATE=$a PROBE_REPS=5 bash tools/pmc_set.sh r5c12 "$CNT" tools/graph_probe.py uniform:8:100000 > /dev/null 2>&1
  echo "nb $nb ablate $a: $(grep search_rows gpurun_out/r5c12/pmc.csv | cut -d, -f3-)"
done; done
head -1 gpurun_out/r5c12/pmc.csv
