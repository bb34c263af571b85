#!/bin/bash
# tools/rows_variants.sh -- stage timings of the graph build under the k_search_rows variants (rounds x waves)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export DAGR_HIP_LIB="$ROOT/dagr_amd/lib/libdagr_hip_measure.so"   # the knobs below exist in the measurement build only
OUT=$ROOT/gpurun_out/rows_variants
mkdir -p "$OUT"; : > "$OUT/results.txt"
for v in 47 46 45 36; do
  echo "variant $v" >> "$OUT/results.txt"
  DAGR_ROWS_VARIANT=$v timeout 200 python "$ROOT/tools/stage_probe.py" uniform:8:100000 uniform:1:25000 edges:8:50000 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print(d['spec'], 'graph_ms', d['stages_ms']['graph'], 'pool1_ms', d['stages_ms']['pool1'])" >> "$OUT/results.txt"
done
cat "$OUT/results.txt"
