#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-calib}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python "$ROOT/tools/calib.py" > "$OUT/$c.log" 2>&1 < /dev/null
  f=$(find "$OUT/$c" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$ROOT/tools/pmc_agg.py" "$f" | grep -i calib | tee "$OUT/calib_$c.csv"
  rm -rf "$OUT/$c"
done
