#!/bin/bash
# A/B of the XCD-contiguous block order of k_count / k_scatter / k_order (DAGR_XCD_REMAP=0: natural block order)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5x1
mkdir -p "$OUT"
cd "$ROOT"
SPECS="uniform:8:100000 edges:8:100000 edges:8:200000 uniform:1:25000 uniform:3:70000 uniform:2:3000 uniform:8:400000"
cmp() { python - "$1" "$2" "$3" <<'PY'
import json, sys
a = [json.loads(l) for l in open(sys.argv[1])]
b = [json.loads(l) for l in open(sys.argv[2])]
print(sys.argv[3], " ".join(f'{x["spec"]}: {x["build_us"]:.0f}->{y["build_us"]:.0f}{"" if x["digest"] == y["digest"] else " DIFF!"}' for x, y in zip(a, b)))
PY
}
for rep in 1; do
DAGR_XCD_REMAP=0 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_nat$rep.jsonl" 2>/dev/null
DAGR_XCD_REMAP=1 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_xcd$rep.jsonl" 2>/dev/null
cmp "$OUT/probe_nat$rep.jsonl" "$OUT/probe_xcd$rep.jsonl" "natural -> xcd (rep $rep)"
done
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
DAGR_XCD_REMAP=$m rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$m" -- python "$ROOT/tools/graph_probe.py" uniform:8:100000 edges:8:100000 > /dev/null 2>&1
find "$OUT/prof$m" -name "*kernel_trace.csv" -delete 2>/dev/null
f=$(find "$OUT/prof$m" -name "*kernel_stats.csv" | head -1)
echo "remap=$m"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_count", "k_scatter", "k_order", "k_fix", "scan_chained", "k_search")):
        print(f'  {n[:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:7.1f} us')
PY
done
