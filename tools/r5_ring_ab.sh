#!/bin/bash
# A/B of the ring-limited first pass of k_search_rows (DAGR_RING_THR=0: off) + the bit-exact graph suites
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5ring
mkdir -p "$OUT"
cd "$ROOT"
SPECS=${SPECS:-"uniform:8:100000 edges:8:100000 edges:8:200000 edges:1:400000 uniform:8:400000 uniform:1:200000 edges:1:25000 uniform:2:3000"}
cmp() { python - "$1" "$2" "$3" <<'PY'
import json, sys
a = [json.loads(l) for l in open(sys.argv[1])]
b = [json.loads(l) for l in open(sys.argv[2])]
print(sys.argv[3], " ".join(f'{x["spec"]}: {x["build_us"]:.0f}->{y["build_us"]:.0f}{"" if x["digest"] == y["digest"] else " DIFF!"}' for x, y in zip(a, b)))
PY
}
DAGR_RING_THR=0 PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_off.jsonl" 2>/dev/null
for thr in ${THRS:-96 64 128}; do
DAGR_RING_THR=$thr PROBE_CHECK=1 timeout 600 python tools/graph_probe.py $SPECS > "$OUT/probe_$thr.jsonl" 2>/dev/null
cmp "$OUT/probe_off.jsonl" "$OUT/probe_$thr.jsonl" "off -> thr $thr:"
done
if [ "${TESTS:-1}" = "1" ]; then
( timeout 900 python -m pytest -q -m gpu tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_async_update_gpu.py -x ) > "$OUT/pytest_graph.log" 2>&1
tail -3 "$OUT/pytest_graph.log"
fi
