#!/bin/bash
# sweeps of the search's knobs (deferral cap, kernel variant; the heavy-list shapes this script also swept in round 6 are
# gone: profiles/r6_search_sweep.md) on the graph build alone, MEASUREMENT build of the library (digests must stay the same:
# c9aa3b e27ed6 7d2cf0 e20e2f 06d027 848a80 = round 4 / 5, profiles/r5_ring_sweep.txt)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export DAGR_HIP_LIB="$ROOT/dagr_amd/lib/libdagr_hip_measure.so"
SPECS=${SPECS:-"edges:8:100000 edges:8:200000 uniform:8:400000 uniform:1:200000 uniform:8:100000 edges:1:25000"}
run() { env "$@" PROBE_CHECK=1 timeout 300 python tools/graph_probe.py $SPECS 2>/dev/null | python -c "
import json,sys
print('$*', ' | '.join(f\"{d['build_us']:.0f} heavy {d.get('deferred')} inner {d.get('ring_limited')} {d['digest'][:6]}\" for d in map(json.loads, sys.stdin)))"; }
for cap in ${CAPS:-320 256 200}; do run DAGR_DEFER_CAP=$cap; done
run DAGR_RING_THR=0
run DAGR_ROWS_VARIANT=47
run DAGR_ROWS_VARIANT=45
run DAGR_ROWS_VARIANT=36
