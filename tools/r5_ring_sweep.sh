#!/bin/bash
# sweeps of the row search's knobs on the graph build alone (digests must stay the same)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export DAGR_HIP_LIB="$ROOT/dagr_amd/lib/libdagr_hip_measure.so"   # the knobs below exist in the measurement build only
cd "$ROOT"
SPECS=${SPECS:-"edges:8:100000 edges:8:200000 uniform:8:400000 uniform:1:200000 uniform:8:100000 edges:1:25000"}
run() { env "$@" PROBE_CHECK=1 timeout 300 python tools/graph_probe.py $SPECS 2>/dev/null | python -c "
import json,sys
print('$*', ' | '.join(f\"{d['build_us']:.0f} ring {d.get('ring_limited')} def {d.get('deferred')} {d['digest'][:6]}\" for d in map(json.loads, sys.stdin)))"; }
run DAGR_DEFER_CAP=320
run DAGR_DEFER_CAP=256
run DAGR_DEFER_CAP=192
run DAGR_DEFER_CAP=320 DAGR_RING_WANT=7
run DAGR_DEFER_CAP=256 DAGR_RING_THR=160
run DAGR_DEFER_CAP=320
