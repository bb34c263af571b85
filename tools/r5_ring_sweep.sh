#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
SPECS="edges:8:100000 edges:8:200000 uniform:8:400000 uniform:1:200000 uniform:8:100000"
run() { DAGR_RING_THR=$1 DAGR_RING_WANT=$2 timeout 300 python tools/graph_probe.py $SPECS 2>/dev/null | python -c "
import json,sys
print('thr=$1 want=$2', ' | '.join(f\"{d['build_us']:.0f} def {d.get('deferred')} ring {d.get('ring_limited')}\" for d in map(json.loads, sys.stdin)))"; }
run 0 5
run 200 5
run 200 6
