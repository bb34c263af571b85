#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
SPECS="edges:8:100000 edges:8:200000 uniform:8:400000 uniform:1:200000 uniform:8:100000 edges:1:25000"
run() { DAGR_ROWS_VARIANT=$3 DAGR_RING_THR=$1 DAGR_RING_WANT=$2 PROBE_CHECK=1 timeout 300 python tools/graph_probe.py $SPECS 2>/dev/null | python -c "
import json,sys
print('thr=$1 want=$2 variant=$3', ' | '.join(f\"{d['build_us']:.0f} ring {d.get('ring_limited')} {d['digest'][:6]}\" for d in map(json.loads, sys.stdin)))"; }
run 0 5 47
run 0 5 46
for want in 4 5 6 8; do run 200 $want 47; run 200 $want 46; done
run 128 6 47
run 128 8 47
