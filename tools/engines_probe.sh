#!/bin/bash
# tools/engines_probe.sh -- throughput of the bench workload vs the number of engines (streams) in flight
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/engines_probe; mkdir -p "$OUT"; : > "$OUT/results.txt"
for e in 3 4 6; do
  timeout 200 python "$ROOT/bench.py" --no-cpu-baseline --no-latency --engines $e 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('engines $e image value %.1f M ms %.3f | events-only %.1f M ms %.3f' % (d['value']/1e6, d['ms_per_step'], d['events_only']['value']/1e6, d['events_only']['ms_per_step']))" >> "$OUT/results.txt"
done
cat "$OUT/results.txt"
