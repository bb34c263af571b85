#!/bin/bash
# graph builder regression + timing: GPU graph tests, full-size oracle comparison, build times, kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3_graph_check}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_properties_gpu.py tests/test_queue_compat_gpu.py -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
timeout 600 python tools/graph_vs_oracle.py edges:8:100000 uniform:2:100000 2>&1 | grep -c ": ok"
PROBE_CHECK=1 python tools/graph_probe.py uniform:8:100000 edges:8:100000 uniform:1:25000 uniform:8:400000 edges:8:25000 edges:8:400000 2>&1 | grep spec | tee "$OUT/probe.jsonl"
bash tools/prof_any.sh $TAG/prof tools/graph_probe.py uniform:8:100000 edges:8:100000 > /dev/null
python - "$OUT/prof/kernel_stats.csv" <<'PY'
import csv,sys,re
for r in list(csv.reader(open(sys.argv[1])))[1:9]:
    print(re.sub(r"\(.*","",r[0].replace("(anonymous namespace)::","").replace("void ",""))[:40],"calls",r[1],"avg us",round(float(r[3])/1e3,1),"min",round(float(r[5])/1e3,1),"max",round(float(r[6])/1e3,1))
PY
