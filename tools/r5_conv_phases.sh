#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export DAGR_HIP_LIB="$ROOT/dagr_amd/lib/libdagr_hip_measure.so"   # the knobs below exist in the measurement build only
OUT=$ROOT/gpurun_out/r5c15
mkdir -p "$OUT"
cd "$ROOT"
echo "== level-0 conv phases (stage_probe: l0_conv1 / l0_conv2 ms), events-only then image"
for a in 0 1 2 3; do
  echo "ablate $a ev : $(DAGR_L0_ABLATE=$a timeout 300 python tools/stage_probe.py uniform:8:100000 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print({k: d["stages_ms"][k] for k in ("l0_conv1","l0_conv2")})')"
  echo "ablate $a img: $(PROBE_IMAGE=1 DAGR_L0_ABLATE=$a timeout 300 python tools/stage_probe.py uniform:8:100000 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print({k: d["stages_ms"][k] for k in ("l0_conv1","l0_conv2")})')"
done
echo "== image branch: bias+ReLU pass vs MIOpen fusion plan"
timeout 600 python tools/img_branch_probe.py 2>/dev/null | tail -1
DAGR_MIOPEN_FUSED=1 timeout 600 python tools/img_branch_probe.py 2>/dev/null | tail -1
echo "== S-edges kernel stats, shipped configuration"
bash tools/prof_any.sh r5c15_prof tools/graph_probe.py edges:8:100000 > /dev/null 2>&1
python - "$ROOT/gpurun_out/r5c15_prof/kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    name = re.sub(r"\(anonymous namespace\)::|void |dagr::", "", r["Name"]).split("(")[0]
    print(f'{name[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f}')
PY
