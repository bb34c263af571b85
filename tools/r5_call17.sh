#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c17
mkdir -p "$OUT"
cd "$ROOT"
( timeout 900 python -m pytest -q -m gpu tests/test_engine_gpu.py -k "window_graph or use_image_resnet18 or image_branch_inference" ) > "$OUT/pytest_img.log" 2>&1
tail -4 "$OUT/pytest_img.log"
for p in 0 1; do
  DAGR_PIPELINE_IMAGE=$p IMG=1 ASYNC=0 timeout 600 python tools/lat_probe.py 25000 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
img = d["image"]
print("pipeline", sys.argv[1], {s: {b: {n: v["p50"] for n, v in img[s][b].items()} for b in img[s]} for s in img})' $p
done
