#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
run() {
  IMG=1 ASYNC=0 timeout 600 python tools/lat_probe.py 25000 2>/dev/null | python -c '
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
img = d["image"]
print(sys.argv[1], {s: {b: {n: v["p50"] for n, v in img[s][b].items()} for b in img[s]} for s in img})' "$1"
}
run "default"
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0 run "no gtc nhwc"
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 run "no implicit gemm"
MIOPEN_FIND_MODE=1 run "find mode normal"
