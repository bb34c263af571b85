#!/bin/bash
# after a (comment-only) change of the kernel sources: the PMC passes again on this build, so that the traffic file's source
# stamp matches the library bench.py runs on, and the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_final
mkdir -p "$OUT"
cd "$ROOT"
bash tools/pmc.sh r5_pmc_ev --events-only > "$OUT/pmc_ev.log" 2>&1
bash tools/pmc.sh r5_pmc_img --no-events-only-leg > "$OUT/pmc_img.log" 2>&1
cd "$ROOT" && python tools/make_traffic_json.py r5_pmc_img r5_pmc_ev r5 && cp profiles/r5_traffic.json "$OUT/"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 300 "$OUT/bench_default.json"
