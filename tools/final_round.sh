#!/bin/bash
# tools/final_round.sh [tag] -- the end-of-round measurement set (run through gpurun): rocprofv3 kernel-trace stats of both
# bench configurations (single engine: isolated per-launch times; default command: what the driver runs), PMC passes
# (SQ / FETCH_SIZE / WRITE_SIZE, separate runs), the S-edges stage probe, and the default bench line.
set -u
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final_$TAG
mkdir -p "$OUT"
cd "$ROOT"
bash tools/prof.sh ${TAG}_img_e1 --engines 1 --no-events-only-leg > "$OUT/prof_img_e1.log" 2>&1
bash tools/prof.sh ${TAG}_ev_e1 --events-only --engines 1 > "$OUT/prof_ev_e1.log" 2>&1
bash tools/prof.sh ${TAG}_default > "$OUT/prof_default.log" 2>&1
bash tools/pmc.sh ${TAG}_pmc_img --no-events-only-leg > "$OUT/pmc_img.log" 2>&1
bash tools/pmc.sh ${TAG}_pmc_ev --events-only > "$OUT/pmc_ev.log" 2>&1
bash tools/prof_any.sh ${TAG}_edges tools/stage_probe.py edges:8:100000 > "$OUT/prof_edges.log" 2>&1
cd "$ROOT"
timeout 500 python bench.py < /dev/null > "$OUT/bench_default.log" 2>&1; tail -1 "$OUT/bench_default.log" | cut -c1-300
