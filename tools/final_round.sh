#!/bin/bash
# tools/final_round.sh -- the end-of-round measurement set (run through gpurun): GPU tests, kernel-trace stats of
# both bench configurations (default command and --engines 1), PMC passes, and the two bench lines with cpu_baseline.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests -q -m gpu < /dev/null > "$OUT/pytest_gpu.log" 2>&1; tail -2 "$OUT/pytest_gpu.log"
bash tools/prof.sh final_img_e1 --engines 1 > "$OUT/prof_img_e1.log" 2>&1
bash tools/prof.sh final_ev_e1 --events-only --engines 1 > "$OUT/prof_ev_e1.log" 2>&1
bash tools/prof.sh final_img > "$OUT/prof_img.log" 2>&1
bash tools/prof.sh final_ev --events-only > "$OUT/prof_ev.log" 2>&1
bash tools/pmc.sh final_pmc_img > "$OUT/pmc_img.log" 2>&1
bash tools/pmc.sh final_pmc_ev --events-only > "$OUT/pmc_ev.log" 2>&1
cd "$ROOT"
timeout 400 python bench.py < /dev/null > "$OUT/bench_default.log" 2>&1; tail -1 "$OUT/bench_default.log" | cut -c1-200
timeout 400 python bench.py --events-only < /dev/null > "$OUT/bench_events_only.log" 2>&1; tail -1 "$OUT/bench_events_only.log" | cut -c1-200
