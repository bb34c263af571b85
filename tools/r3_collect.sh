#!/bin/bash
# copy the merged end-of-round evidence (tools/r3_final.sh) from gpurun_out/ into profiles/
set -eu
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/r3_final/r3_traffic.json profiles/r3_traffic.json
python tools/stats_md.py $G/r3_img_e1/kernel_stats.csv profiles/r3_image_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --engines 1 --no-events-only-leg" "single engine (end of round 3)"
python tools/stats_md.py $G/r3_ev_e1/kernel_stats.csv profiles/r3_events_only_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --events-only --engines 1" "single engine: isolated per-launch times (end of round 3)"
python tools/stats_md.py $G/r3_default/kernel_stats.csv profiles/r3_default3eng_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency" "the driver's command (3 engines in flight), end of round 3"
python tools/stats_md.py $G/r3_edges/kernel_stats.csv profiles/r3_edges_stage_probe_kernel_stats.md "python tools/stage_probe.py edges:8:100000" "S-edges stream, B = 8 x 100 k (end of round 3)"
cp $G/r3_img_e1/kernel_stats.csv profiles/r3_image_e1_kernel_stats.csv
cp $G/r3_ev_e1/kernel_stats.csv profiles/r3_events_only_e1_kernel_stats.csv
cp $G/r3_default/kernel_stats.csv profiles/r3_default3eng_kernel_stats.csv
for p in fetch write sq; do cp $G/r3_pmc_ev/pmc_$p.csv profiles/r3_events_only_pmc_$p.csv; cp $G/r3_pmc_img/pmc_$p.csv profiles/r3_image_pmc_$p.csv; done
tail -1 $G/r3_final/bench_default.json > profiles/r3_bench_default.json
cp $G/r3_final/pytest_gpu.log profiles/r3_pytest_gpu.log
cp $G/r3_final/parity_stage_errors.jsonl profiles/r3_parity_stage_errors.jsonl
tail -1 $G/r3_final/train_probe.json > profiles/r3_train_probe.json
grep '^{' $G/r3_final/tail_probe.jsonl > profiles/r3_tail_probe.jsonl
