#!/bin/bash
# copy the merged end-of-round evidence (tools/r4_final.sh) from gpurun_out/ into profiles/
set -eu
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/r4_final/r4_traffic.json profiles/r4_traffic.json
python tools/stats_md.py $G/r4_img_e1/kernel_stats.csv profiles/r4_image_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --engines 1 --no-events-only-leg" "single engine (end of round 4)"
python tools/stats_md.py $G/r4_ev_e1/kernel_stats.csv profiles/r4_events_only_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --events-only --engines 1" "single engine: isolated per-launch times (end of round 4)"
python tools/stats_md.py $G/r4_default/kernel_stats.csv profiles/r4_default3eng_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency" "the driver's command (3 engines in flight), end of round 4"
python tools/stats_md.py $G/r4_edges/kernel_stats.csv profiles/r4_edges_stage_probe_kernel_stats.md "python tools/stage_probe.py edges:8:100000" "S-edges stream, B = 8 x 100 k (end of round 4)"
cp $G/r4_img_e1/kernel_stats.csv profiles/r4_image_e1_kernel_stats.csv
cp $G/r4_ev_e1/kernel_stats.csv profiles/r4_events_only_e1_kernel_stats.csv
cp $G/r4_default/kernel_stats.csv profiles/r4_default3eng_kernel_stats.csv
for p in fetch write sq; do cp $G/r4_pmc_ev/pmc_$p.csv profiles/r4_events_only_pmc_$p.csv; cp $G/r4_pmc_img/pmc_$p.csv profiles/r4_image_pmc_$p.csv; done
tail -1 $G/r4_final/bench_default.json > profiles/r4_bench_default.json
cp $G/r4_final/pytest_gpu.log profiles/r4_pytest_gpu.log
cp $G/r4_final/parity_stage_errors.jsonl profiles/r4_parity_stage_errors.jsonl
cp $G/r4_final/timeline_b1_25k.txt profiles/r4_timeline_b1_25k.txt
cp $G/r4_final/timeline_b8_100k.txt profiles/r4_timeline_b8_100k.txt
cp $G/r4_final/pool_probe.jsonl profiles/r4_pool_probe.jsonl
cp $G/r4_final/img_branch_probe.jsonl profiles/r4_img_branch_probe.jsonl
tail -1 $G/r4_final/train_probe.json > profiles/r4_train_probe.json
cp $G/r4_final/train_syncs.txt profiles/r4_train_syncs.txt
cp $G/r4_final/tail_probe.jsonl profiles/r4_tail_probe.jsonl
cp $G/r4_final/lat_probe.json profiles/r4_lat_probe.json

