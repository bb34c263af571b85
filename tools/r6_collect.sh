#!/bin/bash
# copy the merged end-of-round evidence (tools/r6_final.sh) from gpurun_out/ into profiles/
set -eu
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/r6_final/r6_traffic.json profiles/r6_traffic.json
python tools/stats_md.py $G/r6_img_e1/kernel_stats.csv profiles/r6_image_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --no-side-legs --engines 1 --no-events-only-leg" "single engine (end of round 6)"
python tools/stats_md.py $G/r6_ev_e1/kernel_stats.csv profiles/r6_events_only_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --no-side-legs --events-only --engines 1" "single engine: isolated per-launch times (end of round 6)"
python tools/stats_md.py $G/r6_default/kernel_stats.csv profiles/r6_default3eng_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --no-side-legs" "the driver's command (3 engines in flight), end of round 6"
python tools/stats_md.py $G/r6_edges/kernel_stats.csv profiles/r6_edges_stage_probe_kernel_stats.md "python tools/stage_probe.py edges:8:100000" "S-edges stream, B = 8 x 100 k (end of round 6)"
cp $G/r6_img_e1/kernel_stats.csv profiles/r6_image_e1_kernel_stats.csv
cp $G/r6_ev_e1/kernel_stats.csv profiles/r6_events_only_e1_kernel_stats.csv
cp $G/r6_default/kernel_stats.csv profiles/r6_default3eng_kernel_stats.csv
for p in fetch write sq; do cp $G/r6_pmc_ev/pmc_$p.csv profiles/r6_events_only_pmc_$p.csv; cp $G/r6_pmc_img/pmc_$p.csv profiles/r6_image_pmc_$p.csv; done
tail -1 $G/r6_final/bench_default.json > profiles/r6_bench_default.json
cp $G/r6_final/pytest_gpu.log profiles/r6_pytest_gpu.log
cp $G/r6_final/parity_stage_errors.jsonl profiles/r6_parity_stage_errors.jsonl
cp $G/r6_final/timeline_b1_25k.txt profiles/r6_timeline_b1_25k.txt
cp $G/r6_final/timeline_b8_100k.txt profiles/r6_timeline_b8_100k.txt
cp $G/r6_final/pool_probe.jsonl profiles/r6_pool_probe.jsonl
cp $G/r6_final/img_branch_probe.jsonl profiles/r6_img_branch_probe.jsonl
tail -1 $G/r6_final/train_probe.json > profiles/r6_train_probe.json
cp $G/r6_final/train_syncs.txt profiles/r6_train_syncs.txt
cp $G/r6_final/tail_probe.jsonl profiles/r6_tail_probe.jsonl
cp $G/r6_final/lat_probe.json profiles/r6_lat_probe.json

cp $G/r6_final/lat_probe_img.json profiles/r6_lat_probe_img.json
cp $G/r6_final/graph_probe_r6.jsonl profiles/r6_graph_probe_r6.jsonl; cp $G/r6_final/config_probe.jsonl profiles/r6_config_probe.jsonl
