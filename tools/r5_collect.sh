#!/bin/bash
# copy the merged end-of-round evidence (tools/r5_final.sh) from gpurun_out/ into profiles/
set -eu
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/r5_final/r5_traffic.json profiles/r5_traffic.json
python tools/stats_md.py $G/r5_img_e1/kernel_stats.csv profiles/r5_image_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --engines 1 --no-events-only-leg" "single engine (end of round 5)"
python tools/stats_md.py $G/r5_ev_e1/kernel_stats.csv profiles/r5_events_only_e1_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency --events-only --engines 1" "single engine: isolated per-launch times (end of round 5)"
python tools/stats_md.py $G/r5_default/kernel_stats.csv profiles/r5_default3eng_kernel_stats.md "python bench.py --no-cpu-baseline --no-latency" "the driver's command (3 engines in flight), end of round 5"
python tools/stats_md.py $G/r5_edges/kernel_stats.csv profiles/r5_edges_stage_probe_kernel_stats.md "python tools/stage_probe.py edges:8:100000" "S-edges stream, B = 8 x 100 k (end of round 5)"
cp $G/r5_img_e1/kernel_stats.csv profiles/r5_image_e1_kernel_stats.csv
cp $G/r5_ev_e1/kernel_stats.csv profiles/r5_events_only_e1_kernel_stats.csv
cp $G/r5_default/kernel_stats.csv profiles/r5_default3eng_kernel_stats.csv
for p in fetch write sq; do cp $G/r5_pmc_ev/pmc_$p.csv profiles/r5_events_only_pmc_$p.csv; cp $G/r5_pmc_img/pmc_$p.csv profiles/r5_image_pmc_$p.csv; done
tail -1 $G/r5_final/bench_default.json > profiles/r5_bench_default.json
cp $G/r5_final/pytest_gpu.log profiles/r5_pytest_gpu.log
cp $G/r5_final/parity_stage_errors.jsonl profiles/r5_parity_stage_errors.jsonl
cp $G/r5_final/timeline_b1_25k.txt profiles/r5_timeline_b1_25k.txt
cp $G/r5_final/timeline_b8_100k.txt profiles/r5_timeline_b8_100k.txt
cp $G/r5_final/pool_probe.jsonl profiles/r5_pool_probe.jsonl
cp $G/r5_final/img_branch_probe.jsonl profiles/r5_img_branch_probe.jsonl
tail -1 $G/r5_final/train_probe.json > profiles/r5_train_probe.json
cp $G/r5_final/train_syncs.txt profiles/r5_train_syncs.txt
cp $G/r5_final/tail_probe.jsonl profiles/r5_tail_probe.jsonl
cp $G/r5_final/lat_probe.json profiles/r5_lat_probe.json

cp $G/r5_final/lat_probe_img.json profiles/r5_lat_probe_img.json
for f in graph_probe_r4 graph_probe_r5 graph_probe_r5_5buckets; do [ -f $G/r5_final/$f.jsonl ] && cp $G/r5_final/$f.jsonl profiles/r5_$f.jsonl; done
