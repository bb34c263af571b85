#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5c14
mkdir -p "$OUT"
cd "$ROOT"
R4=$ROOT/dagr_amd/lib/libdagr_hip_r4.so
for i in 1 2; do
DAGR_HIP_LIB=$R4 timeout 600 python bench.py --events-only --no-cpu-baseline --no-latency --steps 40 --warmup 10 > "$OUT/bench_ev_r4_$i.json" 2>/dev/null
timeout 600 python bench.py --events-only --no-cpu-baseline --no-latency --steps 40 --warmup 10 > "$OUT/bench_ev_new_$i.json" 2>/dev/null
done
python - "$OUT"/bench_ev_*.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read())
    print(f.split("/")[-1], "ms_per_step", d["ms_per_step"], {k: v["ms"] for k, v in d["stages"].items() if k in ("graph", "graph_search", "l0_conv1", "l0_conv2", "pool1", "tail", "head")})
PY
echo "== tail probe: fused vs two-launch on level 1"
timeout 300 python tools/tail_probe.py uniform:8:100000 2>/dev/null | grep spec | cut -c1-400
DAGR_FUSE_MAX_NODES=10000 timeout 300 python tools/tail_probe.py uniform:8:100000 2>/dev/null | grep spec | cut -c1-400
echo "== level-0 conv pipes: MFMA / VALU co-execution"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d "$OUT/pipes" -- \
    python "$ROOT/bench.py" --no-cpu-baseline --no-latency --steps 3 --warmup 1 --engines 1 --no-events-only-leg > "$OUT/pipes.log" 2>&1 < /dev/null
f=$(find "$OUT/pipes" -name "*counter_collection.csv" | head -1); python "$ROOT/tools/pmc_agg.py" "$f" > "$OUT/pmc_pipes_img.csv"; rm -rf "$OUT/pipes"
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d "$OUT/pipes" -- \
    python "$ROOT/bench.py" --no-cpu-baseline --no-latency --steps 3 --warmup 1 --engines 1 --events-only > "$OUT/pipes.log" 2>&1 < /dev/null
f=$(find "$OUT/pipes" -name "*counter_collection.csv" | head -1); python "$ROOT/tools/pmc_agg.py" "$f" > "$OUT/pmc_pipes_ev.csv"; rm -rf "$OUT/pipes"
grep -E "^kernel|conv_l0|conv_fused" "$OUT/pmc_pipes_img.csv" "$OUT/pmc_pipes_ev.csv" | cut -c1-300
cd "$ROOT"
echo "== full GPU suite"
rm -f gpurun_out/parity_stage_errors.jsonl
( time timeout 1700 python -m pytest tests/ -q -m gpu ) > "$OUT/pytest_gpu.log" 2>&1
tail -8 "$OUT/pytest_gpu.log"
