#!/bin/bash
# round 5: counter list of the box, MFMA-pipe / VALU-pipe busy counters of the level-0 convs, the new GPU tests, a short bench line
# of a tile hidden under phase 1 of other waves' tiles?), (3) the new tests, (4) a short default bench line (one-rank RCCL group).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export DAGR_HIP_LIB="$ROOT/dagr_amd/lib/libdagr_hip_measure.so"   # the knobs below exist in the measurement build only
OUT=$ROOT/gpurun_out/r5c1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_all.txt" 2>&1
grep -o -i "SQ_[A-Z0-9_]*\(MFMA\|VALU\|BUSY\)[A-Z0-9_]*" "$OUT/counters_all.txt" | sort -u > "$OUT/counters_sq.txt"
wc -l "$OUT/counters_sq.txt"; tr '\n' ' ' < "$OUT/counters_sq.txt"; echo
pass() {   # pass <name> <counters...>
  local name=$1; shift
  local have=""
  for c in "$@"; do grep -qx "$c" "$OUT/counters_sq.txt" && have="$have $c"; done
  echo "pass $name:$have"
  [ -z "$have" ] && return
  timeout 400 rocprofv3 --kernel-trace --pmc $have --output-format csv -d "$OUT/$name" -- \
      python "$ROOT/bench.py" --no-cpu-baseline --no-latency --no-rccl --steps 3 --warmup 1 --engines 1 $( [ "${EVONLY:-0}" = 1 ] && echo --events-only || echo --no-events-only-leg ) > "$OUT/$name.log" 2>&1 < /dev/null
  echo "rc=$?"
  local f=$(find "$OUT/$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python "$ROOT/tools/pmc_agg.py" "$f" > "$OUT/pmc_$name.csv"; grep -E "^kernel|conv_l0|search_rows|conv_fused" "$OUT/pmc_$name.csv" | cut -c1-400; else tail -5 "$OUT/$name.log"; fi
  rm -rf "$OUT/$name"
}
pass_ev() { EVONLY=1 pass "$@"; }
pass pipes SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass pipes2 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA
pass_ev pipes_ev SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES
cd "$ROOT"
( time timeout 1500 python -m pytest -q -m gpu -x tests/test_rccl_world1_gpu.py tests/test_run_test_script_gpu.py \
    "tests/test_training_gpu.py::test_two_forwards_before_one_backward_do_not_share_the_loss_graph" \
    "tests/test_training_gpu.py::test_train_script_under_distributed_data_parallel_on_one_gpu" \
    "tests/test_engine_gpu.py::test_vga_b8_dense_windows" "tests/test_engine_gpu.py::test_bench_workload_dagr_s_resnet50_vga_b8_100k_edges" ) > "$OUT/pytest_new.log" 2>&1
tail -15 "$OUT/pytest_new.log"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency > "$OUT/bench_short.json" 2> "$OUT/bench_short.err"
tail -c 1500 "$OUT/bench_short.json"; tail -3 "$OUT/bench_short.err"
