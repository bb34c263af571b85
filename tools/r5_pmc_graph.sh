#!/bin/bash
# SQ counters of the graph build kernels (tools/graph_probe.py on one workload), new library and round-4 library
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
SPEC=${1:-uniform:8:100000}
TAG=${2:-r5pmc}
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
CNT2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
PROBE_REPS=5 bash tools/pmc_set.sh ${TAG}_new_a "$CNT" tools/graph_probe.py $SPEC
PROBE_REPS=5 bash tools/pmc_set.sh ${TAG}_new_b "$CNT2" tools/graph_probe.py $SPEC
DAGR_HIP_LIB=$ROOT/dagr_amd/lib/libdagr_hip_r4.so PROBE_REPS=5 bash tools/pmc_set.sh ${TAG}_r4_a "$CNT" tools/graph_probe.py $SPEC
DAGR_HIP_LIB=$ROOT/dagr_amd/lib/libdagr_hip_r4.so PROBE_REPS=5 bash tools/pmc_set.sh ${TAG}_r4_b "$CNT2" tools/graph_probe.py $SPEC
for t in new_a new_b r4_a r4_b; do echo "== $t"; grep -E "^kernel|search|k_count|scan_chained|k_order|k_scatter|k_fix" gpurun_out/${TAG}_$t/pmc.csv | cut -c1-330; done
