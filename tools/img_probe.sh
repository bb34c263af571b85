#!/bin/bash
# tools/img_probe.sh -- the image branch under a set of MIOpen solver switches, one fresh process (and find-db) each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/img_probe
mkdir -p "$OUT"
run() { tag=$1; shift; env MIOPEN_USER_DB_PATH=/tmp/miopen_$tag "$@" timeout 200 python "$ROOT/tools/img_probe.py" "$tag" ${BENCH:-1} 2>/dev/null | tail -1 >> "$OUT/results.jsonl"; }
: > "$OUT/results.jsonl"
run default
run no_asm_nhwc MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0
run find_normal MIOPEN_FIND_MODE=1
run find_normal_no_asm MIOPEN_FIND_MODE=1 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0
run no_ck_group MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_FWD_XDLOPS=0
BENCH=0 run immediate
cat "$OUT/results.jsonl"
