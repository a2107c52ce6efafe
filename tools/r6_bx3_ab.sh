#!/bin/bash
# Round 6: the bf16-product low-degree kernel (spmm_gemm_bx3_kernel) against the exact-f32 tile kernel (variant bx0 =
# -DTMGCN_BX3_MAX_DEG=0), after the whole -m gpu suite.   usage: tools/r6_bx3_ab.sh TAG
tag=${1:-r6_44}; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/$tag/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/$tag/pytest_gpu.log | tail -1
for g in "chess_tiled 32" "er 3" "er 7" "er 11" "er 15" "er 32"; do set -- $g; AB_GRAPH=$1 AB_DEG=$2 AB_T=16 timeout 200 python tools/ab_fused.py bx0 > gpurun_out/$tag/ab_bx3_$1_$2.txt 2>&1; echo "== $g rc=$?"; grep -E "median|check" gpurun_out/$tag/ab_bx3_$1_$2.txt | grep -v "^spmm "; done
