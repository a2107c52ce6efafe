#!/bin/bash
# Round 6: the bf16-product kernel's Y tile stored as whole rows through LDS (in-tree) against 64-byte pieces from the accumulators
# (variant `direct` = -DTMGCN_BX_STAGED_Y=0), after the kernel tests.   usage: VARS="direct …" tools/r6_bx3_store_ab.sh TAG
tag=${1:-r6_70}; mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_operand_wide.py tests/test_gpu_fuzz.py tests/test_gpu_pools.py -x -q -m gpu > gpurun_out/$tag/test_kernels.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/$tag/test_kernels.log
for g in "chess_tiled 32" "er 3" "er 7" "er 11"; do set -- $g; AB_GRAPH=$1 AB_DEG=$2 AB_T=16 timeout 300 python tools/ab_fused.py ${VARS:-direct} > gpurun_out/$tag/ab_store_$1_$2.txt 2>&1; echo "== $g rc=$?"; grep -E "median|check" gpurun_out/$tag/ab_store_$1_$2.txt | grep -v "^spmm "; done
