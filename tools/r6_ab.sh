#!/bin/bash
# Round 6: the kernel tests, then tools/ab_fused.py (in-tree library against build/variants/<names>) on the chess operand at bench
# size, 4 / 8 random entries per row and the S4 graph.   usage: tools/r6_ab.sh TAG variant…   (variants: tools/ab_variants.sh, or
# tools/ab_r4_baseline.sh REV NAME for the kernels of a commit)
tag=$1; shift
mkdir -p gpurun_out/$tag
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_real_operand_wide.py tests/test_gpu_fuzz.py tests/test_gpu_pools.py -x -q -m gpu > gpurun_out/$tag/test_kernels.log 2>&1; tail -2 gpurun_out/$tag/test_kernels.log
for g in "chess_tiled 32" "er 3" "er 7" "er 32" "powerlaw 32"; do set -- $g; AB_GRAPH=$1 AB_DEG=$2 AB_T=16 python tools/ab_fused.py "${@:3}" $VARS > gpurun_out/$tag/ab_$1_$2.txt 2>&1; echo == $g; grep -E "median|check" gpurun_out/$tag/ab_$1_$2.txt | grep -v "^spmm "; done
