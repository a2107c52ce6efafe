#!/bin/bash
# Round 6: the low-degree ring kernel (variants built with -DTMGCN_RING_MAX_DEG=14) against the in-tree library (bf16-product tile
# kernel) on the chess operand at bench size and at 4 / 8 / 12 random entries per row.   usage: VARS="ring …" tools/r6_ring_ab.sh TAG
tag=${1:-r6_60}; mkdir -p gpurun_out/$tag
# a small launch first, under a short timeout: a stalled hand-over must not hold the box
AB_GRAPH=er AB_DEG=3 AB_T=2 AB_N=20000 timeout 120 python tools/ab_fused.py $VARS > gpurun_out/$tag/ab_small.txt 2>&1; echo "small rc=$?"; grep -E "median|check" gpurun_out/$tag/ab_small.txt | grep -v "^spmm "
grep -q "check" gpurun_out/$tag/ab_small.txt || { tail -5 gpurun_out/$tag/ab_small.txt; exit 1; }
for g in "chess_tiled 32" "er 3"; do set -- $g; AB_GRAPH=$1 AB_DEG=$2 AB_T=16 timeout 300 python tools/ab_fused.py $VARS > gpurun_out/$tag/ab_ring_$1_$2.txt 2>&1; echo "== $g rc=$?"; grep -E "median|check" gpurun_out/$tag/ab_ring_$1_$2.txt | grep -v "^spmm "; done
