#!/bin/bash
# Round 6: the entry-major walk of short tiles (csrc/spmm_row.h) against the row-per-wave path, interleaved in one process.
# Needs build/variants/noshort (tools/ab_variants.sh noshort "-DTMGCN_SHORT_TILE=0").  usage: tools/r6_short_tile_ab.sh TAG [variants…]
tag=${1:-r6b}; shift
vars=${@:-noshort}
mkdir -p gpurun_out/$tag
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/$tag/test_kernels.log 2>&1; tail -3 gpurun_out/$tag/test_kernels.log
for g in chess_tiled er; do AB_GRAPH=$g AB_T=16 python tools/ab_fused.py $vars > gpurun_out/$tag/ab_$g.txt 2>&1; echo == $g; grep -E "median|check" gpurun_out/$tag/ab_$g.txt; done
for d in 3 7; do AB_GRAPH=er AB_T=16 AB_DEG=$d python tools/ab_fused.py $vars > gpurun_out/$tag/ab_er_deg$d.txt 2>&1; echo == er deg $d; grep -E "median" gpurun_out/$tag/ab_er_deg$d.txt; done
python tools/real_structure_probe.py --json gpurun_out/$tag/real_structure_chess_tiled.json > gpurun_out/$tag/probe_chess.log 2>&1; grep -A12 '"ms"' gpurun_out/$tag/probe_chess.log | tr -d '\n '; echo
