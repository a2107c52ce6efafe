#!/bin/bash
# SQ counters of the kernels of a training epoch (one PMC pass, kernel trace only).  Usage: tools/pmc_epoch.sh TAG CONFIG MODE "COUNTERS"
TAG=$1; CFG=$2; MODE=$3; CTR=$4
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT/pmc_$CFG" -- python3 tools/epoch_bench.py $CFG --epoch-reps 5 --cpu-epoch-reps 0 --modes $MODE > /dev/null 2> "$OUT/pmc_$CFG.err"
python3 tools/pmc_kernel_summary.py "$OUT/pmc_$CFG" > "$OUT/pmc_${CFG}_$(echo $CTR | tr ' ' '+' | cut -c1-60).json"
rm -rf "$OUT/pmc_$CFG"
