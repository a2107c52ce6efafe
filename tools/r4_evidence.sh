#!/bin/bash
# The GPU-box session behind the round-4 evidence files (profiles/archive/r4z_*, final tree).  Usage: tools/r4_evidence.sh TAG
TAG=${1:-r4q}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# 1. the default bench line, as the driver runs it (all legs: verify, live traffic, hbm-only, epochs, CPU baseline)
( time python3 bench.py --steps 20 --warmup 5 ) > "$OUT/bench_20_steps_5_warmup.json" 2> "$OUT/bench_20_steps_5_warmup.err"
# 2. the same command's GPU legs under rocprofv3 --kernel-trace --stats, in ONE process (line + profile of the same run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-epochs > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
f=$(find "$OUT/prof_bench" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_kernel_stats.csv"
rm -rf "$OUT/prof_bench"
# 3. matrix-core utilisation of the dominant kernels (PMC pass of its own: kernel trace only)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-epochs --no-verify --no-measure-traffic --no-hbm-only > /dev/null 2> "$OUT/pmc_mfma.err"
python3 tools/mfma_util.py "$OUT/pmc_mfma" > "$OUT/mfma_utilisation_bench.json" 2>> "$OUT/pmc_mfma.err"
rm -rf "$OUT/pmc_mfma"
# 4. epochs: kernel stats + one-epoch sequences of the captured steps
for c in S1 S2 S3; do tools/epoch_profile.sh "$TAG" $c graph_fused > /dev/null 2>&1; done
tools/epoch_profile.sh "$TAG" S2 "fused script" > /dev/null 2>&1
ls "$OUT"
tail -3 "$OUT/bench_20_steps_5_warmup.err"
