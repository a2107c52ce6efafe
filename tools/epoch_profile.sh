#!/bin/bash
# Kernel-trace statistics of the training epochs of one reference-shaped config in chosen modes.
#   tools/epoch_profile.sh TAG CONFIG "MODES"      e.g.  tools/epoch_profile.sh r4b S2 "fused"
# Writes gpurun_out/TAG/<CONFIG>_<modes>_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the epoch
# times of the same process.  Setup kernels (adjacency ingest, sorts) appear with a handful of calls; the
# per-epoch kernels with 6 passes x --epoch-reps (+ warm-up) calls.
set -u
TAG=$1; CFG=$2; MODES=$3
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
NAME=${CFG}_$(echo $MODES | tr ' ' '+')
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$NAME" -- python3 tools/epoch_bench.py $CFG --epoch-reps 20 --cpu-epoch-reps 0 --modes $MODES > "$OUT/${NAME}_under_rocprof.json" 2> "$OUT/${NAME}_under_rocprof.err"
f=$(find "$OUT/prof_$NAME" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${NAME}_kernel_stats.csv"
t=$(find "$OUT/prof_$NAME" -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python3 tools/epoch_sequence.py "$t" > "$OUT/${NAME}_epoch_sequence.txt" 2>&1
rm -rf "$OUT/prof_$NAME"
head -14 "$OUT/${NAME}_kernel_stats.csv" | cut -c1-200
cat "$OUT/${NAME}_epoch_sequence.txt"
