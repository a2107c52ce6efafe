#!/bin/bash
# One GPU-box session producing the evidence files of a round: kernel-trace stats of the bench, the two
# PMC traffic passes, the MALL-vs-HBM separation run and an S2 epoch trace.  Usage: tools/gpu_profiles.sh TAG
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="--no-cpu-baseline --no-epochs --no-verify --no-measure-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -- python3 bench.py --steps 10 --warmup 3 $B > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 $B > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 2 --warmup 1 $B > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
python3 tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" --profile-id "$TAG" > "$OUT/pmc_traffic.json" 2> "$OUT/pmc_traffic.err"
# MALL vs HBM: the same 1.056 G edge-slices per launch with ONE gather window of 8 GB per slice (N = 16 M,
# 2 slices) instead of 1 GB (N = 2 M, 16 slices): the 256 MB Infinity Cache then covers 3 % of the window, not 25 %
python3 bench.py --nodes 16000000 --slices-per-gpu 2 --steps 6 --warmup 2 $B > "$OUT/bench_gather_window_8gb.json" 2> "$OUT/bench_gather_window_8gb.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_8gb" -- python3 bench.py --nodes 16000000 --slices-per-gpu 2 --steps 2 --warmup 1 $B > /dev/null 2> "$OUT/pmc_fetch_8gb.err"
python3 tools/pmc_traffic.py "$OUT/pmc_fetch_8gb" "$OUT/pmc_write" --profile-id "$TAG-8gb" --nodes 16000000 --slices-per-gpu 2 > "$OUT/pmc_traffic_8gb.json" 2>> "$OUT/pmc_traffic.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_s2_epoch" -- python3 tools/epoch_bench.py S2 --epoch-reps 20 --cpu-epoch-reps 0 > "$OUT/s2_epoch_under_rocprof.json" 2> "$OUT/s2_epoch_under_rocprof.err"
rocprofv3 -L 2>/dev/null | grep -i -E "dram|mall|TCC_EA0|HBM" | head -60 > "$OUT/counters_dram_like.txt"
# keep the merged-back payload small: stats + counter CSVs only
find "$OUT" -name "*.csv" -size +8M -delete
find "$OUT" -name "*agent_info.csv" -delete
ls -R "$OUT" | head -80 > "$OUT/listing.txt"
