#!/bin/bash
# The GPU-box session behind the round-5 evidence files (profiles/archive/r5z_*, final tree).  Usage: tools/r5_evidence.sh TAG
TAG=${1:-r5z}
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
for c in S1 S2 S3 S2z2; do timeout 300 tools/epoch_profile.sh "$TAG" $c graph_fused > /dev/null 2>&1; done
timeout 300 tools/epoch_profile.sh "$TAG" S2 "fused script" > /dev/null 2>&1
# 4b. the reference's chess data: kernel stats of 1 900 steps (eager + captured)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_chess" -- python3 tools/chess_epoch.py --epochs 200 --only graph_fused > "$OUT/chess_epoch.json" 2> "$OUT/chess_epoch.err"
f=$(find "$OUT/prof_chess" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/chess_graph_fused_kernel_stats.csv"
rm -rf "$OUT/prof_chess"
ls "$OUT"
tail -3 "$OUT/bench_20_steps_5_warmup.err"
# 5. the -m gpu suite (writes gpurun_out/parity_clauses.json and tolerance_record.jsonl) and their summaries
python3 -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/status.log"
cp gpurun_out/parity_clauses.json "$OUT/parity_clauses.json" 2>/dev/null
python3 tools/tolerance_summary.py gpurun_out/tolerance_record.jsonl "$OUT/tolerance_summary.json" > /dev/null 2>&1
tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/status.log"
# 6. LAST (it replaces the in-tree library of this scratch copy by the development build with phase stamps): where the time of
#    the entry-major layer kernels goes, block by block
if [ -f build/variants/trace/libtmgcn_hip.so ]; then
  cp build/variants/trace/libtmgcn_hip.so tm-gcn_amd/libtmgcn_hip.so
  for c in chess S1; do
    timeout 200 python3 tools/l12_trace.py $c > /dev/null 2>&1 && cp gpurun_out/l12_trace_$c.json "$OUT/l12_trace_bwd_$c.json"
    timeout 200 python3 tools/l12_trace.py $c --fwd > /dev/null 2>&1 && cp gpurun_out/l12_trace_fwd_$c.json "$OUT/l12_trace_fwd_$c.json"
  done
fi
