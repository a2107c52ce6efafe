#!/bin/bash
# Entry-balanced row blocks of the entry-major layer kernels on the reference's chess data: one captured training step under
# rocprofv3 --kernel-trace --stats with the partition (default) and without it (TMGCN_L12_ROW_BLOCKS=0).
# usage (GPU box): bash tools/l12_partition_ab.sh TAG
TAG=${1:-l12ab}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in 1 0; do
  export TMGCN_L12_ROW_BLOCKS=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$v" -- python3 tools/chess_epoch.py --epochs 200 --only graph_fused > "$OUT/chess_row_blocks_$v.json" 2> "$OUT/chess_row_blocks_$v.err"
  f=$(find "$OUT/prof_$v" -name '*kernel_stats.csv' | head -1)
  echo "== TMGCN_L12_ROW_BLOCKS=$v"; tail -1 "$OUT/chess_row_blocks_$v.json" | cut -c1-400
  [ -n "$f" ] && { cp "$f" "$OUT/chess_row_blocks_${v}_kernel_stats.csv"; grep -E "l12_|head_loss|gemm_dw_narrow|sgd" "$f" | cut -d, -f1-4 | cut -c1-160; }
  rm -rf "$OUT/prof_$v"
done
for c in S1; do
  for v in 1 0; do
    export TMGCN_L12_ROW_BLOCKS=$v
    python3 tools/epoch_bench.py $c --epoch-reps 50 --cpu-epoch-reps 0 --modes graph_fused 2>/dev/null | tail -1 | cut -c1-300
  done
done
