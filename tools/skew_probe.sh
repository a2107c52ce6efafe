#!/bin/bash
# Headline kernel on balanced vs skewed row lengths: one bench.py run per graph kind under rocprofv3 --kernel-trace --stats.
# usage (GPU box): bash tools/skew_probe.sh [tag] [graphs...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-skew}; shift
GRAPHS=${@:-er powerlaw powerlaw_sym}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for g in $GRAPHS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$g" -- python3 "$ROOT/bench.py" --graph "$g" --steps 5 --warmup 2 \
    --no-epochs --no-cpu-baseline --no-hbm-only --no-measure-traffic > "$OUT/bench_$g.json" 2> "$OUT/bench_$g.log"
  echo "$g rc=$?"
  tail -1 "$OUT/bench_$g.json" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(' ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'fwd_ms', r['forward_launch_ms'], 'bwd_ms', r['backward_launch_ms'], 'B/unit', round(r['bytes_per_edge_slice'],1), 'units', r['edge_slices_per_launch'], 'verify', d['verify'] and d['verify']['ok'])
print(' kernels', d['kernels_ms'])"
  find "$OUT/prof_$g" -name '*kernel_stats.csv' | head -1 | xargs -r head -8
  find "$OUT/prof_$g" -name '*kernel_trace.csv' -size +8M -delete
done
