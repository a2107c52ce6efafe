#!/bin/bash
# First contact with more than one GPU: what to run, in this order, and what to look at.
# Nothing here has ever run on a multi-GPU box (none was available through round 3); every step writes
# its record under gpurun_out/TAG/ so that a partial session is still evidence.
#   usage: tools/multigpu_first_contact.sh TAG [MAX_GPUS]
TAG=${1:-mg}; MAX=${2:-$(python3 -c "import torch; print(torch.cuda.device_count())")}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC: RCCL needs it on this driver
echo "GPUs visible: $MAX" | tee "$OUT/status.log"
[ "$MAX" -ge 2 ] || { echo "needs >= 2 GPUs"; exit 0; }

# 1. correctness over RCCL: the sharded layer (both exchange modes, chunked == literal all-gather), five
#    sharded model fixtures, and bench.py verifying itself — every case with a deadline
timeout 3000 python3 -m pytest tests/test_gpu_multi.py -q -m gpu > "$OUT/pytest_multi.log" 2>&1
echo "pytest test_gpu_multi rc=$?" | tee -a "$OUT/status.log"; tail -3 "$OUT/pytest_multi.log"

# 2. the scaling curve as the driver runs it (look at: value, ms_per_step, phases_ms, verify.ok,
#    exchange_compare.full_n — DESIGN.md §6 has the predicted table; tools/predict_scaling.py reproduces it)
for g in 1 2 4 8; do
  [ "$g" -le "$MAX" ] || continue
  timeout 1800 python3 bench.py --gpus $g --steps 20 --warmup 5 --no-epochs --no-cpu-baseline > "$OUT/scale_g$g.json" 2> "$OUT/scale_g$g.err"
  echo "bench --gpus $g rc=$? $(grep -h 'headline' "$OUT/scale_g$g.err" | tail -1 | sed 's/.*headline: //')" | tee -a "$OUT/status.log"
done

# 3. the one knob that can only be tuned here: CUs left to RCCL's kernels (and round 2's slot reserve for reference)
G=$MAX; [ "$G" -gt 8 ] && G=8
for cu in 0 16 32 64; do
  timeout 900 python3 bench.py --gpus $G --steps 10 --warmup 3 --no-compare-exchange --no-verify --cu-reserve $cu > "$OUT/cu${cu}_g$G.json" 2> "$OUT/cu${cu}_g$G.err"
  echo "cu_reserve=$cu g=$G rc=$? $(grep -h 'ms/step' "$OUT/cu${cu}_g$G.err" | grep ' r0 ' | tail -1 | sed 's/.*: //')" | tee -a "$OUT/status.log"
done
timeout 900 python3 bench.py --gpus $G --steps 10 --warmup 3 --no-compare-exchange --no-verify --cu-reserve 0 --grid-reserve 256 > "$OUT/grid256_g$G.json" 2> "$OUT/grid256_g$G.err"
echo "grid_reserve=256 g=$G rc=$? $(grep -h 'ms/step' "$OUT/grid256_g$G.err" | grep ' r0 ' | tail -1 | sed 's/.*: //')" | tee -a "$OUT/status.log"
TMGCN_PIPELINE_LANES=1 timeout 900 python3 bench.py --gpus $G --steps 10 --warmup 3 --no-compare-exchange --no-verify > "$OUT/lanes1_g$G.json" 2> "$OUT/lanes1_g$G.err"
echo "lanes=1 g=$G rc=$? $(grep -h 'ms/step' "$OUT/lanes1_g$G.err" | grep ' r0 ' | tail -1 | sed 's/.*: //')" | tee -a "$OUT/status.log"

# 4. where the time goes: kernel trace of rank 0's process tree is not separable under torchrun; trace the
#    2-rank case instead and decompose a step (RCCL kernel time, overlap with the fused kernel, idle gaps)
# the launcher starts FIRST and every rank is profiled directly (the profiled program sits right behind `--`, with no
# process hop behind it: WORLD_SIZE is set by torchrun, so bench.py runs its worker without spawning anything)
python3 -m torch.distributed.run --no-python --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 \
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_g2" -- python3 bench.py --gpus 2 --steps 4 --warmup 2 \
  --no-compare-exchange --no-verify --no-measure-traffic > "$OUT/traced_g2.json" 2> "$OUT/traced_g2.err"
for f in $(find "$OUT/trace_g2" -name "*kernel_trace.csv" | head -4); do python3 tools/timeline_gaps.py "$f" --steps 2 > "$f.gaps.json" 2>&1; done
find "$OUT" -name "*.csv" -size +6M -delete
cat "$OUT/status.log"
