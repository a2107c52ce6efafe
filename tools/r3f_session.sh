set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3f; mkdir -p $OUT
python3 -m pytest tests/test_gpu_dist1.py tests/test_gpu_dist2.py -x -q -m gpu > $OUT/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -2 $OUT/pytest_dist.log
B="--force-collectives --steps 4 --warmup 2 --no-epochs --no-cpu-baseline --no-measure-traffic --no-compare-exchange --no-verify"
for lanes in 1 2; do
  TMGCN_PIPELINE_LANES=$lanes rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_lanes$lanes -- python3 bench.py $B > $OUT/bench_lanes$lanes.json 2> $OUT/bench_lanes$lanes.err
  echo "lanes=$lanes rc=$?" >> $OUT/status.log
  python3 tools/timeline_gaps.py $OUT/trace_lanes$lanes --steps 2 > $OUT/gaps_lanes$lanes.json 2>&1
done
# the same two without the profiler, with the slot reserve the multi-GPU path uses
for lanes in 1 2; do
  TMGCN_PIPELINE_LANES=$lanes python3 bench.py $B --steps 8 > $OUT/bench_plain_lanes$lanes.json 2> $OUT/bench_plain_lanes$lanes.err
  TMGCN_PIPELINE_LANES=$lanes python3 bench.py $B --steps 8 --grid-reserve 256 > $OUT/bench_plain_lanes${lanes}_reserve256.json 2> $OUT/bench_plain_lanes${lanes}_reserve256.err
done
grep -h "ms/step" $OUT/*.err
find $OUT -name "*.csv" -size +6M -delete
cat $OUT/status.log
