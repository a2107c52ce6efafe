set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3i; mkdir -p $OUT
TMGCN_CU_RESERVE=16 python3 -m pytest tests/test_gpu_dist1.py tests/test_gpu_dist2.py -x -q -m gpu > $OUT/pytest_dist_cu16.log 2>&1; echo "pytest(cu16) rc=$?" >> $OUT/status.log
tail -2 $OUT/pytest_dist_cu16.log
B="--force-collectives --steps 8 --warmup 2 --no-epochs --no-cpu-baseline --no-measure-traffic --no-compare-exchange --no-verify"
for cu in 0 8 16 32 64; do
  python3 bench.py $B --cu-reserve $cu > $OUT/bench_cu$cu.json 2> $OUT/bench_cu$cu.err; echo "cu=$cu rc=$? $(grep -h 'ms/step' $OUT/bench_cu$cu.err | sed 's/.*: //')" >> $OUT/status.log
done
python3 bench.py $B --cu-reserve 0 --grid-reserve 256 > $OUT/bench_grid256.json 2> $OUT/bench_grid256.err; echo "grid256 rc=$? $(grep -h 'ms/step' $OUT/bench_grid256.err | sed 's/.*: //')" >> $OUT/status.log
for cu in 0 16; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_cu$cu -- python3 bench.py $B --steps 4 --cu-reserve $cu > $OUT/bench_traced_cu$cu.json 2> $OUT/bench_traced_cu$cu.err
  python3 tools/timeline_gaps.py $OUT/trace_cu$cu --steps 2 > $OUT/gaps_cu$cu.json 2>&1
  echo "traced cu=$cu $(grep -h 'ms/step' $OUT/bench_traced_cu$cu.err | sed 's/.*: //')" >> $OUT/status.log
done
find $OUT -name "*.csv" -size +6M -delete
cat $OUT/status.log
