set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3k; mkdir -p $OUT
rm -f gpurun_out/tolerance_record.jsonl
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/status.log
python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -3 $OUT/pytest_gpu.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_20_5.json 2> $OUT/bench_n1_20_5.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?" >> $OUT/status.log
cat $OUT/bench_time.txt; cat $OUT/status.log
