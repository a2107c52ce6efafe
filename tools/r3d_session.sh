set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3d; mkdir -p $OUT
rm -f gpurun_out/tolerance_record.jsonl
python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -3 $OUT/pytest_gpu.log
python3 tools/tolerance_summary.py gpurun_out/tolerance_record.jsonl $OUT/tolerance_summary.json > $OUT/tolerance_summary.txt 2>&1
( time python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?" >> $OUT/status.log
grep -i "traffic\|verify\|plan" $OUT/bench_n1.err | tail; cat $OUT/bench_time.txt
cat $OUT/status.log
