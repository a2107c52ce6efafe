set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3e; mkdir -p $OUT
rm -f gpurun_out/tolerance_record.jsonl
python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -3 $OUT/pytest_gpu.log
python3 tools/tolerance_summary.py gpurun_out/tolerance_record.jsonl $OUT/tolerance_summary.json > $OUT/tolerance_summary.txt 2>&1
# RCCL at world size 1: both exchange modes at FULL S4 size through the collectives (the chunked all-gather: 2 chunks here), verified
timeout 900 python3 bench.py --force-collectives --steps 5 --warmup 2 --no-epochs --no-cpu-baseline --no-measure-traffic > $OUT/bench_force_collectives.json 2> $OUT/bench_force_collectives.err; echo "force-collectives rc=$?" >> $OUT/status.log
grep "ms/step\|verify\|plan" $OUT/bench_force_collectives.err
timeout 900 python3 bench.py --force-collectives --exchange allgather --gather-chunk-nodes 125000 --steps 5 --warmup 2 --no-epochs --no-cpu-baseline --no-measure-traffic --no-compare-exchange > $OUT/bench_force_allgather16.json 2> $OUT/bench_force_allgather16.err; echo "force-allgather16 rc=$?" >> $OUT/status.log
grep "ms/step\|verify" $OUT/bench_force_allgather16.err
bash tools/gpu_profiles.sh r3e_prof > $OUT/gpu_profiles.log 2>&1; echo "profiles rc=$?" >> $OUT/status.log
cat $OUT/status.log
