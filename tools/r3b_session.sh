set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3b; mkdir -p $OUT
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_dist1.py tests/test_gpu_dist2.py tests/test_gpu_dist_models.py tests/test_gpu_multi.py -x -q -m gpu > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -5 $OUT/pytest_new.log
timeout 900 python3 bench.py --no-epochs --no-cpu-baseline > $OUT/bench_n1_verify.json 2> $OUT/bench_n1_verify.err; echo "bench rc=$?" >> $OUT/status.log
tail -3 $OUT/bench_n1_verify.err
for g in 2 4; do
  timeout 900 python3 bench.py --gpus $g --backend gloo --single-device --nodes 250000 --steps 3 --warmup 1 --deadline 600 --watchdog 240 > $OUT/emul_g$g.json 2> $OUT/emul_g$g.err; echo "emul g=$g rc=$?" >> $OUT/status.log
done
cat $OUT/status.log
