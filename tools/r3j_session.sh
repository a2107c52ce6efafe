set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3j; mkdir -p $OUT
python3 -m pytest tests/test_gpu_dist1.py tests/test_gpu_dist2.py tests/test_gpu_dist_models.py -x -q -m gpu > $OUT/pytest_dist.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -2 $OUT/pytest_dist.log
for g in 2 8; do
  timeout 900 python3 bench.py --gpus $g --backend gloo --single-device --nodes 100000 --steps 2 --warmup 1 --deadline 600 --no-compare-exchange > $OUT/emul_g$g.json 2> $OUT/emul_g$g.err
  echo "emul g=$g rc=$? $(python3 -c "import json; d=json.loads(open('$OUT/emul_g$g.json').read().strip().split('\n')[-1]); print(d['verify']['ok'], d['config']['cu_reserve'], d['config']['grid_reserve'])")" >> $OUT/status.log
done
cat $OUT/status.log
