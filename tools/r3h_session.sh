set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3h; mkdir -p $OUT
python3 -m pytest tests/test_gpu_abi_driver.py -q -m gpu > $OUT/pytest_abi.log 2>&1; echo "abi driver rc=$?" >> $OUT/status.log
# soak: self-launched multi-rank bench (gloo, one device) with the verify leg, back to back
for i in 1 2 3; do for g in 2 4 8; do
  t0=$(date +%s)
  timeout 600 python3 bench.py --gpus $g --backend gloo --single-device --nodes $((40000 * i + 20000)) --steps 2 --warmup 1 --deadline 400 --watchdog 200 --no-compare-exchange > $OUT/soak_${i}_g$g.json 2> $OUT/soak_${i}_g$g.err
  rc=$?; ok=$(python3 -c "import json,sys; d=json.loads(open('$OUT/soak_${i}_g$g.json').read().strip().split('\n')[-1]); print(d['verify']['ok'], d['verify']['seconds'])" 2>/dev/null)
  echo "soak $i g=$g rc=$rc verify=$ok wall=$(( $(date +%s) - t0 ))s" >> $OUT/status.log
done; done
for e in a2a allgather; do
  timeout 600 python3 bench.py --gpus 4 --backend gloo --single-device --exchange $e --nodes 60000 --steps 2 --warmup 1 --deadline 400 --no-compare-exchange --gather-chunk-nodes 7000 > $OUT/soak_ex_$e.json 2> $OUT/soak_ex_$e.err
  echo "exchange $e g=4 rc=$? $(python3 -c "import json; d=json.loads(open('$OUT/soak_ex_$e.json').read().strip().split('\n')[-1]); print(d['verify']['ok'], d['config'].get('gather_chunks'))")" >> $OUT/status.log
done
cat $OUT/status.log
