set -x
export TMPDIR=/tmp
OUT=gpurun_out/r3g; mkdir -p $OUT
rm -f gpurun_out/tolerance_record.jsonl
python3 -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/status.log
tail -3 $OUT/pytest_gpu.log
python3 tools/tolerance_summary.py gpurun_out/tolerance_record.jsonl $OUT/tolerance_summary.json > $OUT/tolerance_summary.txt 2>&1
for g in 8; do
  t0=$(date +%s)
  timeout 1200 python3 bench.py --gpus $g --backend gloo --single-device --nodes 250000 --steps 3 --warmup 1 --deadline 900 --watchdog 400 > $OUT/emul_g$g.json 2> $OUT/emul_g$g.err
  echo "emul g=$g rc=$? wall=$(( $(date +%s) - t0 ))s" >> $OUT/status.log
done
grep "verify\|ms/step" $OUT/emul_g8.err | grep "r0 " | tail -8
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/status.log
cat $OUT/status.log
