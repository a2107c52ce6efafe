#!/bin/bash
# One GPU-box session: multi-rank emulation of bench.py (self-launched ranks sharing the one GPU over
# gloo), the -m gpu tests, and the default bench line.  Usage: tools/gpu_check.sh TAG [emul] [tests] [bench]
# Everything lands in gpurun_out/TAG/ (copy what should be judged into profiles/).
TAG=${1:-run}; shift
WHAT=${*:-emul tests bench}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
{ echo "hostname: $(hostname)"; ( time python3 -c "import socket; print(socket.gethostbyname(socket.gethostname()))" ) 2>&1; } > "$OUT/hostname_resolution.log" 2>&1
for w in $WHAT; do
  case $w in
    emul)
      for g in 2 4 8; do
        t0=$(date +%s)
        timeout 900 python3 bench.py --gpus $g --backend gloo --single-device --nodes 250000 \
          --steps 3 --warmup 1 --deadline 600 --watchdog 240 > "$OUT/emul_g$g.json" 2> "$OUT/emul_g$g.err"
        echo "emul g=$g rc=$? wall=$(( $(date +%s) - t0 ))s" >> "$OUT/status.log"
      done ;;
    tests)
      python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
      echo "pytest rc=$?" >> "$OUT/status.log" ;;
    bench)
      python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
      echo "bench rc=$?" >> "$OUT/status.log" ;;
  esac
done
cat "$OUT/status.log"
