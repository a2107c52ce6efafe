#!/bin/bash
# Round 6 GPU-box session: the new -m gpu tests, then the driver-protocol bench line (and, with "prof", the same command under
# rocprofv3 --kernel-trace --stats).   usage: tools/r6_evidence.sh TAG [tests] [bench] [prof] [full]
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests) python -m pytest tests/test_gpu_real_operand_wide.py tests/test_gpu_s4_bench_size.py tests/test_gpu_kernels.py -x -q -m gpu > $out/tests_new.log 2>&1; tail -3 $out/tests_new.log;;
    full)  python -m pytest tests -x -q -m gpu > $out/tests_gpu.log 2>&1; tail -3 $out/tests_gpu.log
           test -f gpurun_out/tolerance_record.jsonl && python tools/tolerance_summary.py gpurun_out/tolerance_record.jsonl $out/tolerance_summary.json > /dev/null 2>&1;;
    bench) SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_20_steps_5_warmup.json 2> $out/bench.log; echo "bench wall ${SECONDS}s" >> $out/bench.log; grep -E "^\[bench" $out/bench.log | tail -70;;
    prof)  d=$out/prof; rm -rf $d
           rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-epochs > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.log
           f=$(find $d -name "*kernel_stats.csv" | head -1); test -n "$f" && cp $f $out/bench_kernel_stats.csv && head -12 $out/bench_kernel_stats.csv;;
  esac
done
