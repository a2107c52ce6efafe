#!/bin/bash
# The GPU-box session behind the round-6 evidence files (profiles/r6/).   usage: tools/r6_evidence.sh TAG [tests] [full] [bench] [prof] [legs] [mfma]
#   tests  the round's new -m gpu tests            full  the whole -m gpu suite + parity-clause / tolerance summaries
#   bench  the driver-protocol line (python bench.py --gpus 1 --steps 20 --warmup 5, every leg)
#   prof   the same command's GPU legs under rocprofv3 --kernel-trace --stats (line + kernel stats of ONE process)
#   legs   the T = 128 and the real-structure workloads, each under rocprofv3 --kernel-trace --stats (per-kernel rows of the side legs)
#   mfma   matrix-core busy fraction + shader clock of the kernels of the headline and of the real-structure workload (PMC pass, kernel trace only)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
stats() {   # stats DIR OUTFILE: the kernel-stats CSV of a rocprofv3 --stats run
  f=$(find $1 -name "*kernel_stats.csv" | head -1); test -n "$f" && cp $f $2 && head -8 $2; rm -rf $1
}
for what in "$@"; do
  case $what in
    tests) python -m pytest tests/test_gpu_real_operand_wide.py tests/test_gpu_s4_bench_size.py tests/test_gpu_kernels.py tests/test_gpu_experiment.py -x -q -m gpu > $out/tests_new.log 2>&1; tail -3 $out/tests_new.log;;
    full)  python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $out/status.log; tail -3 $out/pytest_gpu.log; cat $out/status.log
           cp gpurun_out/parity_clauses.json $out/parity_clauses.json 2>/dev/null; cp gpurun_out/g11_trajectory.json $out/g11_trajectory.json 2>/dev/null
           python tools/tolerance_summary.py gpurun_out/tolerance_record.jsonl $out/tolerance_summary.json > /dev/null 2>&1;;
    bench) SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_20_steps_5_warmup.json 2> $out/bench.log; echo "bench wall ${SECONDS}s" >> $out/bench.log
           grep -E "^\[bench" $out/bench.log | grep -E "ms/step|verify|done"; tail -1 $out/bench.log;;
    prof)  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-epochs > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.log
           stats $out/prof $out/bench_kernel_stats.csv;;
    legs)  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_t128 -- python3 bench.py --gpus 1 --steps 5 --warmup 2 --nodes 250000 --slices-per-gpu 128 --verify-slices 16 --no-cpu-baseline --no-epochs --no-legs --no-hbm-only --no-measure-traffic > $out/T128_under_rocprof.json 2> $out/T128_under_rocprof.log
           stats $out/prof_t128 $out/T128_kernel_stats.csv
           rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_real -- python3 bench.py --gpus 1 --steps 5 --warmup 2 --graph chess_tiled --no-cpu-baseline --no-epochs --no-legs --no-hbm-only --no-measure-traffic > $out/real_structure_under_rocprof.json 2> $out/real_structure_under_rocprof.log
           stats $out/prof_real $out/real_structure_kernel_stats.csv;;
    mfma)  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-epochs --no-verify --no-measure-traffic --no-hbm-only --no-legs > /dev/null 2> $out/pmc_mfma.log
           python3 tools/mfma_util.py $out/pmc_mfma > $out/mfma_utilisation_bench.json 2>> $out/pmc_mfma.log; rm -rf $out/pmc_mfma
           rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma2 -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --graph chess_tiled --no-cpu-baseline --no-epochs --no-verify --no-measure-traffic --no-hbm-only --no-legs > /dev/null 2> $out/pmc_mfma_real.log
           python3 tools/mfma_util.py $out/pmc_mfma2 > $out/mfma_utilisation_real_structure.json 2>> $out/pmc_mfma_real.log; rm -rf $out/pmc_mfma2
           cat $out/mfma_utilisation_real_structure.json | head -40;;
  esac
done
