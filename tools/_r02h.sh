mkdir -p gpurun_out/r02h
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "mtransform" > gpurun_out/r02h/pytest_mt.log 2>&1
echo "mt tests rc=$?" > gpurun_out/r02h/status.log
timeout 300 python3 tools/perf_kernels.py dense > gpurun_out/r02h/perf_dense.txt 2>&1
