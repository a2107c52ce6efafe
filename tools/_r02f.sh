mkdir -p gpurun_out/r02f
python3 -m pytest tests/test_gpu_kernels.py -x -q > gpurun_out/r02f/pytest_kernels.log 2>&1
echo "kernels rc=$?" > gpurun_out/r02f/status.log
python3 tools/perf_kernels.py dense gemm mtransform > gpurun_out/r02f/perf_kernels.txt 2>&1
