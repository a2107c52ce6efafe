mkdir -p gpurun_out/r02g
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or spmm_gemm or fused" > gpurun_out/r02g/pytest_gemm.log 2>&1
echo "gemm tests rc=$?" > gpurun_out/r02g/status.log
timeout 300 python3 tools/ab_gemm.py > gpurun_out/r02g/ab_gemm.txt 2>&1
timeout 300 python3 tools/perf_kernels.py gemm > gpurun_out/r02g/perf_gemm.txt 2>&1
