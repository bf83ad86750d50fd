#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RVLM_GEMM_SUPER=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/k64_pytest.log 2>&1
tail -3 gpurun_out/k64_pytest.log
GEMM_BENCH_TORCH=1 RVLM_GEMM_SUPER=0 timeout 300 python scripts/gemm_bench.py 1 > gpurun_out/k64_bench0.log 2>&1
RVLM_GEMM_SUPER=3 timeout 300 python scripts/gemm_bench.py 1 > gpurun_out/k64_bench3.log 2>&1
paste -d'\n' gpurun_out/k64_bench0.log gpurun_out/k64_bench3.log | grep -v amdgpu.ids
