#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RVLM_GEMM_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/p_pytest.log 2>&1
tail -5 gpurun_out/p_pytest.log
RVLM_GEMM_PERSIST=1 RVLM_GEMM_SUPER=3 timeout 300 python scripts/gemm_bench.py 1 > gpurun_out/p_bench1.log 2>&1
grep -v amdgpu gpurun_out/p_bench1.log
RVLM_GEMM_PERSIST=1 timeout 300 python scripts/gemm_trace.py > gpurun_out/p_trace.log 2>&1
grep -v amdgpu gpurun_out/p_trace.log | grep -E "kernel|tile [01]:"
