#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RVLM_GEMM_WAVES=4 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/q_pytest.log 2>&1
tail -5 gpurun_out/q_pytest.log
echo "== 4 waves"; RVLM_GEMM_WAVES=4 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu
echo "== 8 waves"; RVLM_GEMM_WAVES=8 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu
