#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RVLM_GEMM_PERSIST=1 timeout 300 python scripts/gemm_debug.py > gpurun_out/dbg.log 2>&1
grep -v amdgpu gpurun_out/dbg.log | grep -c "bad=0"; grep -v amdgpu gpurun_out/dbg.log | grep -A4 "bad=[1-9]" | head -20
RVLM_GEMM_PERSIST=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" > gpurun_out/p_pytest.log 2>&1
tail -3 gpurun_out/p_pytest.log
for pad in 0 64; do for a in 0 6; do echo "== ldpad $pad ablate $a"; GEMM_LDPAD=$pad GEMM_ABLATE=$a RVLM_GEMM_PERSIST=1 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -E "qkv|fc1_dgrad|cube8k"; done; done
