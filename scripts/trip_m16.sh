#!/bin/bash
# 16x16x32 MFMA variant of the persistent GEMM: kernel parity tests under it, then same-box micro-benchmark alternations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( RVLM_GEMM_M16=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm" ) 2>&1 | tail -4
for rep in 1 2; do for m in 0 1; do
  echo "== rep $rep RVLM_GEMM_M16=$m"
  RVLM_GEMM_M16=$m python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu.ids | grep "variant=1"
done; done 2>&1 | tee gpurun_out/m16_gemm_bench.log
