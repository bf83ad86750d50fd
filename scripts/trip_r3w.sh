#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/strip_first.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
for sf in 0 1 0 1; do
  echo "== RVLM_GEMM_STRIP_FIRST=$sf" >> gpurun_out/strip_first.log
  RVLM_GEMM_STRIP_FIRST=$sf timeout 200 python scripts/gemm_bench.py 2 2>&1 | grep -v amdgpu.ids | grep -v cube >> gpurun_out/strip_first.log
done
cat gpurun_out/strip_first.log
