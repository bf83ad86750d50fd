#!/bin/bash
mkdir -p gpurun_out
for a in 0 256 4 5 1; do
  GEMM_ABLATE=$a timeout 200 python scripts/gemm_bench.py 2 2>&1 | grep -v amdgpu.ids | grep "qkv\|fc1_dgrad\|cube" | sed "s/^/abl$a /" >> gpurun_out/gemm_abl_power.log
done
cat gpurun_out/gemm_abl_power.log
