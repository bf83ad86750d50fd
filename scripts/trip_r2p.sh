#!/bin/bash
mkdir -p gpurun_out
GEMM_BENCH_TORCH=1 timeout 300 python scripts/gemm_bench.py 2 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_bench_vendor.log
cat gpurun_out/gemm_bench_vendor.log
