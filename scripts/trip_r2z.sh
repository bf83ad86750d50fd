#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/gemm_waits.py 128 4224 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_waits7.log
cat gpurun_out/gemm_waits7.log
