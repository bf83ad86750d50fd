#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/gemm_timeline.py 2048 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_timeline.log
timeout 300 python scripts/gemm_timeline.py 2053 2>&1 | grep -v amdgpu.ids >> gpurun_out/gemm_timeline.log
cat gpurun_out/gemm_timeline.log
