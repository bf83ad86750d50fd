#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/gemm_trace.py 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_trace_split.log
cat gpurun_out/gemm_trace_split.log
