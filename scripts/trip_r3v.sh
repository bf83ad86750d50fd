#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/gemm_trace.py 2>&1 | grep -v amdgpu.ids | grep "kernel\|tiles/WG\|before the first\|tile 0" > gpurun_out/gemm_trace_tail.log
cat gpurun_out/gemm_trace_tail.log
