#!/bin/bash
mkdir -p gpurun_out
timeout 400 python scripts/sched_probe.py 0 4096 8192 16384 > gpurun_out/sched_probe8.log 2>&1
tail -7 gpurun_out/sched_probe8.log
timeout 300 python scripts/gemm_timeline.py 6144 2>&1 | grep -v amdgpu.ids | head -10 | cut -c1-330 > gpurun_out/gemm_timeline2.log
cat gpurun_out/gemm_timeline2.log
