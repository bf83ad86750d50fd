#!/bin/bash
mkdir -p gpurun_out
timeout 400 python scripts/sched_probe.py 0 256 768 > gpurun_out/sched_probe3.log 2>&1
tail -12 gpurun_out/sched_probe3.log
timeout 300 python scripts/gemm_waits.py 384 896 > gpurun_out/gemm_waits2.log 2>&1
tail -9 gpurun_out/gemm_waits2.log
