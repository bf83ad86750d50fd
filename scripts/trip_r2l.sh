#!/bin/bash
mkdir -p gpurun_out
timeout 400 python scripts/sched_probe.py 0 768 1792 2816 1280 > gpurun_out/sched_probe4.log 2>&1
tail -12 gpurun_out/sched_probe4.log
RVLM_GEMM_PRIO=1 timeout 400 python scripts/sched_probe.py 0 768 1792 > gpurun_out/sched_probe4_prio.log 2>&1
tail -8 gpurun_out/sched_probe4_prio.log
timeout 300 python scripts/gemm_waits.py 1920 > gpurun_out/gemm_waits3.log 2>&1
tail -5 gpurun_out/gemm_waits3.log
