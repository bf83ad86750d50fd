#!/bin/bash
mkdir -p gpurun_out
timeout 400 python scripts/sched_probe.py 0 2048 4096 > gpurun_out/sched_probe7.log 2>&1
tail -7 gpurun_out/sched_probe7.log
RVLM_GEMM_PRIO=1 timeout 400 python scripts/sched_probe.py 0 2048 4096 > gpurun_out/sched_probe7_prio.log 2>&1
tail -7 gpurun_out/sched_probe7_prio.log
