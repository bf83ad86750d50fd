#!/bin/bash
mkdir -p gpurun_out
timeout 400 python scripts/sched_probe.py 2048 0 1024 > gpurun_out/sched_probe6.log 2>&1
tail -8 gpurun_out/sched_probe6.log
timeout 300 python scripts/gemm_waits.py 128 2176 > gpurun_out/gemm_waits6.log 2>&1
tail -9 gpurun_out/gemm_waits6.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
