#!/bin/bash
mkdir -p gpurun_out
timeout 400 python scripts/sched_probe.py 1024 0 256 > gpurun_out/sched_probe5.log 2>&1
tail -8 gpurun_out/sched_probe5.log
timeout 300 python scripts/gemm_waits.py 128 1152 > gpurun_out/gemm_waits5.log 2>&1
tail -9 gpurun_out/gemm_waits5.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
