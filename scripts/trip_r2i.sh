#!/bin/bash
# schedule experiments on the persistent GEMM: DMA pieces spread among MFMAs (16), staggered per SIMD-sharing wave (32), priority (64)
mkdir -p gpurun_out
timeout 600 python scripts/sched_probe.py 0 16 48 64 112 > gpurun_out/sched_probe.log 2>&1
tail -20 gpurun_out/sched_probe.log
