#!/bin/bash
mkdir -p gpurun_out
timeout 500 python scripts/gemm_waits.py 128 129 132 133 896 897 900 901 898 > gpurun_out/gemm_waits4.log 2>&1
tail -40 gpurun_out/gemm_waits4.log
