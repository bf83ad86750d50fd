#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/tests_split.log 2>&1
tail -5 gpurun_out/tests_split.log
timeout 300 python scripts/gemm_bench.py 2 > gpurun_out/gemm_bench_split.log 2>&1
cat gpurun_out/gemm_bench_split.log | tail -9
timeout 600 python bench.py > gpurun_out/bench_split.log 2>&1
tail -2 gpurun_out/bench_split.log
