#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -n 1 -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( timeout 600 python scripts/gemm_bench.py 0 1 ) > gpurun_out/gemm_bench.log 2>&1
( timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_v0.log 2>&1
( RVLM_GEMM_VARIANT=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_v1.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -3; cat gpurun_out/gemm_bench.log; tail -c 600 gpurun_out/bench_v0.log; echo; tail -c 600 gpurun_out/bench_v1.log
