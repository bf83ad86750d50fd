#!/bin/bash
# Round 5, trip C: the pruned production GEMM file (measured form only) against the file it was cut from, kernel + engine tests,
# the experimental library's own tests, the 4-wave energy probe.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider ) > gpurun_out/t_kern.log 2>&1; tail -3 gpurun_out/t_kern.log
( RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/librvlm_exp.so timeout 900 python -m pytest tests/test_gpu_gemm_pingpong.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider ) > gpurun_out/t_exp.log 2>&1; tail -3 gpurun_out/t_exp.log
SKIP_TESTS=1 AB_VAR=RVLM_LIB_PATH AB_VALS="robustvlm_amd/librvlm_base.so robustvlm_amd/librvlm.so" AB_REPS=3 AB_STEPS=5 BENCH_ARGS="--no-pmc" bash scripts/trip_ab.sh
( PROBE_KINDS=1,4,5,6,7 timeout 300 python scripts/gemm_power_split.py 4 2 ) > gpurun_out/power_split.log 2>&1; cat gpurun_out/power_split.log
