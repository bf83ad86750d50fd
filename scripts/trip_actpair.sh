#!/bin/bash
# activation-pair epilogue without register spills + "strip first for every m-th workgroup": parity, then same-box A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_gemm_pingpong.py -m gpu -q -x -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
AB_VAR=RVLM_LIB_PATH AB_VALS="robustvlm_amd/librvlm_base.so robustvlm_amd/librvlm.so" AB_REPS=2 SKIP_TESTS=1 bash scripts/trip_ab.sh
cp gpurun_out/ab.log gpurun_out/ab_actpair.log
AB_VAR=RVLM_GEMM_STRIP_FIRST AB_VALS="0 2 3" AB_REPS=2 SKIP_TESTS=1 bash scripts/trip_ab.sh
cp gpurun_out/ab.log gpurun_out/ab_strip_first.log
