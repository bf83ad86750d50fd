#!/bin/bash
# the phase-shifted persistent GEMM: parity tests, then the same-process A/B + timeline (gpurun_out/pingpong_*.log)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_gemm_pingpong.py -m gpu -q -x -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pingpong_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pingpong_tests.log
tail -15 gpurun_out/pingpong_tests.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
( timeout 600 python scripts/pingpong_bench.py ${PP_REPS:-30} ) > gpurun_out/pingpong_bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/pingpong_bench.log
grep -v "^    wg" gpurun_out/pingpong_bench.log | tail -24
fi
