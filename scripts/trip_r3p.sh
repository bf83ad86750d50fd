#!/bin/bash
# re-tune the launch knobs under the split-role schedule: epilogue stagger, tile-order group size
mkdir -p gpurun_out; rm -f gpurun_out/gemm_knobs.log
run() { echo "== $1" >> gpurun_out/gemm_knobs.log; env $1 timeout 200 python scripts/gemm_bench.py 2 2>&1 | grep -v amdgpu.ids | grep -v cube >> gpurun_out/gemm_knobs.log; }
run "RVLM_GEMM_STAGGER=2"
run "RVLM_GEMM_STAGGER=0"
run "RVLM_GEMM_STAGGER=2 RVLM_GEMM_STAGGER_EPI=15"
run "RVLM_GEMM_STAGGER=1 RVLM_GEMM_STAGGER_EPI=15"
run "RVLM_GEMM_GROUP_M=4"
run "RVLM_GEMM_GROUP_M=16"
run "RVLM_GEMM_STAGGER=2"
cat gpurun_out/gemm_knobs.log
