#!/bin/bash
# round-2 trip G: socket power / clock of the 8-wave (shipped) and 4-wave (experimental build) persistent GEMMs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/librvlm_exp.so
( RVLM_GEMM_WAVES=8 timeout 300 python scripts/power_probe.py ) > gpurun_out/power_w8.log 2>&1
( RVLM_GEMM_WAVES=4 timeout 300 python scripts/power_probe.py ) > gpurun_out/power_w4.log 2>&1
( RVLM_GEMM_WAVES=4 timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_w4.log 2>&1
( RVLM_GEMM_WAVES=8 timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_w8.log 2>&1
echo "== 8 waves"; grep -v amdgpu gpurun_out/power_w8.log | head -3
echo "== 4 waves"; grep -v amdgpu gpurun_out/power_w4.log | head -3
echo "w8: "; grep -v "amdgpu" gpurun_out/gemm_w8.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
echo "w4: "; grep -v "amdgpu" gpurun_out/gemm_w4.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
