#!/bin/bash
# round-2 trip H: operand-delivery experiments on the persistent GEMM: K-step rotation per workgroup, leading-dimension padding
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for kr in 0 4 16 -2 -4 2; do
  ( RVLM_GEMM_KROT=$kr timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_krot$kr.log 2>&1
  echo "KROT=$kr: "; grep -v "amdgpu" gpurun_out/gemm_krot$kr.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
done
for pad in 64 192 8; do
  ( GEMM_LDPAD=$pad timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_pad$pad.log 2>&1
  echo "LDPAD=$pad: "; grep -v "amdgpu" gpurun_out/gemm_pad$pad.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
done
