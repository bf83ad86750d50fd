#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/attn_pf.log
for pf in 0 256 512 128; do
  echo "== RVLM_ATTN_PREFETCH=$pf" >> gpurun_out/attn_pf.log
  RVLM_ATTN_PREFETCH=$pf RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/attn_pf.log
done
cat gpurun_out/attn_pf.log
