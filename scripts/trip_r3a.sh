#!/bin/bash
mkdir -p gpurun_out
for hm in 0 1; do
  echo "== RVLM_ATTN_HM=$hm" >> gpurun_out/attn_hm.log
  RVLM_ATTN_HM=$hm RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/attn_hm.log
done
cat gpurun_out/attn_hm.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k attn 2>&1 | tail -2
