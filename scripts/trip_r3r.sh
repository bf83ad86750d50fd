#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/attn_persist.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -3
for ps in 0 1 0 1; do
  echo "== RVLM_ATTN_PERSIST=$ps" >> gpurun_out/attn_persist.log
  RVLM_ATTN_PERSIST=$ps RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/attn_persist.log
done
cat gpurun_out/attn_persist.log
