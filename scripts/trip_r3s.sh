#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ab_attn_persist.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do
  for ps in 0 1; do
    v=$(RVLM_ATTN_PERSIST=$ps timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']; print(round(d['value'],2), round(d['ms_per_step'],2), 'attn_bwd', pc['attn_bwd']['ms'], 'attn_fwd', pc['attn_fwd']['ms'], 'subset', round(d['roofline']['attention_gemm_subset']['frac'],4))")
    echo "RVLM_ATTN_PERSIST=$ps round $r: $v" | tee -a gpurun_out/ab_attn_persist.log
  done
done
