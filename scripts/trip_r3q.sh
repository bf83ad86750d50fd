#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/b32_fill.log
for f in 0.7 0.45 0.2 0.7; do
  v=$(RVLM_GEMM_MIN_FILL=$f timeout 300 python bench.py --model ViT-B-32 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2))")
  echo "RVLM_GEMM_MIN_FILL=$f: $v" | tee -a gpurun_out/b32_fill.log
done
