#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ab_odd.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -4
for r in 1 2 3; do
  for odd in 0 1; do
    v=$(RVLM_GEMM_ODD=$odd timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")
    echo "RVLM_GEMM_ODD=$odd round $r: $v" | tee -a gpurun_out/ab_odd.log
  done
done
