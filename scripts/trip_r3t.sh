#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ab_ln_loop.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm or ln" 2>&1 | tail -2
for r in 1 2 3; do
  for ps in 0 1; do
    v=$(RVLM_LN_LOOP=$ps timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']; print(round(d['value'],2), round(d['ms_per_step'],2), 'ln_fwd', pc['layernorm_fwd']['ms'], pc['layernorm_fwd']['gbps'], 'ln_bwd', pc['layernorm_bwd']['ms'], pc['layernorm_bwd']['gbps'])")
    echo "RVLM_LN_LOOP=$ps round $r: $v" | tee -a gpurun_out/ab_ln_loop.log
  done
done
