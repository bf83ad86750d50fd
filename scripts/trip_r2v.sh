#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ab_split2.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
  for lib in librvlm_lockstep.so librvlm.so; do
    v=$(RVLM_LIB_PATH=robustvlm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")
    echo "$lib round $r: $v" | tee -a gpurun_out/ab_split2.log
  done
done
