#!/bin/bash
# A/B on ONE box: whole-library build with the lockstep operand requests vs the role-split build, alternating
mkdir -p gpurun_out; rm -f gpurun_out/ab_split.log
for r in 1 2 3; do
  for lib in librvlm_lockstep.so librvlm.so; do
    v=$(RVLM_LIB_PATH=robustvlm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")
    echo "$lib round $r: $v" | tee -a gpurun_out/ab_split.log
  done
done
