#!/bin/bash
# final-ish evidence refresh of round 2: benches of the other configs with the shipped kernels
mkdir -p gpurun_out
timeout 600 python bench.py --attack apgd --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_c3.log
timeout 900 python bench.py --attack autopgd --iterations 100 --batch 256 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_c5.log
timeout 600 python bench.py --mode train --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_train.log
timeout 600 python bench.py --model ViT-B-32 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_b32.log
for f in c3 c5 train b32; do python -c "
import json,sys
d=json.loads(open('gpurun_out/bench_$f.log').read())
print('$f', round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('workload','')[:80])"; done
