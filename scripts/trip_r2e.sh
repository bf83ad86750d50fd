#!/bin/bash
# round-2 trip E: the other BASELINE configs and the training step, for profiles/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python bench.py --attack apgd --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_c3.log 2>&1
( timeout 900 python bench.py --attack autopgd --iterations 100 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_c5.log 2>&1
( timeout 900 python bench.py --mode train --steps 3 --warmup 1 ) > gpurun_out/bench_train.log 2>&1
( timeout 600 python bench.py --model ViT-B-32 --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_b32.log 2>&1
for f in c3 c5 train b32; do python - <<PY
import json
try:
    l=[x for x in open('gpurun_out/bench_$f.log') if x.startswith('{')][-1]
    d=json.loads(l); print("$f", round(d['value'],2), round(d['ms_per_step'],1), d['metric'])
except Exception as e:
    print("$f FAILED", e); print(open('gpurun_out/bench_$f.log').read()[-1500:])
PY
done
