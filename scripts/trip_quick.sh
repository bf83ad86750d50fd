#!/bin/bash
# quick loop: selected GPU tests + 2 headline bench runs with the per-class table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
for i in 1 2; do
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} ) > gpurun_out/bench_q$i.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_q$i.log') if x.startswith('{')][-1]
d=json.loads(l)
print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1))
pc=d['roofline']['per_class']
print(' '.join(f"{k}={v['ms']:.2f}" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])))
PY
done
