#!/bin/bash
# A/B on ONE box (boxes differ by ~2 %): tests, then the headline bench under each setting of an env switch.
# usage: AB_VAR=RVLM_GEMM_TAIL AB_VALS="0 1" bash scripts/trip_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
fi
for rep in 1 2; do
for v in ${AB_VALS:-0 1}; do
( export ${AB_VAR:-RVLM_GEMM_TAIL}=$v; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} ) > gpurun_out/bench_ab_$v.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_ab_$v.log') if x.startswith('{')][-1]
d=json.loads(l)
pc=d['roofline']['per_class']
print("${AB_VAR:-RVLM_GEMM_TAIL}=$v", round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1),
      ' '.join(f"{k}={v['ms']:.2f}" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])[:12]))
PY
done
done
