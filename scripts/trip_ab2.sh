#!/bin/bash
# Same-box A/B over (library, environment) pairs: AB_CASES="lib.so:VAR=val,VAR2=val2 lib2.so:VAR=val ..." (paths relative to the repo),
# AB_REPS alternations of AB_STEPS pgd() calls each.  Output: gpurun_out/ab2.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/ab2.log
for rep in $(seq 1 ${AB_REPS:-2}); do
for c in $AB_CASES; do
  lib=${c%%:*}; envs=${c#*:}
  ( export RVLM_LIB_PATH=$GRAFT_REPO_ROOT/$lib; IFS=','; for e in $envs; do [ -n "$e" ] && [ "$e" != "_" ] && export $e; done; unset IFS
    timeout 600 python bench.py --steps ${AB_STEPS:-5} --warmup 1 --no-cpu-baseline --no-pmc ${BENCH_ARGS:-} ) > gpurun_out/bench_ab2.log 2>&1
  python - "$c" $rep <<'PY' | tee -a gpurun_out/ab2.log
import json, sys
ls=[x for x in open('gpurun_out/bench_ab2.log') if x.startswith('{')]
if not ls:
    print(sys.argv[1], "rep", sys.argv[2], "FAILED", open('gpurun_out/bench_ab2.log').read()[-300:])
else:
    d=json.loads(ls[-1]); pc=d['roofline']['per_class']
    print(sys.argv[1], "rep", sys.argv[2], round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1),
          ' '.join(f"{k.replace('gemm_','')}={v['ms']:.2f}" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])[:12]))
PY
done; done
