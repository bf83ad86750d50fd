#!/bin/bash
# The L2 threat model at full size: the property test, then bench lines of the three entry points (pgd / apgd_train / APGDAttack, norm l2).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "l2_threat" ) > gpurun_out/t_l2.log 2>&1; tail -3 gpurun_out/t_l2.log
for a in pgd apgd; do
  ( timeout 600 python scripts/bench_extra.py --attack $a --norm l2 --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_l2_$a.log 2>&1
  grep '^{' gpurun_out/bench_l2_$a.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a l2', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms', d['config']['workload'][:90])" || tail -5 gpurun_out/bench_l2_$a.log
done
( timeout 900 python scripts/bench_extra.py --attack autopgd --norm l2 --batch 256 --iterations 100 --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_l2_autopgd.log 2>&1
grep '^{' gpurun_out/bench_l2_autopgd.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('autopgd l2', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms')" || tail -5 gpurun_out/bench_l2_autopgd.log
( timeout 600 python bench.py --attack autopgd --batch 256 --iterations 100 --steps 1 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_linf_autopgd.log 2>&1
grep '^{' gpurun_out/bench_linf_autopgd.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('autopgd linf', round(d['value'],1), d['unit'], round(d['ms_per_step'],1), 'ms')" || tail -5 gpurun_out/bench_linf_autopgd.log
