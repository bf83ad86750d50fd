#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python bench.py --steps 3 --warmup 1 --attack apgd --no-cpu-baseline ) > gpurun_out/bench_c3.log 2>&1
( timeout 1500 python bench.py --steps 2 --warmup 1 --attack autopgd --iterations 100 --batch 256 --no-cpu-baseline --no-roofline ) > gpurun_out/bench_c5.log 2>&1
( timeout 900 python bench.py --steps 3 --warmup 1 --mode train --no-cpu-baseline --no-roofline ) > gpurun_out/bench_train.log 2>&1
for f in c3 c5 train; do tail -1 gpurun_out/bench_$f.log | cut -c1-330; done
