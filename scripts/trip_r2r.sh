#!/bin/bash
# round-2 evidence refresh after the role-split schedule: full GPU suite, bench, kernel trace + stats, HBM traffic PMC
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/tests_full.log 2>&1
tail -3 gpurun_out/tests_full.log
timeout 600 python bench.py > gpurun_out/bench_r02b.log 2>&1
tail -1 gpurun_out/bench_r02b.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02b -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r02b.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_r02b.log | cut -c1-300
ls $GRAFT_REPO_ROOT/gpurun_out/prof_r02b
