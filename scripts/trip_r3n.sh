#!/bin/bash
mkdir -p gpurun_out; rm -rf gpurun_out/prof_train
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_train.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/prof_train -name "*kernel_trace.csv" -delete
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_train.log | cut -c1-200
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_train/trace_kernel_stats.csv | cut -c1-170
