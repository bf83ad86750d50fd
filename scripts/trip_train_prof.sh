#!/bin/bash
# rocprofv3 kernel stats of the full training step (bench.py --mode train)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_train
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --mode train --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof_train.log 2>&1
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete 2>/dev/null
tail -1 gpurun_out/rocprof_train.log | cut -c1-300
find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1 | xargs head -30
