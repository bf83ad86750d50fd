#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "clip_like_x3" ) 2>&1 | tail -5
rm -rf gpurun_out/prof_x3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --precision x3 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof_x3.log 2>&1
find gpurun_out/prof_x3 -name "*kernel_trace.csv" -delete 2>/dev/null
head -22 $(find gpurun_out/prof_x3 -name "*kernel_stats.csv") | cut -c1-150
