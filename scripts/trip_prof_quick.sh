#!/bin/bash
# rocprofv3 kernel stats of the headline bench (1 warm-up + 1 step)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} ) > gpurun_out/rocprof.log 2>&1
find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
grep -v "at::native" gpurun_out/prof/trace_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150 | head -${NROWS:-45}
