#!/bin/bash
# One GPU-box visit: smoke, the -m gpu parity tests, a bench line and a rocprofv3 kernel trace.
# Everything that must come back is written under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
( timeout 1800 python -m pytest tests -m gpu -q -n 1 -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  ( timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS:-} ) > gpurun_out/bench.log 2>&1
  echo "bench rc=$?" >> gpurun_out/bench.log
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  rm -rf gpurun_out/prof
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?" >> gpurun_out/rocprof.log
  find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null   # keep the small stats only
fi
tail -3 gpurun_out/smoke.log; tail -4 gpurun_out/pytest_gpu.log; tail -c 400 gpurun_out/bench.log; ls gpurun_out/prof
