#!/bin/bash
# Round evidence (PYTEST=1: the whole -m gpu suite in front): headline bench line as the driver runs it, rocprofv3 kernel stats of
# the same command, the in-pipeline PMC passes, the HBM-traffic PMC passes, the other configurations.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt; nproc >> gpurun_out/device.txt
if [ "${PYTEST:-0}" = "1" ]; then
  rm -f gpurun_out/parity_metrics.jsonl
  ( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 ) > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
  ( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
fi
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; grep -o '"value": [0-9.]*' gpurun_out/bench.log | head -1
rm -rf gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/rocprof.log
find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
bash scripts/pmc_pipeline.sh > gpurun_out/pmc_pipeline.log 2>&1; tail -2 gpurun_out/pmc_pipeline.log | cut -c1-200
bash scripts/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; tail -2 gpurun_out/pmc_traffic.log | cut -c1-200
bash scripts/trip_configs.sh 2>&1 | cut -c1-250

