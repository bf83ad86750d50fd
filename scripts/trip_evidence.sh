#!/bin/bash
# Round evidence in one visit: the parity tests that carry measured bars, the headline bench line, rocprofv3 kernel stats of
# the same command, the in-pipeline PMC passes and the HBM-traffic PMC passes.  Everything lands under gpurun_out/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt; nproc >> gpurun_out/device.txt
rm -f gpurun_out/parity_metrics.jsonl
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
( timeout 900 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log; tail -c 300 gpurun_out/bench.log
rm -rf gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/rocprof.log
find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
bash scripts/pmc_pipeline.sh
bash scripts/pmc_traffic.sh
