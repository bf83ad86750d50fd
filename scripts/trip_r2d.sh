#!/bin/bash
# round-2 trip D: full -m gpu suite, attention micro-bench, headline bench line, rocprofv3 kernel stats, HBM-traffic PMC passes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_metrics.jsonl
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt; nproc >> gpurun_out/device.txt
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
( RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py ) > gpurun_out/attn_bench.log 2>&1
grep -v amdgpu.ids gpurun_out/attn_bench.log
( timeout 900 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench.log
python - <<PY
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
pc=d['roofline']['per_class']
print("BENCH", round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), round(d['roofline']['attention_gemm_subset']['frac'],4), d['cpu_baseline']['value'],
      ' '.join(f"{k}={v['ms']:.2f}" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])[:13]))
PY
rm -rf gpurun_out/prof
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o trace -- \
    python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?" >> gpurun_out/rocprof.log
find gpurun_out/prof -name "*kernel_trace.csv" -delete 2>/dev/null
ls gpurun_out/prof/*/ 2>/dev/null | head
bash scripts/pmc_traffic.sh 2>&1 | tail -2
