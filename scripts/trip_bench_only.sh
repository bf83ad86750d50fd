#!/bin/bash
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name" | head -2 | tr '\n' ' '
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_run.log 2>/dev/null
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/bench_run.log') if x.startswith('{')][-1])
print("BENCH", round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4), round(d['roofline']['attention_gemm_subset']['frac'],4), d['cpu_baseline']['value'])
PY
