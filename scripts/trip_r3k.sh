#!/bin/bash
mkdir -p gpurun_out
RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | head -3
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']
print('BENCH', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), ' '.join(f\"{k}={v['ms']:.2f}\" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])[:12]))"
