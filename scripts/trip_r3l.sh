#!/bin/bash
# same-box A/B: the library as of the commit before the 257-row tiles (b6554db) vs the current one
mkdir -p gpurun_out; rm -f gpurun_out/ab_prev.log
timeout 300 python scripts/odd_debug.py 2>&1 | grep -v amdgpu.ids | grep "families"
for r in 1 2 3; do
  for lib in librvlm_prev.so librvlm.so; do
    v=$(RVLM_LIB_PATH=robustvlm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); pc=d['roofline']['per_class']; print(round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), ' '.join(f\"{k[5:]}={pc[k]['ms']:.1f}\" for k in ['gemm_qkv_fwd','gemm_qkv_bwd','gemm_fc1_bwd','gemm_out_bwd','gemm_fc1_fwd','gemm_fc2_fwd','gemm_fc2_bwd','gemm_out_fwd']))")
    echo "$lib round $r: $v" | tee -a gpurun_out/ab_prev.log
  done
done
