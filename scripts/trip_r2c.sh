#!/bin/bash
# round-2 trip C: fused2 attention backward variants (A = main lib, B = early transposed reads), no-MFMA ablation of the
# GEMM epilogues (evidence: what bounds a tile), headline bench with the fused2 backward
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in librvlm.so librvlm_earlytr.so; do
  export RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib
  echo "== $lib"
  ( RVLM_ATTN_FUSED=2 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "attention or library" ) > gpurun_out/attn_test_$lib.log 2>&1
  tail -2 gpurun_out/attn_test_$lib.log
  ( RVLM_ATTN_FUSED=2 RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py ) > gpurun_out/attn_bench_$lib.log 2>&1
  ( RVLM_ATTN_FUSED=2 timeout 300 python scripts/attn_bench.py ) >> gpurun_out/attn_bench_$lib.log 2>&1
  grep -v amdgpu.ids gpurun_out/attn_bench_$lib.log
done
unset RVLM_LIB_PATH
( timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_full.log 2>&1
( GEMM_ABLATE=2 timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_nomfma.log 2>&1
echo "full:    "; grep -v "cube\|amdgpu" gpurun_out/gemm_full.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
echo "no MFMA: "; grep -v "cube\|amdgpu" gpurun_out/gemm_nomfma.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
for f in 1 2; do
( RVLM_ATTN_FUSED=$f timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_fused$f.log 2>&1
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_fused$f.log') if x.startswith('{')][-1]
d=json.loads(l)
pc=d['roofline']['per_class']
print("FUSED=$f", round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), round(d['roofline']['attention_gemm_subset']['frac'],4),
      ' '.join(f"{k}={v['ms']:.2f}" for k,v in sorted(pc.items(), key=lambda kv:-kv[1]['ms'])[:12]))
PY
done
