#!/bin/bash
# epilogue-store cache policy A/B (library builds with -DRVLM_STORE_AUX=n) + attention micro-benchmarks at the other
# sequence lengths the reference accepts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in "257 128 16" "577 32 16" "577 128 16" "50 128 12" "197 128 12"; do
  ( timeout 300 python scripts/attn_bench.py $a 2>&1 | grep -v amdgpu.ids ) | tee -a gpurun_out/attn_shapes.log
done
AB_VAR=RVLM_LIB_PATH AB_VALS="robustvlm_amd/librvlm.so robustvlm_amd/librvlm_aux16.so robustvlm_amd/librvlm_aux17.so robustvlm_amd/librvlm_aux1.so" AB_REPS=2 SKIP_TESTS=1 bash scripts/trip_ab.sh
