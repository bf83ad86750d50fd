#!/bin/bash
# Attention LDS swizzles (transposing reads conflict-free) and nt epilogue stores: parity tests, micro-benchmark of the two
# library builds (librvlm_swz_r2.so = round-2 swizzles, -DRVLM_ATTN_SWZ_R2), the LDS-conflict PMC pass, whole-call A/B.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/attn_swz_bench.log gpurun_out/attn_swz_pmc.log
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
for lib in librvlm_swz_r2.so librvlm.so; do
  for rep in 1 2; do
  echo "== $lib rep $rep" | tee -a gpurun_out/attn_swz_bench.log
  ( RVLM_LIB_PATH=robustvlm_amd/$lib RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py 257 2>&1 | grep -v amdgpu.ids ) | tee -a gpurun_out/attn_swz_bench.log
  done
done
for lib in librvlm_swz_r2.so librvlm.so; do
  tag=$(basename $lib .so)
  RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib bash scripts/pmc_run.sh attn_$tag "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" -- python $GRAFT_REPO_ROOT/scripts/attn_bench.py 257 | grep -i "attn_" | tee -a gpurun_out/attn_swz_pmc.log
done
AB_VAR=RVLM_LIB_PATH AB_VALS="robustvlm_amd/librvlm_swz_r2.so robustvlm_amd/librvlm.so robustvlm_amd/librvlm_nt.so" AB_REPS=2 SKIP_TESTS=1 bash scripts/trip_ab.sh
