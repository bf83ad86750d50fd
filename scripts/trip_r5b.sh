#!/bin/bash
# Round 5, trip B: attention A/B on one box - backward tile-loop variants (library builds), forward persistent double-buffered
# kernel (RVLM_ATTN_FWD_PERSIST), each behind the attention parity tests; then fp32 GEMM / engine tests of the padded score path.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
L=gpurun_out/attn_ab.log; rm -f $L
for lib in librvlm.so librvlm_bwdv1.so librvlm_bwdv2.so; do
  echo "== tests lib $lib" | tee -a $L
  ( RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k attention 2>&1 | tail -2 ) | tee -a $L
done
echo "== tests fwd persistent" | tee -a $L
( RVLM_ATTN_FWD_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider -k "attention or l14 or engine_vs" 2>&1 | tail -2 ) | tee -a $L
for rep in 1 2 3; do
  for lib in librvlm.so librvlm_bwdv1.so librvlm_bwdv2.so; do
    for fp in 0 1; do
      echo "== rep $rep lib $lib fwd_persist $fp" | tee -a $L
      ( export RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib RVLM_ATTN_FWD_PERSIST=$fp; timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | tail -2 ) | tee -a $L
    done
  done
done
for lib in librvlm.so librvlm_bwdv1.so librvlm_bwdv2.so; do
  echo "== phase trace lib $lib" | tee -a $L
  ( export RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib RVLM_ATTN_TRACE=1; timeout 300 python scripts/attn_bench.py 2>&1 | grep -v amdgpu.ids | tail -11 ) | tee -a $L
done
( timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -m gpu -q -x -p no:cacheprovider -k "f32 or fp32" 2>&1 | tail -3 ) | tee gpurun_out/t_fp32.log
( timeout 600 python bench.py --steps 2 --warmup 1 --precision fp32 --no-cpu-baseline --no-pmc ) > gpurun_out/bench_fp32.log 2>&1; tail -c 300 gpurun_out/bench_fp32.log
