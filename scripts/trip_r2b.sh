#!/bin/bash
# round-2 trip B: full -m gpu suite, fused-backward attention A/B (RVLM_ATTN_FUSED=1|2), GEMM tile-order / priority knobs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_metrics.jsonl
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for f in 1 2; do
  ( RVLM_ATTN_FUSED=$f timeout 300 python -m pytest tests/test_gpu_kernels.py -q -p no:cacheprovider -k "attention" ) > gpurun_out/attn_test_f$f.log 2>&1
  tail -2 gpurun_out/attn_test_f$f.log
  ( RVLM_ATTN_FUSED=$f RVLM_ATTN_TRACE=1 timeout 300 python scripts/attn_bench.py ) > gpurun_out/attn_bench_f$f.log 2>&1
  ( RVLM_ATTN_FUSED=$f timeout 300 python scripts/attn_bench.py ) >> gpurun_out/attn_bench_f$f.log 2>&1
  cat gpurun_out/attn_bench_f$f.log
done
for gm in 8 4 16 32; do
  ( RVLM_GEMM_GROUP_M=$gm timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_gm$gm.log 2>&1
  echo "GROUP_M=$gm"; grep -v cube gpurun_out/gemm_gm$gm.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
done
( RVLM_GEMM_PRIO=1 timeout 300 python scripts/gemm_bench.py 2 ) > gpurun_out/gemm_prio1.log 2>&1
echo "PRIO=1"; grep -v cube gpurun_out/gemm_prio1.log | awk '{print $1, $(NF-3), $(NF-1)}' | tr '\n' ';'; echo
