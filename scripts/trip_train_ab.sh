#!/bin/bash
# same-box A/B of the training step's weight gradient: round 3's token-chunk transposes + NT GEMM (RVLM_WGRAD_TRANSPOSED=1, EXPERIMENTAL
# build) against the copy-free contraction-major GEMM, alternating; then the rocprofv3 kernel stats of the training step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for i in 1 2 3; do
  for tr in 1 0; do
    echo -n "transposed=$tr "
    RVLM_WGRAD_TRANSPOSED=$tr RVLM_LIB_PATH=$PWD/robustvlm_amd/librvlm_exp.so timeout 300 python bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "images/sec"' | head -1
  done
done
for tr in 1 0; do
  echo -n "transposed=$tr phases "
  RVLM_WGRAD_TRANSPOSED=$tr RVLM_LIB_PATH=$PWD/robustvlm_amd/librvlm_exp.so timeout 300 python scripts/train_phases.py 2>&1 | tail -1
done
} 2>&1 | tee gpurun_out/train_ab.log
