#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== old default (variant auto, 4-stage BK32 / 128 kernel)"; RVLM_GEMM_PERSIST=0 RVLM_GEMM_SUPER=0 timeout 300 python scripts/gemm_bench.py 2 2>&1 | grep -v amdgpu
echo "== force 256 k64 non-persistent"; RVLM_GEMM_PERSIST=0 RVLM_GEMM_SUPER=3 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu
echo "== persistent"; GEMM_BENCH_TORCH=1 RVLM_GEMM_PERSIST=1 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu
echo "== bench.py persist=0"; RVLM_GEMM_PERSIST=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
echo "== bench.py persist=1 variant 1"; RVLM_GEMM_VARIANT=1 RVLM_GEMM_PERSIST=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
