#!/bin/bash
# evidence for profiles/: GEMM micro-benchmark (all variants), per-tile timeline, cycle-accurate ablations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== persistent 8-wave kernel (default) + torch.matmul calibration"; GEMM_BENCH_TORCH=1 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu
echo "== persistent 4-wave kernel (RVLM_GEMM_WAVES=4)"; RVLM_GEMM_WAVES=4 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu
echo "== RVLM_GEMM_PERSIST=0 (round-1 default: 4-stage BK=32 256x256 / 128x128 per shape)"; RVLM_GEMM_PERSIST=0 timeout 300 python scripts/gemm_bench.py 2 2>&1 | grep -v amdgpu
} > gpurun_out/r01_gemm_bench.log
timeout 300 python scripts/gemm_trace.py 2>&1 | grep -v amdgpu > gpurun_out/r01_gemm_tile_timeline.log
{ for a in 0 1 4 5 2; do echo "== ablate $a  (bit 0: no operand DMA, bit 1: no MFMA, bit 2: no LDS fragment reads)"; TRACE_ONLY=cube8k,fc1_dgrad GEMM_ABLATE=$a timeout 300 python scripts/gemm_trace.py 2>&1 | grep -E "^[a-z]|kernel|tile 1:"; done; } > gpurun_out/r01_gemm_ablation_cycles.log
tail -5 gpurun_out/r01_gemm_bench.log
