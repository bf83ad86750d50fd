#!/bin/bash
# Round 5, trip A: the new fp32 matrix-pipe tiles, the mixed-precision loop, the N > 1 rehearsal, the fixture-based full-size
# tests - then the whole GPU suite, and the fp32 / mixed / headline bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt; nproc >> gpurun_out/device.txt
( timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm_f32" ) > gpurun_out/t_f32.log 2>&1; tail -3 gpurun_out/t_f32.log
( timeout 900 python -m pytest tests/test_gpu_bench_entry.py -m gpu -q -p no:cacheprovider ) > gpurun_out/t_bench_entry.log 2>&1; tail -5 gpurun_out/t_bench_entry.log
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_bench_entry.py --durations=15 ) > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 2 --warmup 1 --precision fp32 --no-cpu-baseline --no-pmc ) > gpurun_out/bench_fp32.log 2>&1; tail -c 600 gpurun_out/bench_fp32.log
( timeout 600 python bench.py --steps 3 --warmup 1 --precision bf16+fp32-first --no-cpu-baseline --no-pmc ) > gpurun_out/bench_mixed.log 2>&1; tail -c 400 gpurun_out/bench_mixed.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc ) > gpurun_out/bench.log 2>&1; tail -c 300 gpurun_out/bench.log
