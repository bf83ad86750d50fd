#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/strip_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/strip_cost2.log
cat gpurun_out/strip_cost2.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2
