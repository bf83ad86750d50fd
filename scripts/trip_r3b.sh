#!/bin/bash
mkdir -p gpurun_out
timeout 300 python scripts/strip_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/strip_cost.log
cat gpurun_out/strip_cost.log
