#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/agpr.log
for r in 1 2; do
for lib in librvlm.so librvlm_agpr.so; do
  echo "== $lib" >> gpurun_out/agpr.log
  RVLM_LIB_PATH=robustvlm_amd/$lib timeout 300 python scripts/sched_probe.py 0 2>&1 | grep -v amdgpu.ids >> gpurun_out/agpr.log
done
done
RVLM_LIB_PATH=robustvlm_amd/librvlm_agpr.so timeout 300 python scripts/gemm_timeline.py 2053 2>&1 | grep -v amdgpu.ids | sed -n 1,9p | cut -c1-200 >> gpurun_out/agpr.log
cat gpurun_out/agpr.log
