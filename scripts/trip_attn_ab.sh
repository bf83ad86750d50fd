#!/bin/bash
# Same-box A/B of the attention micro-benchmark between library builds and / or environment settings.
#   AB_LIBS="robustvlm_amd/librvlm_base.so robustvlm_amd/librvlm.so" AB_ENVS="X=1 X=2" bash scripts/trip_attn_ab.sh
# Every (lib, env) pair runs AB_REPS (default 2) times, alternating; output: gpurun_out/attn_ab.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/attn_ab.log
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  ( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k attention ) 2>&1 | tail -2 | tee -a gpurun_out/attn_ab.log
fi
for rep in $(seq 1 ${AB_REPS:-2}); do
  for lib in ${AB_LIBS:-robustvlm_amd/librvlm.so}; do
    IFS='|' read -ra envs <<< "${AB_ENVS:-_=_}"
    for e in "${envs[@]}"; do
      echo "== rep $rep lib $(basename $lib) env [$e]" | tee -a gpurun_out/attn_ab.log
      ( export RVLM_LIB_PATH=$GRAFT_REPO_ROOT/$lib; export $e; timeout 300 python scripts/attn_bench.py ${ATTN_ARGS:-} 2>&1 | grep -v amdgpu.ids | tail -${ATTN_TAIL:-2} ) | tee -a gpurun_out/attn_ab.log
    done
  done
done
