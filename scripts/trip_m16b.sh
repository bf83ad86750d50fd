#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in ${LIBS}; do
  echo "== rep $rep $lib M16=${M16:-3}"
  RVLM_LIB_PATH=$GRAFT_REPO_ROOT/robustvlm_amd/$lib RVLM_GEMM_M16=${M16:-3} python scripts/gemm_bench.py 1 2>&1 | grep -v amdgpu.ids | grep -E "^(out|fc2|fc2_dgrad|fc1) "
done; done
