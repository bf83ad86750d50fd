#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 6 14 4 12; do echo "== ablate $a"; TRACE_ONLY=cube8k,fc1_dgrad GEMM_ABLATE=$a timeout 300 python scripts/gemm_trace.py 2>&1 | grep -E "kernel|tile 1:"; done
