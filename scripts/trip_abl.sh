#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 0 1 4 5 2; do echo "== ablate $a"; TRACE_ONLY=qkv,cube8k GEMM_ABLATE=$a timeout 300 python scripts/gemm_trace.py 2>&1 | grep -E "kernel|tile 1:"; done
