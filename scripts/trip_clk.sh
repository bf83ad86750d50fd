#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in 22 38 46 54; do echo "== ablate $a"; GEMM_ABLATE=$a RVLM_GEMM_PERSIST=1 timeout 300 python scripts/gemm_bench.py 1 2>&1 | grep -E "qkv|fc1_dgrad|cube8k"; done
