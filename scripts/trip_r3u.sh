#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -3
RVLM_ATTN_PERSIST=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -2
