#!/usr/bin/env python3
"""One GEMM shape, a few launches (for rocprofv3 --pmc passes): gemm_one.py M N K variant"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
M, N, K, v = (int(a) for a in sys.argv[1:5])
g = torch.Generator(device=dev).manual_seed(0)
mp = (M + 255) // 256 * 256
A = torch.randn(mp, K, generator=g, device=dev).bfloat16()
Bw = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
lib.rvlm_k_gemm_set_variant(v)
for _ in range(5):
    L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, mp, 0, None, out.data_ptr(), N, None, None, None, 0, L.stream_ptr()))
torch.cuda.synchronize()
