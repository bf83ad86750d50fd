#!/usr/bin/env python3
"""K sweep at fixed M,N: separates the per-tile fixed cost from the per-K-step cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
M = 128 * 257; N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
g = torch.Generator(device=dev).manual_seed(0)
for v in (0, 1):
    lib.rvlm_k_gemm_set_variant(v)
    for K in (256, 512, 1024, 2048, 4096):
        mp = (M + 255) // 256 * 256
        A = torch.randn(mp, K, generator=g, device=dev).bfloat16()
        Bw = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        def run():
            L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, mp, 0, None, out.data_ptr(), N, None, None, None, 0, L.stream_ptr()))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"variant={v} N={N} K={K}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TF", flush=True)
lib.rvlm_k_gemm_set_variant(-1)
