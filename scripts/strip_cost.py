#!/usr/bin/env python3
"""What the in-launch remainder-row phase costs: each encoder GEMM at M = 32 896 (128 remainder rows) against the same
GEMM at M = 32 768 (none), same process, alternating."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(300):
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
lib.rvlm_k_gemm_set_variant(2)
shapes = [("qkv", 3072, 1024, 0), ("out", 1024, 1024, 1), ("fc1", 4096, 1024, 2), ("fc2", 1024, 4096, 1), ("fc2_dgrad", 4096, 1024, 3),
          ("fc1_dgrad", 1024, 4096, 0)]
for name, n, k, epi in shapes:
    res = {}
    bufs = {}
    for m in (32768, 32896):
        mp = (m + 255) // 256 * 256
        A = torch.randn(mp, k, generator=g, device=dev).bfloat16()
        Bw = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
        bias = torch.randn(n, generator=g, device=dev)
        r = torch.randn(m, n, generator=g, device=dev) if epi == 1 else None
        hp = torch.randn(m, n, generator=g, device=dev).bfloat16() if epi == 3 else None
        out = torch.empty(m, n, dtype=torch.float32 if epi in (1, 4) else torch.bfloat16, device=dev)
        pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == 2 else None
        bufs[m] = (A, Bw, bias, r, hp, out, pre, mp)
        res[m] = []

    def run(m):
        A, Bw, bias, r, hp, out, pre, mp = bufs[m]
        L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k, Bw.data_ptr(), k, m, n, k, mp, epi, bias.data_ptr(), out.data_ptr(), n,
                                        L.ptr(pre), L.ptr(hp), L.ptr(r), 0, L.stream_ptr()))
    for rnd in range(4):
        for m in (32768, 32896):
            for _ in range(5):
                run(m)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run(m)
            e1.record()
            torch.cuda.synchronize()
            res[m].append(e0.elapsed_time(e1) / 30 * 1e3)
    t0, t1 = sorted(res[32768])[2], sorted(res[32896])[2]
    print(f"{name:10s} N={n} K={k} epi={epi}: M=32768 {t0:7.1f} us   M=32896 {t1:7.1f} us   remainder phase {t1 - t0:5.1f} us "
          f"({100 * (t1 - t0) / t1:4.1f} % of the launch; its rows are 0.39 %)", flush=True)
