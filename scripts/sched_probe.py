#!/usr/bin/env python3
"""A/B of the persistent GEMM's schedule experiments (rvlm_k_gemm_set_ablate bits 16 / 32 ...) against the production
schedule on the plain-epilogue shapes: bit-exact output check, then interleaved timing rounds (same process, same
buffers, alternating arms so that clock drift hits both)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
arms = [int(v) for v in (sys.argv[1:] or ["0", "16", "48"])]
M = 128 * 257
shapes = [("qkv", M, 3072, 1024), ("fc1_dgrad", M, 1024, 4096), ("wide", 257 * 32, 4096, 1024), ("cube4k", 4096, 4096, 4096),
          ("cube8k", 8192, 8192, 8192), ("k128", 2048, 1024, 128)]
g = torch.Generator(device=dev).manual_seed(0)
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(300):
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
lib.rvlm_k_gemm_set_variant(2)
for name, m, n, k in shapes:
    mp = (m + 255) // 256 * 256
    A = torch.randn(mp, k, generator=g, device=dev).bfloat16()
    Bw = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, generator=g, device=dev)
    outs = {}

    def run(arm, out):
        lib.rvlm_k_gemm_set_ablate(arm)
        L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k, Bw.data_ptr(), k, m, n, k, mp, 0, bias.data_ptr(), out.data_ptr(), n,
                                        None, None, None, 0, L.stream_ptr()))
    for arm in arms:
        outs[arm] = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):       # repeated: a race would show as run-to-run differences
            run(arm, outs[arm])
            torch.cuda.synchronize()
            if arm != arms[0]:
                same = torch.equal(outs[arm], outs[arms[0]])
                if not same:
                    d = (outs[arm].float() - outs[arms[0]].float()).abs()
                    print(f"{name}: arm {arm} DIFFERS from arm {arms[0]}: max {d.max().item():.4g}, "
                          f"{(d > 0).float().mean().item():.3g} of elements", flush=True)
                    break
    times = {arm: [] for arm in arms}
    for rnd in range(4):
        for arm in arms:
            for _ in range(5):
                run(arm, outs[arm])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                run(arm, outs[arm])
            e1.record()
            torch.cuda.synchronize()
            times[arm].append(e0.elapsed_time(e1) / 30 * 1e3)
    line = f"{name:10s} M={m} N={n} K={k}:"
    for arm in arms:
        t = sorted(times[arm])[len(times[arm]) // 2]
        line += f"  arm{arm}: {t:7.1f} us {2.0*m*n*k/t/1e6:7.1f} TF"
    print(line, flush=True)
lib.rvlm_k_gemm_set_ablate(0)
