#!/usr/bin/env python3
"""Where the persistent GEMM's K-step goes, per wave: cycles in the operand wait (s_waitcnt vmcnt), in the barrier, in
issuing the 8 DMA pieces, against the tile loop's total (instrumented instantiations: ablate bit 128)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
arms = [int(v) for v in (sys.argv[1:] or ["128", "144", "130"])]
M = 128 * 257
shapes = [("qkv", M, 3072, 1024), ("fc1_dgrad", M, 1024, 4096), ("cube4k", 4096, 4096, 4096), ("cube8k", 8192, 8192, 8192)]
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
    out = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
    for arm in arms:
        lib.rvlm_k_gemm_set_ablate(arm)
        trace = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)

        def run():
            L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k, Bw.data_ptr(), k, m, n, k, mp, 0, bias.data_ptr(), out.data_ptr(), n,
                                            None, None, None, 0, L.stream_ptr()))
        for _ in range(10):
            run()
        lib.rvlm_k_gemm_set_trace(trace.data_ptr())
        run()
        torch.cuda.synchronize()
        lib.rvlm_k_gemm_set_trace(None)
        t = trace.view(256, 8, 4).double()
        tiles = (m // 256) * (n // 256)
        steps = (tiles / 256.0) * (k // 64)
        vm, bar, dma, tot = (t[:, :, i].mean().item() for i in range(4))
        print(f"{name:10s} arm {arm:3d}: per K-step cycles  total {tot/steps:7.0f}  operand wait {vm/steps:6.0f}  barrier {bar/steps:6.0f}"
              f"  DMA issue {dma/steps:6.0f}   (waves 0-3: wait {t[:, :4, 0].mean().item()/steps:5.0f} bar {t[:, :4, 1].mean().item()/steps:5.0f};"
              f" waves 4-7: wait {t[:, 4:, 0].mean().item()/steps:5.0f} bar {t[:, 4:, 1].mean().item()/steps:5.0f})", flush=True)
lib.rvlm_k_gemm_set_ablate(0)
