#!/usr/bin/env python3
"""Micro-benchmark of the bf16 MFMA GEMM kernels on the encoder's real shapes (ViT-L/14, B=128)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
M = 128 * 257
shapes = [("qkv", M, 3072, 1024, 0), ("out", M, 1024, 1024, 1), ("fc1", M, 4096, 1024, 2), ("fc2", M, 1024, 4096, 1),
          ("fc2_dgrad", M, 4096, 1024, 3), ("fc1_dgrad", M, 1024, 4096, 0), ("cube4k", 4096, 4096, 4096, 0),
          ("cube8k", 8192, 8192, 8192, 0)]
variants = [int(v) for v in (sys.argv[1:] or ["0", "1"])]
if os.environ.get("GEMM_ABLATE"):
    lib.rvlm_k_gemm_set_ablate(int(os.environ["GEMM_ABLATE"]))
g = torch.Generator(device=dev).manual_seed(0)
# the GPU leaves idle at a low clock and ramps up over many milliseconds: burn ~0.5 s before the first measurement
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(400):
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
for name, m, n, k, epi in shapes:
    mp = (m + 255) // 256 * 256
    pad = int(os.environ.get("GEMM_LDPAD", "0"))     # leading-dimension padding experiment (elements)
    A = torch.randn(mp, k + pad, generator=g, device=dev).bfloat16()
    Bw = (torch.randn(n, k + pad, generator=g, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, generator=g, device=dev)
    res = torch.randn(m, n, generator=g, device=dev) if epi == 1 else None
    hp = torch.randn(m, n, generator=g, device=dev).bfloat16() if epi == 3 else None
    out = torch.empty(m, n, dtype=torch.float32 if epi in (1, 4) else torch.bfloat16, device=dev)
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == 2 else None
    if os.environ.get("GEMM_BENCH_TORCH"):      # calibration only: what the vendor library reaches on this box
        Wt = Bw[:, :k].t()
        for _ in range(3):
            torch.matmul(A[:m, :k], Wt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.matmul(A[:m, :k], Wt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:10s} M={m} N={n} K={k} torch.matmul (no epilogue): {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:8.1f} TFLOP/s", flush=True)
    for v in variants:
        lib.rvlm_k_gemm_set_variant(v)

        def run():
            L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k + pad, Bw.data_ptr(), k + pad, m, n, k, mp, epi, bias.data_ptr(),
                                            out.data_ptr(), n, L.ptr(pre), L.ptr(hp), L.ptr(res), 0, L.stream_ptr()))
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name:10s} M={m} N={n} K={k} epi={epi} variant={v}: {ms*1e3:8.1f} us  {2.0*m*n*k/ms/1e9:8.1f} TFLOP/s", flush=True)
lib.rvlm_k_gemm_set_variant(-1)
