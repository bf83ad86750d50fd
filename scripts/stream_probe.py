#!/usr/bin/env python3
"""L2 -> CU operand delivery of the persistent GEMM's request stream as a function of the bytes in flight
(rvlm_k_probe_operand_stream: no MFMA, no LDS; depth x 64 KiB per CU in registers)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
out = torch.zeros(1024, dtype=torch.int32, device=dev)
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(200):
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
for name, M, N, K in (("cube8k", 8192, 8192, 8192), ("qkv", 128 * 256, 3072, 1024), ("fc2", 128 * 256, 1024, 4096))[:int(os.environ.get("PROBE_SHAPES", "3"))]:
    g = torch.Generator(device=dev).manual_seed(0)
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    B = torch.randn(N, K, generator=g, device=dev).bfloat16()
    for depth, mode in ((1, 0), (2, 0), (3, 0), (4, 0), (1, 1), (2, 1), (3, 1), (1, 2), (2, 2), (1, 3), (2, 3),
                        (1, 4), (2, 4), (1, 7), (2, 7), (1, 8), (2, 8), (1, 11), (2, 11), (1, 18), (2, 18), (1, 19), (2, 19), (1, 32), (1, 33), (1, 34), (1, 35)):
        def run(): L.check(lib.rvlm_k_probe_operand_stream(A.data_ptr(), B.data_ptr(), M, N, K, depth, mode, out.data_ptr(), L.stream_ptr()))
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        ksteps = (M // 256) * (N // 256) * (K // 64)            # tile K-steps, 64 KiB of operand requests each
        tb = ksteps * 65536 / (ms * 1e-3) / 1e12
        per_cu_us = ms * 1e3 / (ksteps / 256)
        ring = {32: "GEMM ring: B(g+2),A(g+3); vmcnt(4)", 33: "GEMM ring: A first; vmcnt(4)", 34: "GEMM ring: vmcnt(0)", 35: "GEMM ring: vmcnt(8)"}
        tag = ring[mode] if mode >= 32 else ("regs" if not mode & 2 else "LDS-DMA (MUBUF)" if mode & 16 else "LDS-DMA") + (" + barrier" if mode & 1 else "") + \
              (" + 8-way chunk swizzle" if mode & 4 else " + 4-way chunk swizzle" if mode & 8 else "")
        print(f"{name:7s} {tag:30s} depth {depth} ({depth * 64:3d} KiB/CU in flight): {ms * 1e3:8.1f} us  {tb:6.2f} TB/s requested  "
              f"{per_cu_us * 1e3:7.1f} ns per K-step and CU  (= {2.0 * 256 * 256 * 64 / (per_cu_us * 1e-6) * 256 / 1e12:7.0f} TFLOP/s if MFMA kept up)", flush=True)
