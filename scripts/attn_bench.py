#!/usr/bin/env python3
"""Micro-benchmark of the bf16 attention kernels: python scripts/attn_bench.py [S=257] [B=128] [H=16]
(ViT-L/14: S=257, H=16; ViT-L/14@336: S=577, H=16; ViT-B/32: S=50, H=12)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 257
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = int(sys.argv[3]) if len(sys.argv) > 3 else 16
W = H * 64; Sp = (S + 31) // 32 * 32
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn(B * S, 3 * W, generator=g, device=dev).bfloat16()
d_o = torch.randn(B * S, W, generator=g, device=dev).bfloat16()
o = torch.zeros(B * S, W, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(B * H * Sp, device=dev); dsum = torch.zeros(max(B * H * Sp, B * H * 16), device=dev)
dqkv = torch.zeros_like(qkv)
def fwd(): L.check(lib.rvlm_k_attn_fwd_bf16(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, S, L.stream_ptr()))
def bwd(): L.check(lib.rvlm_k_attn_bwd_bf16(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dsum.data_ptr(), dqkv.data_ptr(), B, H, S, L.stream_ptr()))
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(300):
    torch.matmul(_wa, _wa)     # clock warm-up (the GPU leaves idle at a low clock)
torch.cuda.synchronize()
# algorithmic HBM bytes: forward reads q, k, v and writes o; backward reads q, k, v, o, dO and writes dq, dk, dv
for name, fn, flops, nbytes in (("fwd", fwd, 4.0 * B * H * S * S * 64, 4.0 * B * H * S * 64 * 2),
                                ("bwd", bwd, 8.0 * B * H * S * S * 64, 8.0 * B * H * S * 64 * 2)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"S={S} B={B} H={H} attn {name}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s (algorithmic)  {nbytes/ms/1e9:6.2f} TB/s (algorithmic bytes)", flush=True)

if os.environ.get("RVLM_ATTN_TRACE") == "2" and S == 257:
    t = dsum.view(torch.int64)[: B * H * 8].view(B * H, 8).cpu().double()
    print("   phase 1 of the fused backward, cycles from its barrier to the end of each wave's own work (mean over heads):")
    print("   " + "  ".join(f"w{w}: {float(t[:, w].mean()):6.0f}" for w in range(8)))
elif os.environ.get("RVLM_ATTN_TRACE") and S == 257:
    t = dsum.view(torch.int64)[: B * H * 8].view(B * H, 8).cpu().double()
    names = ["stage Q,dO,K,V", "fragments + D + odd key", "9 query-tile steps", "dK/dV stores"]
    tot = (t[:, 4] - t[:, 0]).mean()
    for i, nm in enumerate(names):
        print(f"   fused bwd phase {nm:26s}: {float((t[:, i + 1] - t[:, i]).mean()):9.0f} cycles ({100 * float((t[:, i + 1] - t[:, i]).mean() / tot):4.1f} %)")
    print(f"   per workgroup: {float(tot):.0f} cycles")
    for i, nm in ((5, "fragments of K, V, K^T read"), (6, "D, odd-key p / dS"), (7, "odd-key dK / dV (thread 0's wave)"), (2, "barrier")):
        prev = {5: 1, 6: 5, 7: 6, 2: 7}[i]
        print(f"      phase 1: {nm:36s} {float((t[:, i] - t[:, prev]).mean()):8.0f} cycles")
