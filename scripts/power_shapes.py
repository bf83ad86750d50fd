#!/usr/bin/env python3
"""TFLOP/s per watt on the encoder's GEMM shapes: this library's persistent kernel WITH its fused epilogue
against torch.matmul (hipBLASLt, no epilogue), socket power and shader clock sampled from rocm-smi while each loops ~2.5 s."""
import os, sys, subprocess, threading, time, re
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([\d.]+)", o) or re.search(r"Socket Power \(W\):\s*([\d.]+)", o)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            samples.append((float(p.group(1)) if p else -1, int(c.group(1)) if c else -1))
        except Exception:
            pass
        time.sleep(0.05)
def run(name, fn, flops, secs=2.5):
    global stop, samples
    for _ in range(30): fn()
    torch.cuda.synchronize()
    samples = []; stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(30): fn()
        torch.cuda.synchronize(); n += 30
    el = time.time() - t0
    stop = True; th.join()
    ps = [p for p, c in samples if p > 0]; cs = [c for p, c in samples if c > 0]
    tf, w = flops * n / el / 1e12, sum(ps) / max(len(ps), 1)
    print(f"{name:44s} {tf:7.1f} TFLOP/s  {w:6.0f} W  {tf / max(w, 1):6.3f} TFLOP/s per W  sclk {sum(cs)/max(len(cs),1):5.0f} MHz (rocm-smi, n={len(ps)})", flush=True)
Mrows = 128 * 257
shapes = [("qkv fwd   (bias)", Mrows, 3072, 1024, 0), ("out fwd   (bias + fp32 residual)", Mrows, 1024, 1024, 1),
          ("fc1 fwd   (QuickGELU pair)", Mrows, 4096, 1024, 2), ("fc2 fwd   (bias + fp32 residual)", Mrows, 1024, 4096, 1),
          ("fc2 dgrad (x act')", Mrows, 4096, 1024, 3), ("fc1 dgrad (plain)", Mrows, 1024, 4096, 0), ("cube 8192 (plain)", 8192, 8192, 8192, 0)]
g = torch.Generator(device=dev).manual_seed(0)
lib.rvlm_k_gemm_set_variant(1)
for name, m, n, k, epi in shapes:
    mp = (m + 255) // 256 * 256
    A = torch.randn(mp, k, generator=g, device=dev).bfloat16()
    Bw = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, generator=g, device=dev)
    res = torch.randn(m, n, generator=g, device=dev) if epi == 1 else None
    hp = torch.randn(m, n, generator=g, device=dev).bfloat16() if epi == 3 else None
    out = torch.empty(m, n, dtype=torch.float32 if epi == 1 else torch.bfloat16, device=dev)
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == 2 else None
    def ours():
        L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k, Bw.data_ptr(), k, m, n, k, mp, epi, bias.data_ptr(), out.data_ptr(), n,
                                        L.ptr(pre), L.ptr(hp), L.ptr(res), 0, L.stream_ptr()))
    Wt = Bw.t(); Am = A[:m]
    def blas(): torch.matmul(Am, Wt)
    fl = 2.0 * m * n * k
    run(f"{name}: this kernel + epilogue", ours, fl)
    run(f"{name}: hipBLASLt, no epilogue", blas, fl)
    del A, Bw, out, res, hp, pre
lib.rvlm_k_gemm_set_variant(-1)
