#!/usr/bin/env python3
"""Average socket power / shader clock while one GEMM kernel runs back to back (rocm-smi sampled from a thread)."""
import os, sys, subprocess, threading, time, re
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([\d.]+)", o) or re.search(r"Socket Power \(W\):\s*([\d.]+)", o)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            samples.append((float(p.group(1)) if p else -1, int(c.group(1)) if c else -1))
        except Exception as e:
            samples.append((-2, -2))
        time.sleep(0.05)
def run(name, fn, flops, secs=4.0):
    global stop, samples
    for _ in range(50): fn()
    torch.cuda.synchronize()
    samples = []; stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(50): fn()
        torch.cuda.synchronize(); n += 50
    el = time.time() - t0
    stop = True; th.join()
    ps = [p for p, c in samples if p > 0]; cs = [c for p, c in samples if c > 0]
    print(f"{name:28s} {flops * n / el / 1e12:7.1f} TFLOP/s  power avg {sum(ps)/max(len(ps),1):6.1f} W (n={len(ps)})  sclk avg {sum(cs)/max(len(cs),1):6.0f} MHz", flush=True)
M = N = K = 8192
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, generator=g, device=dev).bfloat16()
Bw = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
lib.rvlm_k_gemm_set_variant(1)
def ours(): L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, M, 0, None, out.data_ptr(), N, None, None, None, 0, L.stream_ptr()))
Wt = Bw.t()
def blas(): torch.matmul(A, Wt)
fl = 2.0 * M * N * K
run("persistent 8-wave", ours, fl)
run("hipBLASLt (torch.matmul)", blas, fl)
for abl, nm in ((5, "ours: MFMA only"), (1, "ours: no DMA"), (4, "ours: no LDS reads"), (2, "ours: no MFMA")):
    lib.rvlm_k_gemm_set_ablate(abl)
    run(nm, ours, fl)
lib.rvlm_k_gemm_set_ablate(0)
Z = torch.zeros_like(A); 
def ours0(): L.check(lib.rvlm_k_gemm_bf16_nt(Z.data_ptr(), K, Z.data_ptr(), K, M, N, K, M, 0, None, out.data_ptr(), N, None, None, None, 0, L.stream_ptr()))
run("ours, all-zero operands", ours0, fl)
