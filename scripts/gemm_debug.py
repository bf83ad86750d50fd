#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
from tests.gpu_helpers import gemm_bf16, act_ref, dact_ref, dev, lib
lib().rvlm_k_gemm_set_variant(1)
M, N, K = 641, 512, 256
g = torch.Generator(device="cuda").manual_seed(7)
A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
bias = torch.randn(N, generator=g, device=dev())
acc = A.double() @ Bw.double().t()
h = acc + bias.double()
def report(name, got, ref, tol):
    got = got.double()
    bad = ~torch.isfinite(got) | ((got - ref).abs() > tol * ref.abs().max())
    nb = int(bad.sum())
    print(f"{name}: bad={nb}")
    if nb:
        idx = bad.nonzero()
        rows = idx[:, 0].unique().tolist(); cols = idx[:, 1].unique().tolist()
        print("   rows", rows[:40], "... n=", len(rows)); print("   cols", cols[:40], "... n=", len(cols))
        r, c = idx[0].tolist(); print("   first", r, c, float(got[r, c]), float(ref[r, c]))
for act in (0, 1):
    out0, _ = gemm_bf16(A, Bw, epi=0, bias=bias)
    report("epi0", out0, h, 1e-2)
    out, pre = gemm_bf16(A, Bw, epi=2, bias=bias, act=act)
    report(f"epi2 pre act{act}", pre, h, 1e-2)
    report(f"epi2 out act{act}", out, act_ref(h, act), 1.5e-2)
    hp = torch.randn(M, N, generator=g, device=dev()).bfloat16()
    out, _ = gemm_bf16(A, Bw, epi=3, h_pre=hp, act=act)
    report(f"epi3 act{act}", out, acc * dact_ref(hp.double(), act), 1.5e-2)
