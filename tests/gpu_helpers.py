"""Helpers shared by the -m gpu tests (call the C ABI of librvlm.so through ctypes)."""
import numpy as np
import torch

from robustvlm_amd import _lib as L


def dev():
    return torch.device("cuda:0")


def st():
    return L.stream_ptr()


def lib():
    return L.load()


def cos_sim(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def rel_max(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def pad_rows(t, mult=128):
    """Copy of a 2-D tensor whose storage is readable for round_up(rows, mult) rows."""
    r = (t.shape[0] + mult - 1) // mult * mult
    buf = torch.zeros(r, t.shape[1], dtype=t.dtype, device=t.device)
    buf[: t.shape[0]] = t
    return buf


K_128, K_256, K_PERSISTENT, K_256Q, K_SPLITK, K_STRIP = 1, 2, 4, 8, 16, 32   # include/rvlm_kernels.h RVLM_GEMM_K_*


def persistent_expected(M, N, K):
    """Kernel families rvlm_k_gemm_set_variant(3) must launch for an [M,K] x [N,K]^T problem."""
    if M >= 256 and N % 256 == 0 and K % 128 == 0:
        return K_PERSISTENT | (K_STRIP if M % 256 else 0)
    return None


def record(name, **values):
    """Append measured parity numbers to gpurun_out/parity_metrics.jsonl (evidence for DESIGN.md / profiles/)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_metrics.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **{k: float(v) for k, v in values.items()})) + "\n")
    except OSError:
        pass


def gemm_bf16(A, Bw, epi=0, bias=None, residual=None, h_pre=None, act=0, expect=None):
    """A [M,K] bf16, Bw [N,K] bf16 (cuda).  Returns (out, out_pre).  ``expect``: bit mask of kernel families
    (K_*) that must have run - the dispatcher may not silently route a test to another kernel."""
    l = lib()
    M, K = A.shape
    N = Bw.shape[0]
    Ap = pad_rows(A.contiguous())
    Bc = Bw.contiguous()
    out_dtype = torch.float32 if epi in (1, 4) else torch.bfloat16
    out = torch.zeros(M, N, dtype=out_dtype, device=A.device)
    out_pre = torch.zeros(M, N, dtype=torch.bfloat16, device=A.device) if epi == 2 else None
    L.check(l.rvlm_k_gemm_bf16_nt(Ap.data_ptr(), K, Bc.data_ptr(), K, M, N, K, Ap.shape[0], epi,
                                  L.ptr(bias), out.data_ptr(), N, L.ptr(out_pre), L.ptr(h_pre),
                                  L.ptr(residual), act, st()), "gemm_bf16")
    torch.cuda.synchronize()
    if expect is not None:
        ran = l.rvlm_k_gemm_last_kernels()
        assert ran == expect, \
            f"GEMM [{M},{K}]x[{N},{K}]^T epi {epi}: kernel families {ran:#x} ran, expected {expect:#x}"
    return out, out_pre


def act_ref(h, act):
    if act == 0:
        return h * torch.sigmoid(1.702 * h)
    return torch.nn.functional.gelu(h)


def dact_ref(h, act):
    h = h.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(act_ref(h, act).sum(), h)
    return g


def attn_ref(qkv, B, H, S):
    """fp64 reference of softmax(q k^T / 8) v on packed qkv [B*S, 3W]; returns o [B*S, W]."""
    W = H * 64
    q, k, v = qkv.double().reshape(B, S, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    o = torch.softmax(s, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, W)
