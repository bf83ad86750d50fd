#!/usr/bin/env python3
"""Costing study (TEST INFRASTRUCTURE, CPU only): would a SPLIT-bf16 encoder buy the reference's first PGD step?

FARE's first cotangent 2 (phi(x + d0) - phi(x)) is a difference of embeddings ~1e-2 of their norm apart, so a bf16 encoder's
rounding noise (2^-9 per stored activation and per product operand) decides a fifth of the first step's signs
(tests/test_gpu_fullsize.py::test_config2_gradient_signs_along_the_oracle_trajectory_b128).  The engine's remedy so far is one
iteration on the fp32 matrix pipe (1/16 of the bf16 MFMA rate: +84 % per pgd() call).  A split-bf16 product

    a = a_hi + a_lo  (a_hi = bf16(a), a_lo = bf16(a - a_hi)),   a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi    (fp32 accumulate)

carries ~16 mantissa bits at 3 bf16 MFMAs per product.  This script EMULATES encoders built from such products on the CPU (torch
fp32 matmuls of bf16-representable operands are exact products accumulated in fp32, which is what the MFMA does) and measures,
for the first FARE iteration on the first 4 images of the full-size fixtures (ViT-L/14, benign and CLIP-like weights), the
gradient-sign agreement with the fp32 oracle (= the reference's own arithmetic, train/pgd_train.py:30-38):

    bf16          every linear / attention product in bf16 operands, activations STORED in bf16 (the engine's bf16 mode, emulated)
    bf16-f32act   bf16 operands, fp32 stored activations (what only changing the storage would buy)
    x2            activations split (2 products), weights single bf16
    w2            WEIGHTS split (2 products: a_hi w_hi + a_hi w_lo), activations single bf16 and stored in bf16, attention products
                  in bf16: the bf16 engine as it is with K doubled - [A | A] x [W_hi | W_lo]
    x3            both split, 3 products (the proposal), fp32 stored activations; attention products x3 as well
    x3-linear     x3 in the linears only, attention products in fp32 (what gemm_f32's batched tiles would keep doing)

Usage: python oracle/split_bf16_emulation.py [benign|clip|both]   (8 threads: ~2 minutes per weight set)
"""
from __future__ import annotations

import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_ref as V  # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def split(t):
    hi = bf(t)
    return hi, bf(t - hi)


def prod(a, b, mode):
    """a @ b with both operands [.., m, k] x [.., k, n] under the product mode (forward value only)."""
    if mode == "f32":
        return a @ b
    if mode == "bf16":
        return bf(a) @ bf(b)
    if mode == "x2":                      # a split, b single
        ah, al = split(a)
        bh = bf(b)
        return ah @ bh + al @ bh
    if mode == "w2":                      # WEIGHT operand split, the other single: forward / dgrad products have the weight as b
        ah = bf(a)
        bh, bl = split(b)
        return ah @ bh + ah @ bl
    ah, al = split(a)
    bh, bl = split(b)
    return ah @ bh + (ah @ bl + al @ bh)


class Mat(torch.autograd.Function):
    """y = a @ b; the two backward products use the same product mode (dA = dY b^T, dB = a^T dY)."""

    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.save_for_backward(a, b)
        mode, _, bmode = mode.partition("/")      # "x3/bf16": forward products x3, the two backward products bf16
        ctx.mode = bmode or mode
        return prod(a, b, mode)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = prod(g, b.transpose(-1, -2), ctx.mode) if ctx.needs_input_grad[0] else None
        gb = prod(a.transpose(-1, -2), g, ctx.mode) if ctx.needs_input_grad[1] else None
        return ga, gb, None


class Store(torch.autograd.Function):
    """An activation written to memory in bf16 (forward) and its gradient likewise (backward)."""

    @staticmethod
    def forward(ctx, t):
        return bf(t)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class StoreGrad(torch.autograd.Function):
    """fp32 activation in the forward, its GRADIENT written in bf16 (x3 forward followed by the bf16 engine's backward)."""

    @staticmethod
    def forward(ctx, t):
        return t.clone()

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class AttnHandoff(torch.autograd.Function):
    """Attention core of the handoff: forward = the precise path's (products in `att`, fp32 softmax), backward = the bf16 flash
    kernel's arithmetic on bf16-ROUNDED q, k, v, O, dO with the probabilities recomputed from the rounded q, k against the
    PRECISE forward's log-sum-exp (so its rows do not sum to one exactly), D = rowsum(dO * O)."""

    @staticmethod
    def forward(ctx, q, k, v, att, scale):
        s = prod(q, k.transpose(-1, -2), att) * scale
        lse = torch.logsumexp(s, dim=-1, keepdim=True)
        o = prod(torch.exp(s - lse), v, att)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        qb, kb, vb, ob, dob = bf(q), bf(k), bf(v), bf(o), bf(do)
        p = torch.exp((qb @ kb.transpose(-1, -2)) * ctx.scale - lse)
        dv = bf(p).transpose(-1, -2) @ dob
        dp = dob @ vb.transpose(-1, -2)
        d = (dob * ob).sum(-1, keepdim=True)
        ds = bf(p * (dp - d) * ctx.scale)
        return bf(ds @ kb), bf(ds.transpose(-1, -2) @ qb), bf(dv), None, None


def forward(cfg, w, x, lin, att, store_bf16):
    """oracle/vit_ref.py::vit_forward with every GEMM routed through Mat (linears: `lin`, attention products: `att`)."""
    W, H = cfg.width, cfg.heads
    dh = W // H
    res_bf16 = isinstance(store_bf16, str) and store_bf16.endswith("+res")      # the residual GRADIENT stream carried in bf16 too
    res_fwd = isinstance(store_bf16, str) and store_bf16.endswith("+resfwd")    # ... and the residual stream ITSELF (forward) as well
    if res_bf16:
        store_bf16 = store_bf16[:-4]
    if res_fwd:
        store_bf16 = store_bf16[:-7]
    if store_bf16 == "all":
        store_bf16 = True
    rs = Store.apply if res_fwd else StoreGrad.apply if res_bf16 else (lambda t: t)
    st = StoreGrad.apply if store_bf16 == "grad" else Store.apply if store_bf16 else (lambda t: t)
    act = V.quick_gelu
    B = x.shape[0]
    g = cfg.grid
    patches = x.reshape(B, 3, g, cfg.patch, g, cfg.patch).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, -1)
    t = Mat.apply(st(patches), w["conv1.weight"].reshape(W, -1).t(), lin)
    t = torch.cat([w["class_embedding"].expand(B, 1, W), t], dim=1) + w["positional_embedding"]
    t = F.layer_norm(t, (W,), w["ln_pre.weight"], w["ln_pre.bias"], 1e-5)
    N = t.shape[1]
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        h = st(F.layer_norm(t, (W,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], 1e-5))
        qkv = st(Mat.apply(h, w[p + "attn.in_proj_weight"].t(), lin) + w[p + "attn.in_proj_bias"])
        q, k, v = (z.reshape(B, N, H, dh).transpose(1, 2) for z in qkv.split(W, dim=-1))
        if att.endswith("/flash"):
            a = st(AttnHandoff.apply(q, k, v, att.partition("/")[0], 1.0 / math.sqrt(dh)).transpose(1, 2).reshape(B, N, W))
        else:
            s = Mat.apply(q, k.transpose(-1, -2), att) * (1.0 / math.sqrt(dh))
            pr = torch.softmax(s, dim=-1)
            a = st(Mat.apply(st(pr) if att == "bf16" or store_bf16 == "grad" else pr, v, att).transpose(1, 2).reshape(B, N, W))
        t = rs(t + Mat.apply(a, w[p + "attn.out_proj.weight"].t(), lin) + w[p + "attn.out_proj.bias"])
        h = st(F.layer_norm(t, (W,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], 1e-5))
        h = st(act(Mat.apply(h, w[p + "mlp.c_fc.weight"].t(), lin) + w[p + "mlp.c_fc.bias"]))
        t = rs(t + Mat.apply(h, w[p + "mlp.c_proj.weight"].t(), lin) + w[p + "mlp.c_proj.bias"])
    pooled = F.layer_norm(t[:, 0], (W,), w["ln_post.weight"], w["ln_post.bias"], 1e-5)
    return pooled @ w["proj"]


MODES = {
    "f32": ("f32", "f32", False),
    "bf16": ("bf16", "bf16", True),
    "bf16-f32act": ("bf16", "bf16", False),
    "x2": ("x2", "x2", False),
    "w2": ("w2", "bf16", True),
    "x3": ("x3", "x3", False),
    "x3-linear": ("x3", "f32", False),
    # round 6, second costing: is the noise in the FORWARD difference only?  x3 forward (fp32 stored activations), then the bf16
    # engine's backward: bf16 operands (saved activations rounded), gradients stored in bf16
    "x3fwd-bf16bwd": ("x3/bf16", "f32/bf16", "grad"),
    "f32fwd-bf16bwd": ("f32/bf16", "f32/bf16", "grad"),
    "x3fwd-bf16bwd-flash": ("x3/bf16", "f32/flash", "grad"),   # ... with the backward's attention core as the flash kernel runs it
    # would a bf16-only residual-GRADIENT stream (LayerNorm backward 16 -> 10 B per element, ~15 ms per pgd() call) cost signs?
    "f32fwd-bf16bwd-bf16res": ("f32/bf16", "f32/flash", "grad+res"),
    # LATER iterations (run with ITER=k: the perturbation after k sign steps of the fp32 arm): what the bf16 engine's fp32 residual
    # stream is worth there - bf16 as it is, with the bf16 residual-gradient stream (shipped), and with a bf16 residual stream forward
    "bf16-bf16res": ("bf16", "bf16", "all+res"),
    "bf16-bf16resfwd": ("bf16", "bf16", "all+resfwd"),
    "x3lin-bf16att-bf16bwd": ("x3/bf16", "bf16", "grad"),      # ... and the forward's attention products on bf16 operands too
    "x3lin-x2att-bf16bwd": ("x3/bf16", "x2/bf16", "grad"),
}


def first_iteration(cfg, w, x, d0, name):
    lin, att, store = MODES[name]
    with torch.no_grad():
        e0 = forward(cfg, w, V.normalize_pixels(x), lin, att, store)
    d = d0.clone().requires_grad_(True)
    e = forward(cfg, w, V.normalize_pixels(x + d), lin, att, store)
    per = ((e - e0) ** 2).sum(1)
    (gr,) = torch.autograd.grad(per.mean(), d)
    return e0, per.detach(), gr


def run(tag, clip_like, n=4):
    cfg = V.VIT_L_14
    w = V.init_weights(cfg, seed=3, clip_like=clip_like)
    eps = 4 / 255
    x = torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(0))[:n]
    d0 = ((torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1) * eps)[:n]
    for it in range(int(os.environ.get("ITER", "0"))):     # k plain sign steps of the fp32 arm (momentum-free PGD: a pricing study)
        _, _, g = first_iteration(cfg, w, x, d0, "f32")
        d0 = (d0 + (1 / 255) * torch.sign(g)).clamp(-eps, eps)
        d0 = (x + d0).clamp(0, 1) - x
        tag = tag.split(" @")[0] + f" @ iteration {it + 1}"
    ref = None
    for name in (os.environ.get("ARMS", "").split(",") if os.environ.get("ARMS") else MODES):
        t0 = time.time()
        e0, per, g = first_iteration(cfg, w, x, d0, name)
        if ref is None:
            ref = (e0, per, g)
            # the emulation's fp32 arm against the oracle proper (same arithmetic, conv as an explicit patch matmul)
            eo = V.vit_forward(cfg, w, V.normalize_pixels(x))
            print(f"[{tag}] f32 arm vs oracle vit_forward: emb rel {float((e0 - eo).abs().max() / eo.abs().max()):.1e}", flush=True)
            continue
        sign = float((torch.sign(g) == torch.sign(ref[2])).float().mean())
        cos = float((g.double() * ref[2].double()).sum() / (g.double().norm() * ref[2].double().norm()))
        emb = float((e0 - ref[0]).abs().max() / ref[0].abs().max())
        print(f"[{tag}] {name:12s} sign_agree_it0 {sign:.4f}  grad cos {cos:.6f}  loss ratio {float((per / ref[1]).mean()):.4f}  "
              f"emb rel err {emb:.1e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("THREADS", "8")))
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("benign", "both"):
        run("benign weights", False)
    if which in ("clip", "both"):
        run("CLIP-like weights", True)
