"""Kernel-level parity of the HIP building blocks (through include/rvlm_kernels.h)."""
import numpy as np
import pytest
import torch

from robustvlm_amd import _lib as L
from tests.gpu_helpers import (dev, st, lib, cos_sim, rel_max, gemm_bf16, act_ref, dact_ref, attn_ref,
                               persistent_expected, K_128, K_PERSISTENT, K_SPLITK, K_STRIP)

pytestmark = pytest.mark.gpu


def test_library_is_the_in_tree_hip_build():
    l = lib()
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rvlm.h")).read()
    assert l.rvlm_version() == int(re.search(r"#define\s+RVLM_VERSION\s+(\d+)", hdr).group(1))
    assert L.LIB_PATH.endswith("robustvlm_amd/librvlm.so")


@pytest.fixture(params=[0, 3], ids=["gemm128", "persistent"])
def gemm_variant(request):
    """0: the 128x128 kernel only; 3: the persistent 256x256 kernel for every shape it can take (the production
    fill / few-row rules are off), asserted through rvlm_k_gemm_last_kernels."""
    lib().rvlm_k_gemm_set_variant(request.param)
    yield request.param
    lib().rvlm_k_gemm_set_variant(-1)


def family(variant, M, N, K):
    if variant == 0:
        return K_128
    return persistent_expected(M, N, K)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 192, 128), (1028, 3072, 1024), (300, 640, 256),
                                   (64, 64, 64), (514, 1024, 4096), (256, 256, 64), (1285, 768, 3072)])
def test_gemm_bf16_plain(M, N, K, gemm_variant):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
    Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g, device=dev())
    ref = A.double() @ Bw.double().t() + bias.double()
    fam = family(gemm_variant, M, N, K)
    out, _ = gemm_bf16(A, Bw, epi=4, bias=bias, expect=fam)          # fp32 out: only accumulation-order error
    assert rel_max(out, ref) < 2e-5, "transpose / fragment-layout bug"
    outb, _ = gemm_bf16(A, Bw, epi=0, bias=bias, expect=fam)
    assert rel_max(outb.float(), ref) < 1e-2


@pytest.mark.parametrize("act", [0, 1])
def test_gemm_bf16_epilogues(act, gemm_variant):
    M, N, K = 641, 512, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
    Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g, device=dev())
    res = torch.randn(M, N, generator=g, device=dev())
    acc = A.double() @ Bw.double().t()
    fam = family(gemm_variant, M, N, K)
    out, _ = gemm_bf16(A, Bw, epi=1, bias=bias, residual=res, expect=fam)
    assert rel_max(out, acc + bias.double() + res.double()) < 2e-5
    out, pre = gemm_bf16(A, Bw, epi=2, bias=bias, act=act, expect=fam)
    h = acc + bias.double()
    assert rel_max(pre.float(), dact_ref(h, act)) < 1.5e-2          # out_pre = act'(h): all the backward needs of h
    assert rel_max(out.float(), act_ref(h, act)) < 1.5e-2
    hp = torch.randn(M, N, generator=g, device=dev()).bfloat16()     # stands for the stored act'(h)
    out, _ = gemm_bf16(A, Bw, epi=3, h_pre=hp, act=act, expect=fam)
    assert rel_max(out.float(), acc * hp.double()) < 1.5e-2


def test_gemm_f32_strided():
    l = lib()
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 70, 130, 50
    A = torch.randn(M, K, generator=g, device=dev())
    Bt = torch.randn(K, N, generator=g, device=dev())       # (n, k) at k*N + n
    C = torch.zeros(M, N, device=dev())
    L.check(l.rvlm_k_gemm_f32(A.data_ptr(), K, 1, Bt.data_ptr(), 1, N, C.data_ptr(), N, 1, M, N, K, 0.5,
                              None, st()))
    torch.cuda.synchronize()
    assert rel_max(C, 0.5 * (A.double() @ Bt.double())) < 1e-6
    At = A.t().contiguous()                                 # (m, k) at k*M + m
    Bn = Bt.t().contiguous()                                # (n, k) at n*K + k
    L.check(l.rvlm_k_gemm_f32(At.data_ptr(), 1, M, Bn.data_ptr(), K, 1, C.data_ptr(), N, 1, M, N, K, 1.0,
                              None, st()))
    torch.cuda.synchronize()
    assert rel_max(C, A.double() @ Bt.double()) < 1e-6


def _gemm_f32_ex(A, sA, B, sB, M, N, K, nb=1, alpha=1.0, bias=None, act=-1, want_pre=False, dact_h=None, residual=None,
                 scn=1):
    """C[b, m, n] = epi(alpha * sum_k A[b, m, k] B[b, n, k]) through rvlm_k_gemm_f32_ex; sA / sB = (row, k, batch) element
    strides of the buffers as given."""
    l = lib()
    C = torch.full((nb, M, N * scn), float("nan"), device=dev())
    pre = torch.full_like(C, float("nan")) if want_pre else None
    L.check(l.rvlm_k_gemm_f32_ex(A.data_ptr(), sA[0], sA[1], sA[2], B.data_ptr(), sB[0], sB[1], sB[2], C.data_ptr(),
                                 N * scn, scn, M * N * scn, M, N, K, nb, alpha, L.ptr(bias), act, L.ptr(pre), L.ptr(dact_h),
                                 L.ptr(residual), st()))
    torch.cuda.synchronize()
    return C, pre


F32_CASES = [   # M, N, K, nb, A k-contiguous, B k-contiguous
    (70, 130, 50, 1, True, False), (70, 130, 50, 1, False, True), (257, 257, 64, 6, True, True),     # scores = Q K^T
    (257, 64, 257, 6, True, False), (257, 64, 257, 6, False, False),                                  # P V / dS^T Q
    (1028, 3072, 1024, 1, True, True), (514, 1024, 4096, 1, True, False), (1285, 768, 3072, 1, True, True),
    (640, 512, 37, 1, True, True), (33, 1000, 768, 1, True, False), (300, 70, 129, 3, False, True),
    (4224, 3072, 128, 1, True, True), (4230, 3072, 96, 1, False, True),      # > 768 tiles of 128 x 128: the tail-split dispatch
]


@pytest.mark.parametrize("M,N,K,nb,akc,bkc", F32_CASES)
def test_gemm_f32_mfma_bit_identical_to_valu_chain(M, N, K, nb, akc, bkc):
    """The fp32 parity mode runs on v_mfma_f32_32x32x2_f32 since round 5.  The guide states the instruction IS the
    k-ordered fp32 fma chain (one rounding per product, no wider internal sum); this pins it on our kernels: every
    stride form of the encoder's fp32 GEMMs (incl. the unaligned S = 257 attention products, K tails, batches) gives the
    SAME BITS on the matrix-pipe tiles and on the explicit VALU fmaf-chain tiles - and both agree with fp64."""
    l = lib()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K + nb)
    A = torch.randn(nb, M, K, generator=g, device=dev())
    B = torch.randn(nb, N, K, generator=g, device=dev()) * K ** -0.5
    ref = torch.einsum("bmk,bnk->bmn", A.double(), B.double())
    Ab, sA = (A, (K, 1, M * K)) if akc else (A.transpose(1, 2).contiguous(), (1, M, M * K))
    Bb, sB = (B, (K, 1, N * K)) if bkc else (B.transpose(1, 2).contiguous(), (1, N, N * K))
    bias = torch.randn(N, generator=g, device=dev())
    outs = []
    for valu in (0, 1):
        l.rvlm_k_gemm_f32_set_valu(valu)
        try:
            c_plain, _ = _gemm_f32_ex(Ab, sA, Bb, sB, M, N, K, nb, alpha=0.125)
            c_bias, _ = _gemm_f32_ex(Ab, sA, Bb, sB, M, N, K, nb, bias=bias)
            hp = torch.randn(nb, M, N, generator=torch.Generator(device="cuda").manual_seed(5), device=dev())
            res = torch.randn(nb, M, N, generator=torch.Generator(device="cuda").manual_seed(6), device=dev())
            c_act, c_pre = _gemm_f32_ex(Ab, sA, Bb, sB, M, N, K, nb, bias=bias, act=0, want_pre=True)
            c_dact, _ = _gemm_f32_ex(Ab, sA, Bb, sB, M, N, K, nb, dact_h=hp)
            c_res, _ = _gemm_f32_ex(Ab, sA, Bb, sB, M, N, K, nb, bias=bias, residual=res)
        finally:
            l.rvlm_k_gemm_f32_set_valu(0)
        outs.append((c_plain, c_bias, c_act, c_pre, c_dact, c_res))
    for a, b in zip(*outs):
        assert not torch.isnan(a).any()
        assert torch.equal(a, b), float((a - b).abs().max())
    c_plain, c_bias, c_act, c_pre, c_dact, c_res = outs[0]
    tol = 2e-6 * max(1.0, (K / 256) ** 0.5)          # a K-long fp32 chain against fp64 (measured 2.3e-6 at K = 4096)
    assert rel_max(c_plain, 0.125 * ref) < tol
    assert rel_max(c_bias, ref + bias.double()) < tol
    assert rel_max(c_pre, ref + bias.double()) < tol
    assert rel_max(c_act, act_ref(ref + bias.double(), 0)) < 5 * tol
    assert rel_max(c_dact, ref * dact_ref(hp.double(), 0)) < 5 * tol
    assert rel_max(c_res, ref + bias.double() + res.double()) < tol


def test_gemm_f32_column_strided_output():
    """scn != 1 (the strided-output form GemmF32 allows) on the matrix-pipe tiles."""
    g = torch.Generator(device="cuda").manual_seed(11)
    M, N, K = 130, 70, 96
    A = torch.randn(1, M, K, generator=g, device=dev())
    B = torch.randn(1, N, K, generator=g, device=dev())
    C, _ = _gemm_f32_ex(A, (K, 1, M * K), B, (K, 1, N * K), M, N, K, scn=2)
    ref = (A[0].double() @ B[0].double().t())
    assert rel_max(C[0, :, 0::2], ref) < 2e-6
    assert torch.isnan(C[0, :, 1::2]).all()


def test_ds_read_tr16_semantics():
    """Pins the LDS transpose-read the attention kernels rely on: within a 16-lane group, lanes
    4j..4j+3 supply the 8-byte chunks of row j of a 4x16 block and lane i receives column i."""
    l = lib()
    src = torch.arange(2048, dtype=torch.int16, device=dev())            # value = element index
    src_bf = src.view(torch.bfloat16)
    offs = torch.zeros(64, dtype=torch.int32)
    row_stride = 128                                                      # elements
    for lane in range(64):
        grp, i = lane >> 4, lane & 15
        offs[lane] = 2 * ((grp * 4 + (i >> 2)) * row_stride + (i & 3) * 4)
    out = torch.zeros(256, dtype=torch.int16, device=dev())
    L.check(l.rvlm_k_probe_tr16(src_bf.data_ptr(), offs.to(dev()).data_ptr(), out.data_ptr(), st()))
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(64, 4)
    exp = np.zeros((64, 4), dtype=np.int16)
    for lane in range(64):
        grp, i = lane >> 4, lane & 15
        for j in range(4):
            exp[lane, j] = (grp * 4 + j) * row_stride + i
    assert np.array_equal(got, exp), f"ds_read_b64_tr_b16 layout differs:\n{got[:20]}\nexpected\n{exp[:20]}"


@pytest.mark.parametrize("use_tr", [1, 0])
# (20, 16, 257): 320 (image, head) pairs > 256 - some workgroups of the persistent backward walk two heads, the second one
# with its Q / dO tiles prefetched under the first one's tile loop
@pytest.mark.parametrize("B,H,S", [(2, 1, 17), (3, 2, 50), (2, 4, 257), (1, 2, 577), (20, 16, 257)])
def test_attention_bf16_fwd_bwd(B, H, S, use_tr):
    l = lib()
    l.rvlm_k_attn_set_use_tr(use_tr)
    try:
        W = H * 64
        g = torch.Generator(device="cuda").manual_seed(B * 100 + S)
        qkv = torch.randn(B * S, 3 * W, generator=g, device=dev()).bfloat16()
        d_o = torch.randn(B * S, W, generator=g, device=dev()).bfloat16()
        Sp = (S + 31) // 32 * 32
        o = torch.zeros(B * S, W, dtype=torch.bfloat16, device=dev())
        lse = torch.zeros(B * H * Sp, device=dev())
        L.check(l.rvlm_k_attn_fwd_bf16(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, S, st()))
        torch.cuda.synchronize()
        qd = qkv.double().requires_grad_(True)
        ref = attn_ref(qd, B, H, S)
        assert rel_max(o.float(), ref.detach()) < 2e-2
        assert cos_sim(o.float(), ref.detach()) > 0.9999
        (gref,) = torch.autograd.grad((ref * d_o.double()).sum(), qd)
        dqkv = torch.zeros_like(qkv)
        dsum = torch.zeros(B * H * Sp, device=dev())
        L.check(l.rvlm_k_attn_bwd_bf16(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
                                       dsum.data_ptr(), dqkv.data_ptr(), B, H, S, st()))
        torch.cuda.synchronize()
        for name, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
            c = cos_sim(dqkv[:, sl].float(), gref[:, sl])
            r = rel_max(dqkv[:, sl].float(), gref[:, sl])
            assert c > 0.999 and r < 5e-2, f"{name}: cos {c} rel {r}"
    finally:
        l.rvlm_k_attn_set_use_tr(1)


@pytest.mark.parametrize("qi,ki", [(5, 250), (7, 256), (256, 100), (256, 256), (256, 3)])
def test_attention_forced_large_scores(qi, ki):
    """Online-softmax rescale path: one key dominates a query late in the sequence - also for the odd key (token 256,
    folded in by vector code at S = 257) and the odd query (split over the waves and merged through LDS)."""
    l = lib()
    B, H, S = 1, 1, 257
    g = torch.Generator(device="cuda").manual_seed(11 + qi + ki)
    qkv = torch.randn(S, 192, generator=g, device=dev())
    qkv[qi, 0:64] = 6.0 * qkv[ki, 64:128]          # q_qi . k_ki >> others
    qkv = qkv.bfloat16()
    o = torch.zeros(S, 64, dtype=torch.bfloat16, device=dev())
    lse = torch.zeros(288, device=dev())
    L.check(l.rvlm_k_attn_fwd_bf16(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, S, st()))
    torch.cuda.synchronize()
    ref = attn_ref(qkv, B, H, S)
    assert rel_max(o.float(), ref) < 2e-2
    # lse (log2 domain of scale*log2e * q.k) of every row, incl. the odd one
    q, k = qkv[:, 0:64].double(), qkv[:, 64:128].double()
    sc = (q @ k.t()) * 0.125
    lse_ref = torch.logsumexp(sc, dim=1) * 1.4426950408889634
    assert torch.allclose(lse[:S].double().cpu(), lse_ref.cpu(), atol=2e-2)


@pytest.mark.parametrize("M,W", [(37, 64), (50, 768), (257, 1024), (9, 128)])
def test_layernorm_f32(M, W):
    l = lib()
    g = torch.Generator(device="cuda").manual_seed(M + W)
    x = torch.randn(M, W, generator=g, device=dev()) * 3 + 1
    gm = 1 + 0.1 * torch.randn(W, generator=g, device=dev())
    bt = 0.1 * torch.randn(W, generator=g, device=dev())
    dy = torch.randn(M, W, generator=g, device=dev())
    y = torch.zeros_like(x)
    mean = torch.zeros(M, device=dev())
    rstd = torch.zeros(M, device=dev())
    L.check(l.rvlm_k_layernorm_fwd_f32(x.data_ptr(), gm.data_ptr(), bt.data_ptr(), y.data_ptr(),
                                       mean.data_ptr(), rstd.data_ptr(), M, W, st()))
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (W,), gm.double(), bt.double(), 1e-5)
    (gx,) = torch.autograd.grad((ref * dy.double()).sum(), xd)
    dres = torch.ones_like(x)
    L.check(l.rvlm_k_layernorm_bwd_f32(dy.data_ptr(), x.data_ptr(), gm.data_ptr(), mean.data_ptr(),
                                       rstd.data_ptr(), dres.data_ptr(), 1, M, W, st()))
    torch.cuda.synchronize()
    assert rel_max(y, ref.detach()) < 1e-5
    assert rel_max(dres - 1, gx) < 1e-4


@pytest.mark.parametrize("K", [128, 256, 1024])
def test_gemm_bf16_many_tiles_all_epilogues(K):
    """More 256x256 tiles than CUs (the persistent kernel walks >1 tile per workgroup) + a 128-row remainder,
    every epilogue, residual added IN PLACE (out aliases residual, as the weight-gradient accumulation does)."""
    lib().rvlm_k_gemm_set_variant(3)
    try:
        M, N = 256 * 10 + 128, 256 * 30
        g = torch.Generator(device="cuda").manual_seed(K)
        A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
        Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
        bias = torch.randn(N, generator=g, device=dev())
        res = torch.randn(M, N, generator=g, device=dev())
        hp = torch.randn(M, N, generator=g, device=dev()).bfloat16()
        acc = (A.float() @ Bw.float().t()).double()
        fam = K_PERSISTENT | K_STRIP       # 300 tiles of 256x256 on <= 256 workgroups + the 128-row strip phase
        out, _ = gemm_bf16(A, Bw, epi=4, bias=bias, expect=fam)
        assert rel_max(out, acc + bias.double()) < 3e-5
        out, _ = gemm_bf16(A, Bw, epi=4, expect=fam)
        assert rel_max(out, acc) < 3e-5
        out, _ = gemm_bf16(A, Bw, epi=1, bias=bias, residual=res, expect=fam)
        assert rel_max(out, acc + bias.double() + res.double()) < 3e-5
        inplace = res.clone()
        Ap = torch.zeros((M + 255) // 256 * 256, K, dtype=torch.bfloat16, device=dev())
        Ap[:M] = A
        L.check(lib().rvlm_k_gemm_bf16_nt(Ap.data_ptr(), K, Bw.data_ptr(), K, M, N, K, Ap.shape[0], 1, bias.data_ptr(),
                                          inplace.data_ptr(), N, None, None, inplace.data_ptr(), 0, st()), "gemm")
        torch.cuda.synchronize()
        assert lib().rvlm_k_gemm_last_kernels() == fam
        assert rel_max(inplace, acc + bias.double() + res.double()) < 3e-5
        out, _ = gemm_bf16(A, Bw, epi=0, bias=bias, expect=fam)
        assert rel_max(out.float(), acc + bias.double()) < 1e-2
        for act in (0, 1):
            out, pre = gemm_bf16(A, Bw, epi=2, bias=bias, act=act, expect=fam)
            assert rel_max(pre.float(), dact_ref(acc + bias.double(), act)) < 1.5e-2
            assert rel_max(out.float(), act_ref(acc + bias.double(), act)) < 1.5e-2
            out, _ = gemm_bf16(A, Bw, epi=3, h_pre=hp, act=act, expect=fam)
            assert rel_max(out.float(), acc * hp.double()) < 1.5e-2
    finally:
        lib().rvlm_k_gemm_set_variant(-1)


@pytest.mark.parametrize("K", [2048, 3072, 4096])
def test_gemm_bf16_splitk_remainder(K):
    """Few-row problems (M <= 512: the class-token rows of the last block) take split-K slabs on the 128x128 kernel +
    the reduce/epilogue kernel under the production dispatch, for every epilogue."""
    lib().rvlm_k_gemm_set_variant(2)
    try:
        M, N = 256 + 128, 512
        g = torch.Generator(device="cuda").manual_seed(K)
        A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
        Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
        bias = torch.randn(N, generator=g, device=dev())
        res = torch.randn(M, N, generator=g, device=dev())
        hp = torch.randn(M, N, generator=g, device=dev()).bfloat16()
        acc = A.double() @ Bw.double().t()
        fam = K_SPLITK
        out, _ = gemm_bf16(A, Bw, epi=4, bias=bias, expect=fam)
        assert rel_max(out, acc + bias.double()) < 2e-5
        out, _ = gemm_bf16(A, Bw, epi=1, bias=bias, residual=res, expect=fam)
        assert rel_max(out, acc + bias.double() + res.double()) < 2e-5
        out, _ = gemm_bf16(A, Bw, epi=0, bias=bias, expect=fam)
        assert rel_max(out.float(), acc + bias.double()) < 1e-2
        out, pre = gemm_bf16(A, Bw, epi=2, bias=bias, act=0, expect=fam)
        assert rel_max(pre.float(), dact_ref(acc + bias.double(), 0)) < 1.5e-2
        assert rel_max(out.float(), act_ref(acc + bias.double(), 0)) < 1.5e-2
        out, _ = gemm_bf16(A, Bw, epi=3, h_pre=hp, act=0, expect=fam)
        assert rel_max(out.float(), acc * hp.double()) < 1.5e-2
    finally:
        lib().rvlm_k_gemm_set_variant(-1)


@pytest.fixture(params=[0, 1], ids=["copy_free", "transposed"])
def wgrad_form(request):
    """0: the training step's weight-gradient path (column sums + contraction-major persistent GEMM on the token-major operands);
    1: the token-chunk transposes + NT GEMM it replaced (kept as the A/B arm)."""
    lib().rvlm_k_wgrad_set_transposed(request.param)
    yield request.param
    lib().rvlm_k_wgrad_set_transposed(0)


@pytest.mark.parametrize("M,N,K", [(257 * 7, 256, 256), (257 * 24, 512, 256), (257 * 24, 256, 1024), (300, 768, 256),
                                   (257 * 128, 1024, 1024), (257 * 9, 1024, 3072), (50 * 33, 768, 768)])
def test_wgrad_split_vs_fp32(M, N, K, wgrad_form):
    """Split-K weight gradient of the training step (bias gradient + batched persistent GEMM over token chunks + slab
    reduce) against an fp64 matmul of the same bf16 operands; strided operands (column slices of wider buffers, as
    dqkv / the MLP activations are), token counts that are no multiple of the 64-token K-step (the tail chunk's rows
    beyond M must read as zero, whatever follows the operands in memory) and accumulate on top."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    # the operands sit inside larger buffers of NaN: a read of any element that is not theirs poisons the result
    dYw = torch.full((M + 4096, N + 64), float("nan"), device=dev()).bfloat16()
    Xw = torch.full((M + 4096, K + 8), float("nan"), device=dev()).bfloat16()
    dYw[:M, :N] = torch.randn(M, N, generator=g, device=dev()).bfloat16()
    Xw[:M, :K] = torch.randn(M, K, generator=g, device=dev()).bfloat16()
    dY, X = dYw[:M, :N], Xw[:M, :K]
    ref = dY.double().t() @ X.double()
    refb = dY.double().sum(0)
    nb = lib().rvlm_k_wgrad_work_bytes(M, N, K)
    work = torch.empty(nb, dtype=torch.uint8, device=dev())
    dW = torch.full((N, K), 7.0, device=dev())
    db = torch.full((N,), 7.0, device=dev())
    L.check(lib().rvlm_k_wgrad_bf16(dY.data_ptr(), dYw.stride(0), X.data_ptr(), Xw.stride(0), M, N, K, dW.data_ptr(), K,
                                    0, db.data_ptr(), work.data_ptr(), nb, st()), "wgrad")
    torch.cuda.synchronize()
    assert rel_max(dW, ref) < 3e-5
    assert rel_max(db, refb) < 3e-5
    L.check(lib().rvlm_k_wgrad_bf16(dY.data_ptr(), dYw.stride(0), X.data_ptr(), Xw.stride(0), M, N, K, dW.data_ptr(), K,
                                    1, db.data_ptr(), work.data_ptr(), nb, st()), "wgrad")
    torch.cuda.synchronize()
    assert rel_max(dW, 2 * ref) < 3e-5
    assert rel_max(db, 2 * refb) < 3e-5
    # deterministic: same bits on a second run
    dW2 = torch.empty_like(dW)
    dW3 = torch.empty_like(dW)
    for o in (dW2, dW3):
        L.check(lib().rvlm_k_wgrad_bf16(dY.data_ptr(), dYw.stride(0), X.data_ptr(), Xw.stride(0), M, N, K, o.data_ptr(),
                                        K, 0, None, work.data_ptr(), nb, st()), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(dW2, dW3)


def test_wgrad_operand_beyond_2gib():
    """The weight-gradient GEMM addresses its operands with 32-bit byte offsets; an operand whose token range spans more than
    2 GiB (here: a column slice of a [70 000, 16 384] buffer) goes in row chunks that accumulate."""
    M, N, K, ld = 70000, 256, 256, 16384
    g = torch.Generator(device="cuda").manual_seed(5)
    dYw = torch.empty(M, ld, dtype=torch.bfloat16, device=dev())            # 2.3 GB; only the slice is read
    dYw[:, :N] = torch.randn(M, N, generator=g, device=dev()).bfloat16()
    X = torch.randn(M, K, generator=g, device=dev()).bfloat16()
    dY = dYw[:, :N]
    ref = dY.double().t() @ X.double()
    refb = dY.double().sum(0)
    nb = lib().rvlm_k_wgrad_work_bytes(M, N, K)
    work = torch.empty(nb, dtype=torch.uint8, device=dev())
    dW = torch.full((N, K), 7.0, device=dev())
    db = torch.full((N,), 7.0, device=dev())
    L.check(lib().rvlm_k_wgrad_bf16(dY.data_ptr(), ld, X.data_ptr(), K, M, N, K, dW.data_ptr(), K, 0, db.data_ptr(),
                                    work.data_ptr(), nb, st()), "wgrad")
    torch.cuda.synchronize()
    assert rel_max(dW, ref) < 3e-5
    assert rel_max(db, refb) < 3e-5


@pytest.mark.parametrize("r", [32, 100, 128, 255])
@pytest.mark.parametrize("N,K", [(512, 128), (3072, 1024), (1024, 4096)])
def test_gemm_bf16_remainder_rows_in_launch(r, N, K):
    """M = 256 q + r: the r remainder rows are computed by the persistent kernel's strip phase (same launch); every
    epilogue, rows beyond M untouched."""
    lib().rvlm_k_gemm_set_variant(3)
    try:
        M = 256 * 3 + r
        g = torch.Generator(device="cuda").manual_seed(r + N + K)
        A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
        Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
        bias = torch.randn(N, generator=g, device=dev())
        res = torch.randn(M, N, generator=g, device=dev())
        hp = torch.randn(M, N, generator=g, device=dev()).bfloat16()
        acc = (A.float() @ Bw.float().t()).double()
        fam = K_PERSISTENT | K_STRIP
        out, _ = gemm_bf16(A, Bw, epi=4, bias=bias, expect=fam)
        assert rel_max(out, acc + bias.double()) < 3e-5
        out, _ = gemm_bf16(A, Bw, epi=1, bias=bias, residual=res, expect=fam)
        assert rel_max(out, acc + bias.double() + res.double()) < 3e-5
        out, _ = gemm_bf16(A, Bw, epi=0, expect=fam)
        assert rel_max(out.float(), acc) < 1e-2
        out, pre = gemm_bf16(A, Bw, epi=2, bias=bias, act=0, expect=fam)
        assert rel_max(pre.float(), dact_ref(acc + bias.double(), 0)) < 1.5e-2
        assert rel_max(out.float(), act_ref(acc + bias.double(), 0)) < 1.5e-2
        out, _ = gemm_bf16(A, Bw, epi=3, h_pre=hp, expect=fam)
        assert rel_max(out.float(), acc * hp.double()) < 1.5e-2
        # rows beyond M of a taller output buffer stay untouched
        tall = torch.full((M + 64, N), 7.0, device=dev())
        Ap = torch.zeros(256 * 4, K, dtype=torch.bfloat16, device=dev())
        Ap[:M] = A
        L.check(lib().rvlm_k_gemm_bf16_nt(Ap.data_ptr(), K, Bw.data_ptr(), K, M, N, K, Ap.shape[0], 4, None,
                                          tall.data_ptr(), N, None, None, None, 0, st()), "gemm")
        torch.cuda.synchronize()
        assert lib().rvlm_k_gemm_last_kernels() == fam
        assert rel_max(tall[:M], acc) < 3e-5
        assert torch.all(tall[M:] == 7.0)
    finally:
        lib().rvlm_k_gemm_set_variant(-1)


@pytest.mark.parametrize("M,N,K,variant,fam", [(256 * 3 + 100, 512, 256, 3, K_PERSISTENT | K_STRIP),
                                               (128, 256, 1024, 2, K_SPLITK), (300, 192, 64, 2, K_128)])
def test_gemm_bf16_act_epilogue_without_derivative_output(M, N, K, variant, fam, request):
    """Forward-only callers pass out_pre = NULL to the activation epilogue: act(h) is written, act'(h) is not (every
    kernel family: persistent + strip phase, few-row split-K, 128x128 with edge tiles)."""
    lib().rvlm_k_gemm_set_variant(variant)
    request.addfinalizer(lambda: lib().rvlm_k_gemm_set_variant(-1))
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
    Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g, device=dev())
    ref_out, _ = gemm_bf16(A, Bw, epi=2, bias=bias, act=0)
    Ap = torch.zeros((M + 255) // 256 * 256, K, dtype=torch.bfloat16, device=dev())
    Ap[:M] = A
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    L.check(lib().rvlm_k_gemm_bf16_nt(Ap.data_ptr(), K, Bw.data_ptr(), K, M, N, K, Ap.shape[0], 2, bias.data_ptr(),
                                      out.data_ptr(), N, None, None, None, 0, st()), "gemm")
    torch.cuda.synchronize()
    assert lib().rvlm_k_gemm_last_kernels() == fam, hex(lib().rvlm_k_gemm_last_kernels())
    assert torch.equal(out, ref_out)


# the eight GEMM launch types of one ViT-L/14 block at the headline batch (B = 128: M = 128 * 257 = 32 896 rows), as
# (name, N, K, epilogue): forward QKV / out-proj / fc1 / fc2 and their dgrads (robustvlm_amd/csrc/engine.hip)
HEADLINE_GEMMS = [("qkv_fwd", 3072, 1024, 0), ("out_fwd", 1024, 1024, 1), ("fc1_fwd", 4096, 1024, 2),
                  ("fc2_fwd", 1024, 4096, 1), ("fc2_bwd", 4096, 1024, 3), ("fc1_bwd", 1024, 4096, 0),
                  ("qkv_bwd", 1024, 3072, 0), ("out_bwd", 1024, 1024, 0)]


@pytest.mark.parametrize("name,N,K,epi", HEADLINE_GEMMS, ids=[h[0] for h in HEADLINE_GEMMS])
def test_gemm_bf16_headline_shapes_production_dispatch(name, N, K, epi):
    """The exact shapes bench.py times (BASELINE config 2), under the PRODUCTION dispatch (no variant override): the
    persistent 256x256 kernel + its in-launch strip phase must be what runs, and its result must match an fp32-
    accumulated matmul of the same bf16 operands - the dominant kernel of the headline number is compared with
    something other than itself."""
    lib().rvlm_k_gemm_set_variant(-1)
    M = 128 * 257
    g = torch.Generator(device="cuda").manual_seed(N + K + epi)
    A = torch.randn(M, K, generator=g, device=dev()).bfloat16()
    Bw = (torch.randn(N, K, generator=g, device=dev()) * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g, device=dev()) if epi != 3 and "bwd" not in name else None
    acc = (A.float() @ Bw.float().t()).double()
    if bias is not None:
        acc = acc + bias.double()
    fam = K_PERSISTENT | K_STRIP
    if epi == 0:
        out, _ = gemm_bf16(A, Bw, epi=0, bias=bias, expect=fam)
        assert rel_max(out.float(), acc) < 1e-2
    elif epi == 1:
        res = torch.randn(M, N, generator=g, device=dev())
        out, _ = gemm_bf16(A, Bw, epi=1, bias=bias, residual=res, expect=fam)
        assert rel_max(out, acc + res.double()) < 3e-5
    elif epi == 2:
        out, pre = gemm_bf16(A, Bw, epi=2, bias=bias, act=0, expect=fam)
        assert rel_max(pre.float(), dact_ref(acc, 0)) < 1.5e-2
        assert rel_max(out.float(), act_ref(acc, 0)) < 1.5e-2
    else:
        hp = torch.randn(M, N, generator=g, device=dev()).bfloat16()
        out, _ = gemm_bf16(A, Bw, epi=3, h_pre=hp, expect=fam)
        assert rel_max(out.float(), acc * hp.double()) < 1.5e-2
    # the last rows (strip phase) separately: a bug there would be 0.4 % of the elements
    tail = slice(M - 128, M)
    if epi in (0, 2, 3):
        ref_t = acc[tail] if epi == 0 else (act_ref(acc[tail], 0) if epi == 2 else acc[tail] * hp[tail].double())
        assert rel_max(out[tail].float(), ref_t) < 1.5e-2


@pytest.mark.parametrize("cols,ld", [(257, 260), (258, 260), (260, 264), (50, 52), (577, 580)])
def test_softmax_rows_fp32(cols, ld):
    """Row softmax of the fp32 attention path and its backward (round 6: S = 257 rows take a single-pass kernel with 16-byte
    accesses, lane 0 carrying the 1-4 tail columns; other widths the generic three-pass kernel) against torch fp64; pad
    columns cols .. ld - 1 must stay untouched."""
    l = lib()
    rows = 4 * 16 * 3 + 5            # not a multiple of the 16 rows a workgroup takes
    g = torch.Generator(device=dev()).manual_seed(cols)
    s = torch.randn(rows, ld, generator=g, device=dev()) * 3
    s[:, cols:] = 777.0
    ref = torch.softmax(s[:, :cols].double(), dim=-1)
    p = s.clone()
    L.check(l.rvlm_k_softmax_rows(None, p.data_ptr(), rows, cols, ld, 1.0, 0, st()))
    torch.cuda.synchronize()
    assert float((p[:, :cols].double() - ref).abs().max()) < 5e-7, float((p[:, :cols].double() - ref).abs().max())
    assert bool((p[:, cols:] == 777.0).all())
    dp = torch.randn(rows, ld, generator=g, device=dev())
    dp[:, cols:] = 555.0
    want = ref * (dp[:, :cols].double() - (ref * dp[:, :cols].double()).sum(-1, keepdim=True)) * 0.125
    L.check(l.rvlm_k_softmax_rows(p.data_ptr(), dp.data_ptr(), rows, cols, ld, 0.125, 1, st()))
    torch.cuda.synchronize()
    assert float((dp[:, :cols].double() - want).abs().max()) < 5e-7, float((dp[:, :cols].double() - want).abs().max())
    assert bool((dp[:, cols:] == 555.0).all())


@pytest.mark.parametrize("S", [257, 50, 197, 33, 17, 288])
def test_attn_fwd_f32_flash(S):
    """The fp32 flash forward of the fp32-storage engines (round 6, csrc/attention_f32.hip: v_mfma_f32_32x32x2_f32, no score
    matrices) against torch fp64: output, the log-sum-exp rows in the bf16 flash kernels' convention, and the bf16 copies of
    q | k | v and o it writes for the handoff.  Full key tiles run on the matrix pipe, the remainder keys on the VALU: 257 = 8
    tiles + 1, 50 = 1 + 18, 17 = none + 17, 288 = 9 + 0."""
    l = lib()
    B, H = 3, 2
    W = 64 * H
    g = torch.Generator(device=dev()).manual_seed(S)
    qkv = torch.randn(B * S, 3 * W, generator=g, device=dev())
    qkv[:, :W] *= 2.0                                   # logits with a spread of a few units
    o = torch.full((B * S, W), float("nan"), device=dev())
    Sp = (S + 31) // 32 * 32
    lse = torch.full((B * H, Sp), 123.0, device=dev())
    qkv_bf = torch.zeros(B * S, 3 * W, dtype=torch.bfloat16, device=dev())
    o_bf = torch.zeros(B * S, W, dtype=torch.bfloat16, device=dev())
    L.check(l.rvlm_k_attn_fwd_f32_flash(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), qkv_bf.data_ptr(), o_bf.data_ptr(), B, H, S, st()))
    torch.cuda.synchronize()
    q, k, v = (t.reshape(B, S, H, 64).transpose(1, 2).double() for t in qkv.split(W, dim=1))
    sc = q @ k.transpose(-1, -2) * 0.125
    want = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * S, W)
    err = float((o.double() - want).abs().max() / want.abs().max())
    assert err < 2e-6, err
    want_lse = torch.logsumexp(sc, -1) * 1.4426950408889634          # [B, H, S]
    assert float((lse.reshape(B, H, Sp)[:, :, :S].double() - want_lse).abs().max()) < 1e-5
    assert bool((lse.reshape(B, H, Sp)[:, :, S:] == 123.0).all())
    assert torch.equal(qkv_bf, qkv.to(torch.bfloat16)) and torch.equal(o_bf, o.to(torch.bfloat16))
    # without the optional outputs
    o2 = torch.empty_like(o)
    L.check(l.rvlm_k_attn_fwd_f32_flash(qkv.data_ptr(), o2.data_ptr(), None, None, None, B, H, S, st()))
    torch.cuda.synchronize()
    assert torch.equal(o2, o)


@pytest.mark.parametrize("S", [257, 33, 65, 132])
def test_attn_bwd_f32_flash(S):
    """The fp32 flash backward (round 6: dQ kernel + dK / dV kernel on v_mfma_f32_32x32x2_f32, probabilities recomputed from the
    forward's log-sum-exp rows) against torch autograd in fp64, on the flash forward's own outputs.  S = 32 NK + r: the r "+1" tokens
    run on the VALU in both roles (as keys and as queries)."""
    l = lib()
    B, H = 3, 2
    W = 64 * H
    g = torch.Generator(device=dev()).manual_seed(S)
    qkv = torch.randn(B * S, 3 * W, generator=g, device=dev())
    qkv[:, :W] *= 2.0
    d_o = torch.randn(B * S, W, generator=g, device=dev())
    o = torch.empty(B * S, W, device=dev())
    Sp = (S + 31) // 32 * 32
    lse = torch.zeros(B * H, Sp, device=dev())
    L.check(l.rvlm_k_attn_fwd_f32_flash(qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), None, None, B, H, S, st()))
    dsum = torch.zeros(B * H, Sp, device=dev())
    dqkv = torch.full((B * S, 3 * W), float("nan"), device=dev())
    L.check(l.rvlm_k_attn_bwd_f32_flash(qkv.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), dsum.data_ptr(), dqkv.data_ptr(), B, H, S, st()))
    torch.cuda.synchronize()
    x = qkv.double().clone().requires_grad_(True)
    q, k, v = (t.reshape(B, S, H, 64).transpose(1, 2) for t in x.split(W, dim=1))
    out = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B * S, W)
    (want,) = torch.autograd.grad((out * d_o.double()).sum(), x)
    for name, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
        err = float((dqkv[:, sl].double() - want[:, sl]).abs().max() / want[:, sl].abs().max())
        assert err < 5e-6, (name, err)


@pytest.mark.parametrize("W", [1024, 768, 256])
def test_layernorm_bwd_bf16_residual_gradient_stream(W):
    """LayerNorm backward of the bf16 engine, both forms of the residual-gradient stream: fp32 accumulator + bf16 copy (rounds 1-5;
    the handoff's backward and the training step), and the bf16 buffer ALONE (round 6, the attack path: read to accumulate, written
    back - 10 instead of 16 B per element).  Against fp64 torch; the class-token form (accumulate = -S: only every S-th row
    accumulates) in both."""
    l = lib()
    M, S = 4 * 37 + 3, 5
    g = torch.Generator(device="cuda").manual_seed(W)
    x = torch.randn(M, W, generator=g, device=dev()) * 2 + 0.5
    gm = 1 + 0.2 * torch.randn(W, generator=g, device=dev())
    dy = torch.randn(M, W, generator=g, device=dev()).bfloat16()
    acc0 = torch.randn(M, W, generator=g, device=dev()).bfloat16()
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (W,), gm.double(), None, 1e-5)
    (gx,) = torch.autograd.grad((ref * dy.double()).sum(), xd)
    rows = torch.arange(M, device=dev())
    for accumulate in (1, 0, -S):
        keep = torch.ones(M, dtype=torch.bool, device=dev()) if accumulate == 1 else (rows % S == 0) if accumulate < 0 else torch.zeros(M, dtype=torch.bool, device=dev())
        want = gx + acc0.double() * keep[:, None]
        # (a) fp32 stream + bf16 copy
        dres = acc0.float().clone()
        lp = torch.zeros(M, W, dtype=torch.bfloat16, device=dev())
        L.check(l.rvlm_k_layernorm_bwd_bf16(dy.data_ptr(), x.data_ptr(), gm.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dres.data_ptr(),
                                            lp.data_ptr(), accumulate, M, W, st()))
        # (b) the bf16 buffer alone
        lp2 = acc0.clone()
        L.check(l.rvlm_k_layernorm_bwd_bf16(dy.data_ptr(), x.data_ptr(), gm.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None,
                                            lp2.data_ptr(), accumulate, M, W, st()))
        torch.cuda.synchronize()
        assert rel_max(dres, want) < 1e-5, accumulate
        assert torch.equal(lp, dres.bfloat16()), accumulate
        assert torch.equal(lp2, lp), accumulate          # same fp32 arithmetic on the same bf16-representable accumulator: same rounding
