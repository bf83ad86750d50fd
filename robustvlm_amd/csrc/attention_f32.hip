// fp32 flash attention FORWARD for the fp32-storage engines (precisions fp32 / x3), head_dim 64, products on v_mfma_f32_32x32x2_f32.
//
// Why (round 6): the fp32 attention path materialises the score matrices - one batched product, an in-place softmax pass, a second
// batched product: 2.2 GB of traffic and 21.9 ms per ViT-L/14 forward at B = 128 (scripts/x3_forward_profile.py), a quarter of a
// split-bf16 forward.  The kept probabilities are what that engine's OWN backward reads; a forward whose input gradient runs on the
// bf16 handle (the handoff, engine.hip::vit_backward_from) does not need them (the clean embedding of such an engine takes this
// kernel too - rvlm_vit_set_flash_inference - so that it shares one arithmetic with the first iteration's embedding).
// Here one workgroup owns an (image, head) pair: K and V fp32 in LDS (K rows padded to 68 floats: the A-operand reads are
// ds_read_b128 down the keys), wave w owns query tile w, S^T = K Q^T in the swapped orientation (one query per lane: row statistics
// are per-lane scalars and P^T is already the B operand of O^T = V^T P^T, register for register - the contraction index of a
// 32x32x2 block is {key, key + 4} on the two lane halves for both operands), online softmax in the log2 domain with the scale
// folded into Q.  Keys beyond the last full tile of 32 (the class token's "+1" at S = 257) are folded in by VALU dot products.
// Also written, on request: lse2 = log2 sum_j exp(s_j) per row in the bf16 flash kernels' convention and bf16 copies of q, k, v, o in
// the bf16 engine's layouts - every element of them passes through exactly one workgroup's registers here, so the handoff's export
// passes over qkv and the attention output (3.7 of its 7 ms) disappear.
#include "kernels.h"

namespace rvlm {

constexpr int AF_KLD = 68;   // K tile row stride in floats: lane i reads 16 B at i * 272 + c -> 16-B slot (17 i + c / 16) mod 8: conflict-free

template <int NT>   // NT = ceil(S / 32) waves
__global__ void __launch_bounds__(NT * 64)
attn_fwd_f32_flash_kernel(const float* __restrict__ qkv, long ld, float* __restrict__ o, long ldo, float* __restrict__ lse2, int lse_ld,
                          bf16_t* __restrict__ qkv_bf, long ld_bf, bf16_t* __restrict__ o_bf, long ldo_bf, int H, int S, int W,
                          float scale_log2) {
    constexpr int Sp = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* Ks = (float*)smem_f;                 // [Sp][AF_KLD]
    float* Vs = Ks + Sp * AF_KLD;               // [Sp][64]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const float* base = qkv + (long)b * S * ld + h * 64;
    bf16_t* base_bf = qkv_bf ? qkv_bf + (long)b * S * ld_bf + h * 64 : nullptr;
    // ---- stage K, V (16 B per thread and step; rows >= S are zero: their scores are masked / never evaluated, their V rows must be finite)
    for (int i = tid; i < Sp * 16; i += NT * 64) {
        const int row = i >> 4, c = (i & 15) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (row < S) {
            kv = *(const float4*)(base + (long)row * ld + W + c);
            vv = *(const float4*)(base + (long)row * ld + 2 * W + c);
            if (base_bf) {
                bf16x4 kb = {(bf16_t)kv.x, (bf16_t)kv.y, (bf16_t)kv.z, (bf16_t)kv.w}, vb = {(bf16_t)vv.x, (bf16_t)vv.y, (bf16_t)vv.z, (bf16_t)vv.w};
                *(bf16x4*)(base_bf + (long)row * ld_bf + W + c) = kb;
                *(bf16x4*)(base_bf + (long)row * ld_bf + 2 * W + c) = vb;
            }
        }
        *(float4*)(Ks + row * AF_KLD + c) = kv;
        *(float4*)(Vs + row * 64 + c) = vv;
    }
    // ---- this wave's 32 queries: lane (l31, hi) holds q[32 hi .. 32 hi + 31] of query 32 w + l31, pre-scaled by 0.125 log2(e)
    const int q = w * 32 + l31;
    const bool qvalid = q < S;
    float qf[32];
    {
        const float* qp = base + (long)min(q, S - 1) * ld + 32 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) *(float4*)&qf[4 * j] = *(const float4*)(qp + 4 * j);
        if (base_bf && qvalid) {
            bf16_t* qb = base_bf + (long)q * ld_bf + 32 * hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (bf16_t)qf[8 * j + e];
                *(bf16x8*)(qb + 8 * j) = t;
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) qf[j] *= scale_log2;
    }
    __syncthreads();
    // ---- a last query tile of only 1-4 rows (S = 257: the class token's "+1") does not get an MFMA tile walk - with NT waves on 4
    // SIMDs its wave is the third on SIMD 0, and a whole walk (NK x 64 MFMAs = NK x 4 096 matrix-pipe cycles) for one valid row of 32
    // made that SIMD the kernel's critical path.  Its rows run on the VALU instead: lane <-> key for the scores (K rows by the same
    // conflict-free 16-byte reads), probabilities through S floats of LDS, lane <-> head dim for P V: ~5 k cycles per row.
    const int nrem = S - (NT - 1) * 32;
    if (w == NT - 1 && nrem <= 4) {
        constexpr int NP = (NT + 1) / 2;                    // passes of 64 keys
        float* Ps = Vs + Sp * 64;                           // [Sp] probabilities of the row in flight (this wave only)
        for (int qi = 0; qi < nrem; ++qi) {
            const int qr = (NT - 1) * 32 + qi;
            const float* qp = base + (long)qr * ld;
            float qv[64];
#pragma unroll
            for (int j = 0; j < 16; ++j) *(float4*)&qv[4 * j] = *(const float4*)(qp + 4 * j);
            if (base_bf && lane < 8) {
                const float4 a = *(const float4*)(qp + 8 * lane), c = *(const float4*)(qp + 8 * lane + 4);
                const bf16x8 t = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w, (bf16_t)c.x, (bf16_t)c.y, (bf16_t)c.z, (bf16_t)c.w};
                *(bf16x8*)(base_bf + (long)qr * ld_bf + 8 * lane) = t;
            }
#pragma unroll
            for (int j = 0; j < 64; ++j) qv[j] *= scale_log2;
            float sc[NP], mx = -INFINITY;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int key = p * 64 + lane;
                const float* kr = Ks + min(key, Sp - 1) * AF_KLD;
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const float4 k0 = *(const float4*)(kr + 4 * j), k1 = *(const float4*)(kr + 4 * j + 4);
                    a0 = fmaf(k0.x, qv[4 * j], a0); a0 = fmaf(k0.y, qv[4 * j + 1], a0); a0 = fmaf(k0.z, qv[4 * j + 2], a0); a0 = fmaf(k0.w, qv[4 * j + 3], a0);
                    a1 = fmaf(k1.x, qv[4 * j + 4], a1); a1 = fmaf(k1.y, qv[4 * j + 5], a1); a1 = fmaf(k1.z, qv[4 * j + 6], a1); a1 = fmaf(k1.w, qv[4 * j + 7], a1);
                }
                sc[p] = key < S ? a0 + a1 : -INFINITY;
                mx = fmaxf(mx, sc[p]);
            }
            mx = wave_max(mx);
            float lsum = 0.0f;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float e = __builtin_amdgcn_exp2f(sc[p] - mx);
                lsum += e;
                if (p * 64 + lane < Sp) Ps[p * 64 + lane] = e;
            }
            lsum = wave_sum(lsum);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS writes are ordered in front of its reads)
            float acc0 = 0.0f, acc1 = 0.0f;
            const int S4 = (S + 3) & ~3;                            // (rows S .. Sp - 1 of V are zero, their probabilities too)
            for (int key = 0; key < S4; key += 4) {
                const float4 p4 = *(const float4*)(Ps + key);
                const float* vk = Vs + key * 64 + lane;
                acc0 = fmaf(p4.x, vk[0], acc0); acc1 = fmaf(p4.y, vk[64], acc1);
                acc0 = fmaf(p4.z, vk[128], acc0); acc1 = fmaf(p4.w, vk[192], acc1);
            }
            const float ov = (acc0 + acc1) / lsum;
            o[((long)b * S + qr) * ldo + h * 64 + lane] = ov;
            if (o_bf) o_bf[((long)b * S + qr) * ldo_bf + h * 64 + lane] = (bf16_t)ov;
            if (lse2 && lane == 0) lse2[((long)b * H + h) * lse_ld + qr] = mx + log2f(lsum);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next row's probabilities overwrite Ps
        }
        return;
    }
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;
    const int NK = S >> 5;                       // full key tiles
    for (int kt = 0; kt < NK; ++kt) {
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.0f;
        const float* kr = Ks + (kt * 32 + l31) * AF_KLD + 32 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *(const float4*)(kr + 4 * j);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[4 * j + 0], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[4 * j + 1], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[4 * j + 2], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[4 * j + 3], st, 0, 0, 0);
        }
        // st[r] = score of key 32 kt + (r & 3) + 8 (r >> 2) + 4 hi against query q
        float tmax = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - mnew); psum += st[r]; }
        l = fmaf(l, alpha, psum);
        m = mnew;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        const float* vr = Vs + (kt * 32 + 4 * hi) * 64 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vk = vr + ((r & 3) + 8 * (r >> 2)) * 64;
            oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vk[0], st[r], oacc[0], 0, 0, 0);
            oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vk[32], st[r], oacc[1], 0, 0, 0);
        }
    }
    // ---- the keys beyond the last full tile, one at a time on the VALU (S = 257: the class token's key)
    for (int key = NK * 32; key < S; ++key) {
        const float* kr = Ks + key * AF_KLD + 32 * hi;
        float sc = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *(const float4*)(kr + 4 * j);
            sc = fmaf(a.x, qf[4 * j], sc); sc = fmaf(a.y, qf[4 * j + 1], sc); sc = fmaf(a.z, qf[4 * j + 2], sc); sc = fmaf(a.w, qf[4 * j + 3], sc);
        }
        sc += __shfl_xor(sc, 32, 64);
        const float mnew = fmaxf(m, sc);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        const float p = __builtin_amdgcn_exp2f(sc - mnew);
        l = fmaf(l, alpha, hi == 0 ? p : 0.0f);           // (l is a per-lane partial: the two halves are added at the end)
        m = mnew;
        const float* vk = Vs + key * 64 + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v4 = *(const float4*)(vk + 32 * dt + 8 * g);
                oacc[dt][4 * g + 0] = fmaf(p, v4.x, oacc[dt][4 * g + 0] * alpha);
                oacc[dt][4 * g + 1] = fmaf(p, v4.y, oacc[dt][4 * g + 1] * alpha);
                oacc[dt][4 * g + 2] = fmaf(p, v4.z, oacc[dt][4 * g + 2] * alpha);
                oacc[dt][4 * g + 3] = fmaf(p, v4.w, oacc[dt][4 * g + 3] * alpha);
            }
    }
    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = 1.0f / ltot;
    if (qvalid) {
        // oacc[dt][4 g + e] = O[q][32 dt + 8 g + 4 hi + e]
        float* orow = o + ((long)b * S + q) * ldo + h * 64 + 4 * hi;
        bf16_t* orow_bf = o_bf ? o_bf + ((long)b * S + q) * ldo_bf + h * 64 + 4 * hi : nullptr;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
                *(float4*)(orow + 32 * dt + 8 * g) = v;
                if (orow_bf) {
                    const bf16x4 vb = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
                    *(bf16x4*)(orow_bf + 32 * dt + 8 * g) = vb;
                }
            }
        if (lse2 && hi == 0) lse2[((long)b * H + h) * lse_ld + q] = m + log2f(ltot);
    }
}

// =============================================================================================
// fp32 flash BACKWARD (round 6): the fp32-storage engines' own backward without kept probabilities - P is recomputed from q, k and
// the forward's log-sum-exp rows, like the bf16 flash kernels do.  Two kernels (K, V, Q, dO of a head in fp32 are 4 x 77 KiB: one
// workgroup cannot hold them all):
//   dQ  : K and V in LDS, wave w owns query tile w; per key tile S^T = K Q^T, dP^T = V dO^T (lane <-> query: lse and D are per-lane
//         scalars), dS^T = P^T (dP^T - D) in registers - already the B operand of dQ^T += K^T dS^T.  96 MFMAs per tile pair.  Also
//         writes D = rowsum(dO * O) for the second kernel.
//   dKV : Q and dO in LDS, wave w owns key tile w (K, V rows in registers); per query tile S = Q K^T, dP = dO V^T (lane <-> key,
//         lse / D per accumulator row), P and dS are the B operands of dV^T += dO^T P and dK^T += Q^T dS.  128 MFMAs per tile pair.
// The "+1" token (S = 32 NK + 1..4) never gets an MFMA tile: as a key it is folded into the dQ walk / handled by one wave's VALU pass in
// dKV, as a query the other way round.  Covered: S = 32 NK + r, 1 <= r <= 4, NK <= 8 (S = 257; other lengths keep the batched path).
// =============================================================================================
template <int NT>       // NT = NK + 1 padded tiles; NK waves
__global__ void __launch_bounds__((NT - 1) * 64)
attn_bwd_f32_dq_kernel(const float* __restrict__ qkv, long ld, const float* __restrict__ o, long ldo, const float* __restrict__ d_o, long lddo,
                       const float* __restrict__ lse2, int lse_ld, float* __restrict__ dsum, float* __restrict__ dqkv, long lddq, int H, int S,
                       int W, float scale, float scale_log2) {
    constexpr int Sp = NT * 32, NK = NT - 1;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* Ks = (float*)smem_f;                 // [Sp][AF_KLD]
    float* Vs = Ks + Sp * AF_KLD;               // [Sp][AF_KLD]
    float* Xs = Vs + Sp * AF_KLD;               // VALU path: q row [64] | dO row [64] | dS row [Sp]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const float* base = qkv + (long)b * S * ld + h * 64;
    const float* obase = o + (long)b * S * ldo + h * 64;
    const float* dobase = d_o + (long)b * S * lddo + h * 64;
    float* dqbase = dqkv + (long)b * S * lddq + h * 64;
    const float* lrow = lse2 + ((long)b * H + h) * lse_ld;
    float* drow = dsum + ((long)b * H + h) * lse_ld;
    for (int i = tid; i < Sp * 16; i += NK * 64) {
        const int row = i >> 4, c = (i & 15) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (row < S) {
            kv = *(const float4*)(base + (long)row * ld + W + c);
            vv = *(const float4*)(base + (long)row * ld + 2 * W + c);
        }
        *(float4*)(Ks + row * AF_KLD + c) = kv;
        *(float4*)(Vs + row * AF_KLD + c) = vv;
    }
    __syncthreads();
    // the 1-4 queries beyond the last full tile, by the last wave once its own tile is done (below): lane <-> key for the scores,
    // lane <-> head dim for dQ (~8 k cycles per row)
    auto odd_queries = [&]() {
        constexpr int NP = (NT + 1) / 2;
        for (int qr = NK * 32; qr < S; ++qr) {
            const float dov = dobase[(long)qr * lddo + lane];
            Xs[lane] = base[(long)qr * ld + lane] * scale_log2;
            Xs[64 + lane] = dov;
            const float D = wave_sum(dov * obase[(long)qr * ldo + lane]);
            const float lq = lrow[qr];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
                const int key = p * 64 + lane;
                const float* kr = Ks + min(key, Sp - 1) * AF_KLD;
                const float* vr = Vs + min(key, Sp - 1) * AF_KLD;
                float sc = 0.0f, dp = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float4 k4 = *(const float4*)(kr + 4 * j), v4 = *(const float4*)(vr + 4 * j);
                    const float4 q4 = *(const float4*)(Xs + 4 * j), d4 = *(const float4*)(Xs + 64 + 4 * j);
                    sc = fmaf(k4.x, q4.x, sc); sc = fmaf(k4.y, q4.y, sc); sc = fmaf(k4.z, q4.z, sc); sc = fmaf(k4.w, q4.w, sc);
                    dp = fmaf(v4.x, d4.x, dp); dp = fmaf(v4.y, d4.y, dp); dp = fmaf(v4.z, d4.z, dp); dp = fmaf(v4.w, d4.w, dp);
                }
                if (key < Sp) Xs[128 + key] = key < S ? __builtin_amdgcn_exp2f(sc - lq) * (dp - D) : 0.0f;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float a0 = 0.0f, a1 = 0.0f;
            const int S4 = (S + 3) & ~3;
            for (int key = 0; key < S4; key += 4) {
                const float4 d4 = *(const float4*)(Xs + 128 + key);
                const float* kk = Ks + key * AF_KLD + lane;
                a0 = fmaf(d4.x, kk[0], a0); a1 = fmaf(d4.y, kk[AF_KLD], a1);
                a0 = fmaf(d4.z, kk[2 * AF_KLD], a0); a1 = fmaf(d4.w, kk[3 * AF_KLD], a1);
            }
            dqbase[(long)qr * lddq + lane] = (a0 + a1) * scale;
            if (lane == 0) drow[qr] = D;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    };
    const int q = w * 32 + l31;
    float qf[32], dof[32];
    float D = 0.0f;
    {
        const float* qp = base + (long)q * ld + 32 * hi;
        const float* dp_ = dobase + (long)q * lddo + 32 * hi;
        const float* op = obase + (long)q * ldo + 32 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *(float4*)&qf[4 * j] = *(const float4*)(qp + 4 * j);
            *(float4*)&dof[4 * j] = *(const float4*)(dp_ + 4 * j);
            const float4 o4 = *(const float4*)(op + 4 * j);
            D = fmaf(dof[4 * j], o4.x, D); D = fmaf(dof[4 * j + 1], o4.y, D); D = fmaf(dof[4 * j + 2], o4.z, D); D = fmaf(dof[4 * j + 3], o4.w, D);
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) qf[j] *= scale_log2;
        D += __shfl_xor(D, 32, 64);
        if (hi == 0) drow[q] = D;
    }
    const float lq = lrow[q];
    f32x16 dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.0f;
    for (int kt = 0; kt < NK; ++kt) {
        f32x16 st, dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.0f; dpt[r] = 0.0f; }
        const float* kr = Ks + (kt * 32 + l31) * AF_KLD + 32 * hi;
        const float* vr = Vs + (kt * 32 + l31) * AF_KLD + 32 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *(const float4*)(kr + 4 * j), c = *(const float4*)(vr + 4 * j);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[4 * j + 0], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, dof[4 * j + 0], dpt, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[4 * j + 1], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, dof[4 * j + 1], dpt, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[4 * j + 2], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, dof[4 * j + 2], dpt, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[4 * j + 3], st, 0, 0, 0);
            dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, dof[4 * j + 3], dpt, 0, 0, 0);
            if (j & 1) asm volatile("" ::: "memory");      // (keeps hipcc from requesting all 16 operand quads up front: 64 live registers)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r] - lq) * (dpt[r] - D);      // dS^T[key][q]
        const float* kc = Ks + (kt * 32 + 4 * hi) * AF_KLD + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* kk = kc + ((r & 3) + 8 * (r >> 2)) * AF_KLD;
            dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[0], st[r], dq[0], 0, 0, 0);
            dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[32], st[r], dq[1], 0, 0, 0);
            if ((r & 3) == 3) asm volatile("" ::: "memory");
        }
    }
    for (int key = NK * 32; key < S; ++key) {      // the keys beyond the last full tile, on the VALU
        const float* kr = Ks + key * AF_KLD + 32 * hi;
        const float* vr = Vs + key * AF_KLD + 32 * hi;
        float sc = 0.0f, dp = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *(const float4*)(kr + 4 * j), c = *(const float4*)(vr + 4 * j);
            sc = fmaf(a.x, qf[4 * j], sc); sc = fmaf(a.y, qf[4 * j + 1], sc); sc = fmaf(a.z, qf[4 * j + 2], sc); sc = fmaf(a.w, qf[4 * j + 3], sc);
            dp = fmaf(c.x, dof[4 * j], dp); dp = fmaf(c.y, dof[4 * j + 1], dp); dp = fmaf(c.z, dof[4 * j + 2], dp); dp = fmaf(c.w, dof[4 * j + 3], dp);
        }
        sc += __shfl_xor(sc, 32, 64);
        dp += __shfl_xor(dp, 32, 64);
        const float ds = __builtin_amdgcn_exp2f(sc - lq) * (dp - D);
        const float* kk = Ks + key * AF_KLD + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 k4 = *(const float4*)(kk + 32 * dt + 8 * g);
                dq[dt][4 * g + 0] = fmaf(ds, k4.x, dq[dt][4 * g + 0]); dq[dt][4 * g + 1] = fmaf(ds, k4.y, dq[dt][4 * g + 1]);
                dq[dt][4 * g + 2] = fmaf(ds, k4.z, dq[dt][4 * g + 2]); dq[dt][4 * g + 3] = fmaf(ds, k4.w, dq[dt][4 * g + 3]);
            }
    }
    float* orow = dqbase + (long)q * lddq + 4 * hi;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(orow + 32 * dt + 8 * g) = make_float4(dq[dt][4 * g] * scale, dq[dt][4 * g + 1] * scale, dq[dt][4 * g + 2] * scale, dq[dt][4 * g + 3] * scale);
    if (w == NK - 1) odd_queries();
}

template <int NK>       // NK = S / 32 waves, one key tile each; NT = NK + 1 padded tiles
__global__ void __launch_bounds__(NK * 64)
attn_bwd_f32_dkv_kernel(const float* __restrict__ qkv, long ld, const float* __restrict__ d_o, long lddo, const float* __restrict__ lse2,
                        int lse_ld, const float* __restrict__ dsum, float* __restrict__ dqkv, long lddq, int H, int S, int W, float scale,
                        float scale_log2) {
    constexpr int Sp = (NK + 1) * 32;
    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    float* Qs = (float*)smem_f;                 // [Sp][AF_KLD]
    float* Os = Qs + Sp * AF_KLD;               // dO [Sp][AF_KLD]
    float* Ls = Os + Sp * AF_KLD;               // [Sp] lse rows (+inf beyond S: P = 0)
    float* Dd = Ls + Sp;                        // [Sp] D
    float* Xs = Dd + Sp;                        // VALU path: k row [64] | v row [64] | P row [Sp] | dS row [Sp]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const float* base = qkv + (long)b * S * ld + h * 64;
    const float* dobase = d_o + (long)b * S * lddo + h * 64;
    float* dkbase = dqkv + (long)b * S * lddq + W + h * 64;
    for (int i = tid; i < Sp * 16; i += NK * 64) {
        const int row = i >> 4, c = (i & 15) * 4;
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), dv = qv;
        if (row < S) {
            qv = *(const float4*)(base + (long)row * ld + c);
            dv = *(const float4*)(dobase + (long)row * lddo + c);
        }
        *(float4*)(Qs + row * AF_KLD + c) = qv;
        *(float4*)(Os + row * AF_KLD + c) = dv;
    }
    for (int i = tid; i < Sp; i += NK * 64) {
        Ls[i] = i < S ? lse2[((long)b * H + h) * lse_ld + i] : INFINITY;
        Dd[i] = i < S ? dsum[((long)b * H + h) * lse_ld + i] : 0.0f;
    }
    const int key = w * 32 + l31;
    float kf[32], vf[32];
    {
        const float* kp = base + (long)key * ld + W + 32 * hi;
        const float* vp = base + (long)key * ld + 2 * W + 32 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) { *(float4*)&kf[4 * j] = *(const float4*)(kp + 4 * j); *(float4*)&vf[4 * j] = *(const float4*)(vp + 4 * j); }
#pragma unroll
        for (int j = 0; j < 32; ++j) kf[j] *= scale_log2;
    }
    __syncthreads();
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.0f; dv[dt][r] = 0.0f; }
    for (int qt = 0; qt < NK; ++qt) {
        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.0f; dp[r] = 0.0f; }
        const float* qr = Qs + (qt * 32 + l31) * AF_KLD + 32 * hi;
        const float* dr = Os + (qt * 32 + l31) * AF_KLD + 32 * hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *(const float4*)(qr + 4 * j), c = *(const float4*)(dr + 4 * j);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kf[4 * j + 0], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, vf[4 * j + 0], dp, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kf[4 * j + 1], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, vf[4 * j + 1], dp, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kf[4 * j + 2], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, vf[4 * j + 2], dp, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kf[4 * j + 3], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, vf[4 * j + 3], dp, 0, 0, 0);
            asm volatile("" ::: "memory");                 // (one operand quad pair in flight: the accumulators and K / V rows fill the file)
        }
        // sc[4 g + e] = S[q = 32 qt + 8 g + 4 hi + e][key]: lse / D vary along the accumulator rows
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *(const float4*)(Ls + qt * 32 + 8 * g + 4 * hi), d4 = *(const float4*)(Dd + qt * 32 + 8 * g + 4 * hi);
            const float la[4] = {l4.x, l4.y, l4.z, l4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = __builtin_amdgcn_exp2f(sc[4 * g + e] - la[e]);
                sc[4 * g + e] = p;                                   // P
                dp[4 * g + e] = p * (dp[4 * g + e] - da[e]);         // dS
            }
        }
        const float* oc = Os + (qt * 32 + 4 * hi) * AF_KLD + l31;
        const float* qc = Qs + (qt * 32 + 4 * hi) * AF_KLD + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = ((r & 3) + 8 * (r >> 2)) * AF_KLD;
            dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(oc[ro], sc[r], dv[0], 0, 0, 0);
            dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[ro], dp[r], dk[0], 0, 0, 0);
            dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oc[ro + 32], sc[r], dv[1], 0, 0, 0);
            dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[ro + 32], dp[r], dk[1], 0, 0, 0);
            if (r & 1) asm volatile("" ::: "memory");
        }
    }
    for (int qr = NK * 32; qr < S; ++qr) {      // the queries beyond the last full tile: rank-1 terms on the VALU
        const float* qp = Qs + qr * AF_KLD + 32 * hi;
        const float* dp_ = Os + qr * AF_KLD + 32 * hi;
        float sc = 0.0f, dpv = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 a = *(const float4*)(qp + 4 * j), c = *(const float4*)(dp_ + 4 * j);
            sc = fmaf(a.x, kf[4 * j], sc); sc = fmaf(a.y, kf[4 * j + 1], sc); sc = fmaf(a.z, kf[4 * j + 2], sc); sc = fmaf(a.w, kf[4 * j + 3], sc);
            dpv = fmaf(c.x, vf[4 * j], dpv); dpv = fmaf(c.y, vf[4 * j + 1], dpv); dpv = fmaf(c.z, vf[4 * j + 2], dpv); dpv = fmaf(c.w, vf[4 * j + 3], dpv);
        }
        sc += __shfl_xor(sc, 32, 64);
        dpv += __shfl_xor(dpv, 32, 64);
        const float p = __builtin_amdgcn_exp2f(sc - Ls[qr]);
        const float ds = p * (dpv - Dd[qr]);
        const float* q4p = Qs + qr * AF_KLD + 4 * hi;
        const float* d4p = Os + qr * AF_KLD + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q4 = *(const float4*)(q4p + 32 * dt + 8 * g), o4 = *(const float4*)(d4p + 32 * dt + 8 * g);
                dk[dt][4 * g + 0] = fmaf(ds, q4.x, dk[dt][4 * g + 0]); dk[dt][4 * g + 1] = fmaf(ds, q4.y, dk[dt][4 * g + 1]);
                dk[dt][4 * g + 2] = fmaf(ds, q4.z, dk[dt][4 * g + 2]); dk[dt][4 * g + 3] = fmaf(ds, q4.w, dk[dt][4 * g + 3]);
                dv[dt][4 * g + 0] = fmaf(p, o4.x, dv[dt][4 * g + 0]); dv[dt][4 * g + 1] = fmaf(p, o4.y, dv[dt][4 * g + 1]);
                dv[dt][4 * g + 2] = fmaf(p, o4.z, dv[dt][4 * g + 2]); dv[dt][4 * g + 3] = fmaf(p, o4.w, dv[dt][4 * g + 3]);
            }
    }
    {
        float* krow = dkbase + (long)key * lddq + 4 * hi;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *(float4*)(krow + 32 * dt + 8 * g) = make_float4(dk[dt][4 * g] * scale, dk[dt][4 * g + 1] * scale, dk[dt][4 * g + 2] * scale, dk[dt][4 * g + 3] * scale);
                *(float4*)(krow + W + 32 * dt + 8 * g) = make_float4(dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
            }
    }
    if (w == 0) {       // the keys beyond the last full tile: lane <-> query for P / dS, lane <-> head dim for the sums over the queries
        constexpr int NP = (NK + 2) / 2;
        for (int kr = NK * 32; kr < S; ++kr) {
            Xs[lane] = base[(long)kr * ld + W + lane] * scale_log2;
            Xs[64 + lane] = base[(long)kr * ld + 2 * W + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int p = 0; p < NP; ++p) {
                const int qi = p * 64 + lane;
                const float* qp = Qs + min(qi, Sp - 1) * AF_KLD;
                const float* dp_ = Os + min(qi, Sp - 1) * AF_KLD;
                float sc = 0.0f, dpv = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float4 q4 = *(const float4*)(qp + 4 * j), o4 = *(const float4*)(dp_ + 4 * j);
                    const float4 k4 = *(const float4*)(Xs + 4 * j), v4 = *(const float4*)(Xs + 64 + 4 * j);
                    sc = fmaf(q4.x, k4.x, sc); sc = fmaf(q4.y, k4.y, sc); sc = fmaf(q4.z, k4.z, sc); sc = fmaf(q4.w, k4.w, sc);
                    dpv = fmaf(o4.x, v4.x, dpv); dpv = fmaf(o4.y, v4.y, dpv); dpv = fmaf(o4.z, v4.z, dpv); dpv = fmaf(o4.w, v4.w, dpv);
                }
                if (qi < Sp) {
                    const float pv = __builtin_amdgcn_exp2f(sc - Ls[qi]);          // (rows >= S: lse = +inf -> 0)
                    Xs[128 + qi] = pv;
                    Xs[128 + Sp + qi] = pv * (dpv - Dd[qi]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float ak0 = 0.0f, ak1 = 0.0f, av0 = 0.0f, av1 = 0.0f;
            const int S4 = (S + 3) & ~3;
            for (int qi = 0; qi < S4; qi += 4) {
                const float4 p4 = *(const float4*)(Xs + 128 + qi), d4 = *(const float4*)(Xs + 128 + Sp + qi);
                const float* qq = Qs + qi * AF_KLD + lane;
                const float* oo = Os + qi * AF_KLD + lane;
                av0 = fmaf(p4.x, oo[0], av0); av1 = fmaf(p4.y, oo[AF_KLD], av1); av0 = fmaf(p4.z, oo[2 * AF_KLD], av0); av1 = fmaf(p4.w, oo[3 * AF_KLD], av1);
                ak0 = fmaf(d4.x, qq[0], ak0); ak1 = fmaf(d4.y, qq[AF_KLD], ak1); ak0 = fmaf(d4.z, qq[2 * AF_KLD], ak0); ak1 = fmaf(d4.w, qq[3 * AF_KLD], ak1);
            }
            dkbase[(long)kr * lddq + lane] = (ak0 + ak1) * scale;
            dkbase[(long)kr * lddq + W + lane] = av0 + av1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

bool attn_bwd_f32_flash_covers(int S) {
    const int r = S & 31, NK = S >> 5;
    return r >= 1 && r <= 4 && NK >= 1 && NK <= 8;
}

static unsigned long long g_ab_attr[2][10];

// dqkv [B * S, 3 W] = d(q | k | v) of O = softmax(0.125 Q K^T) V for the cotangent d_o, from qkv, o and the forward's lse2 rows
// ([B * H, lse_ld]); dsum: [B * H, lse_ld] scratch (D = rowsum(dO * O)).  false: sequence length not covered.
bool attn_bwd_f32_flash(const float* qkv, const float* o, const float* d_o, const float* lse2, int lse_ld, float* dsum, float* dqkv, int B,
                        int H, int S, hipStream_t s, int* rc_out) {
    *rc_out = RVLM_OK;
    if (!attn_bwd_f32_flash_covers(S) || ((((size_t)qkv | (size_t)o | (size_t)d_o | (size_t)dqkv)) & 15)) return false;
    const int W = H * 64, NK = S >> 5, NT = NK + 1, Sp = NT * 32;
    const size_t lds_q = (size_t)(2 * Sp * AF_KLD + 128 + Sp) * sizeof(float);
    const size_t lds_kv = (size_t)(2 * Sp * AF_KLD + 2 * Sp + 128 + 2 * Sp) * sizeof(float);
    const float scale = 0.125f, sl2 = 0.125f * 1.4426950408889634f;
#define RVLM_AB_CASE(N)                                                                                                                       \
    case N: {                                                                                                                                 \
        hipError_t err = hipSuccess;                                                                                                          \
        RVLM_ONCE_PER_DEVICE(g_ab_attr[0][N], err = hipFuncSetAttribute((const void*)attn_bwd_f32_dq_kernel<N + 1>,                           \
                                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));             \
        if (err == hipSuccess)                                                                                                                \
            RVLM_ONCE_PER_DEVICE(g_ab_attr[1][N], err = hipFuncSetAttribute((const void*)attn_bwd_f32_dkv_kernel<N>,                          \
                                                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));        \
        if (err != hipSuccess) { *rc_out = fail(RVLM_ERR_HIP, std::string("attn_bwd_f32_flash: ") + hipGetErrorString(err)); return true; }   \
        hipLaunchKernelGGL((attn_bwd_f32_dq_kernel<N + 1>), dim3(B * H), dim3(N * 64), lds_q, s, qkv, 3L * W, o, (long)W, d_o, (long)W, \
                           lse2, lse_ld, dsum, dqkv, 3L * W, H, S, W, scale, sl2);                                                            \
        hipLaunchKernelGGL((attn_bwd_f32_dkv_kernel<N>), dim3(B * H), dim3(N * 64), lds_kv, s, qkv, 3L * W, d_o, (long)W, lse2, lse_ld,      \
                           (const float*)dsum, dqkv, 3L * W, H, S, W, scale, sl2);                                                            \
        break;                                                                                                                                \
    }
    switch (NK) {
        RVLM_AB_CASE(1) RVLM_AB_CASE(2) RVLM_AB_CASE(3) RVLM_AB_CASE(4) RVLM_AB_CASE(5) RVLM_AB_CASE(6) RVLM_AB_CASE(7) RVLM_AB_CASE(8)
        default: return false;
    }
#undef RVLM_AB_CASE
    if (hipGetLastError() != hipSuccess) *rc_out = fail(RVLM_ERR_HIP, "attn_bwd_f32_flash: kernel launch");
    return true;
}

static unsigned long long g_af_attr[17];

// sequence lengths the kernel takes: one wave per query tile (<= 9: K and V of a head must fit the CU's 160 KiB of LDS in fp32)
bool attn_fwd_f32_flash_covers(int S) {
    const int NT = (S + 31) / 32;
    return NT >= 1 && NT <= 9 && (size_t)NT * 32 * (AF_KLD + 64 + 1) * sizeof(float) <= 160 * 1024;
}

// O = softmax(0.125 Q K^T) V, qkv fp32 [B * S, 3 W] (q | k | v, head h at columns 64 h), o fp32 [B * S, W].  Optional: lse2
// [B * H, lse_ld], bf16 copies qkv_bf [B * S, 3 W] / o_bf [B * S, W].  false: shape not covered (the caller runs the batched path).
bool attn_fwd_f32_flash(const float* qkv, float* o, float* lse2, int lse_ld, bf16_t* qkv_bf, bf16_t* o_bf, int B, int H, int S, hipStream_t s,
                        int* rc_out) {
    const int W = H * 64, NT = (S + 31) / 32;
    *rc_out = RVLM_OK;
    if (!attn_fwd_f32_flash_covers(S) || (((size_t)qkv | (size_t)o) & 15)) return false;
    const size_t lds = (size_t)NT * 32 * (AF_KLD + 64 + 1) * sizeof(float);     // K | V | one row of probabilities
    const float sl2 = 0.125f * 1.4426950408889634f;
#define RVLM_AF_CASE(N)                                                                                                             \
    case N: {                                                                                                                       \
        hipError_t err = hipSuccess;                                                                                                \
        RVLM_ONCE_PER_DEVICE(g_af_attr[N], err = hipFuncSetAttribute((const void*)attn_fwd_f32_flash_kernel<N>,                     \
                                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        if (err != hipSuccess) { *rc_out = fail(RVLM_ERR_HIP, std::string("attn_fwd_f32_flash: ") + hipGetErrorString(err)); return true; } \
        hipLaunchKernelGGL((attn_fwd_f32_flash_kernel<N>), dim3(B * H), dim3(N * 64), lds, s, qkv, 3L * W, o, (long)W, lse2, lse_ld, \
                           qkv_bf, 3L * W, o_bf, (long)W, H, S, W, sl2);                                                            \
        break;                                                                                                                      \
    }
    switch (NT) {
        RVLM_AF_CASE(1) RVLM_AF_CASE(2) RVLM_AF_CASE(3) RVLM_AF_CASE(4) RVLM_AF_CASE(5) RVLM_AF_CASE(6) RVLM_AF_CASE(7) RVLM_AF_CASE(8)
        RVLM_AF_CASE(9)
        default: return false;
    }
#undef RVLM_AF_CASE
    if (hipGetLastError() != hipSuccess) *rc_out = fail(RVLM_ERR_HIP, "attn_fwd_f32_flash: kernel launch");
    return true;
}

}  // namespace rvlm
