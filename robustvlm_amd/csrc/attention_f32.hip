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
