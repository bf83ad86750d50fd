// bf16 flash attention forward / backward for CLIP ViTs on gfx950 (head_dim = 64, no mask, no
// dropout, S = 50 / 257 / 577 tokens), v_mfma_f32_32x32x16_bf16 throughout.
//
// One workgroup per (image, head).  A whole head's K and V (S x 64 bf16 = 32 KiB each at S = 257)
// are LDS-resident, so there is no K/V streaming loop over HBM: every wave owns 32-row tiles and
// walks the LDS-resident operand in 32-row tiles with an online (running max / sum) softmax.
//
// Layout tricks
//  * LDS tiles are [rows][64] bf16 (128-B rows), 16-B chunk index XOR swz_key(row) (a permutation of row bits 1-3):
//    ds_read_b128 fragment reads AND the ds_read_b64_tr_b16 transposing reads are bank-conflict free.
//  * "Swapped" products: S^T = K.Q^T puts one query per lane (lane&31), so softmax row statistics are
//    per-lane scalars and P^T / dS^T are already in the B-operand register layout of the next MFMA
//    (the k index is permuted identically on both operands, so no cross-lane exchange is needed).
//  * The operand that needs a transpose (V^T in fwd, K^T / Q^T / dO^T in bwd) comes from the same
//    row-major LDS tile through ds_read_b64_tr_b16 (hardware transpose read, 4 rows x 16 cols per
//    16-lane group).  USE_TR=false builds the same fragment with scalar LDS reads (validation).
//
// Forward also stores lse2[b,h,q] = m + log2(sum) in the log2 domain of (scale*log2e)*q.k for the
// backward.  Backward = prep (D = rowsum(dO*O)) + dQ kernel (K,V in LDS) + dK/dV kernel (Q,dO in LDS)
// = 7 tile GEMMs instead of the minimal 5; the attention core is ~8 % of the backward FLOPs.
#include "kernels.h"

namespace rvlm {
// element strides of the (image, head) blocks: token-major [B][S][3W] (b = S*ld, h = 64) or head-blocked [3][B*H][S][64]
struct AttnLayout { long qkv_b, qkv_h, o_b, o_h; };


typedef __attribute__((address_space(3))) char lds_char;

// Swizzle key of a tile row: row bits (1, 3, 2) -> chunk bits (2, 1, 0).  Any bijection of those three row bits keeps the
// row-major ds_read_b128 fragments conflict-free (the 8 same-parity rows of a 16-lane group get 8 different keys); the
// TRANSPOSING reads (ds_read_b64_tr_b16: a 32-lane half reads 4 consecutive rows x one 64-B half row) additionally need
// rows r and r + 2 - same bank parity - to land in DIFFERENT 64-B halves, i.e. row bit 1 on chunk bit 2.  Round 2 used
// (row >> 1) & 7: every transposing read was a 2-way bank conflict (scripts/lds_bank_model.py: 16 of 40 extra LDS cycles
// per query tile and wave; PMC: 32.5 % of the backward's LDS-active cycles in conflicts, profiles/r02_pmc_attention.json).
__device__ __forceinline__ int swz_key(int row) {
    return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1);
}
constexpr int DS_KEY_SHIFT = 1;
__device__ __forceinline__ int swz_off(int row, int chunk) {  // byte offset inside a [rows][64] bf16 tile
    return row * 128 + ((chunk ^ swz_key(row)) << 4);
}

// cooperative stage of rows [0,Sp) x 64 bf16 from global (row stride ld elements) into a swizzled tile
// by LDS-DMA (global_load_lds_dwordx4: 8 rows = 1 KiB per wave instruction, asynchronous, no VGPR
// round trip).  The DMA writes LDS lane-linearly, so the swizzle goes on the per-lane source address.
// Rows >= S replicate row S-1 (finite data); every consumer masks them (scores -> -inf / p = 0).
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* src, long ld, int S, int Sp,
                                           int w, int nw, int lane) {
    for (int blk = w; blk < (Sp >> 3); blk += nw) {
        const int row = blk * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ swz_key(row);
        const int rs = min(row, S - 1);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (long)rs * ld + lc * 8),
            (__attribute__((address_space(3))) void*)(tile + blk * 1024), 16, 0, 0);
    }
}

// A/B operand fragment with k along the 64-wide (contiguous) dim: row = row0 + lane&31, k-chunk kk
__device__ __forceinline__ bf16x8 frag_rowmajor(const char* tile, int row0, int kk, int lane) {
    const int row = row0 + (lane & 31);
    return *(const bf16x8*)(tile + swz_off(row, kk * 2 + (lane >> 5)));
}
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* src, long ld, int row, int kk, int lane) {
    return *(const bf16x8*)(src + (long)row * ld + (kk * 2 + (lane >> 5)) * 8);
}

// Transposed fragment X^T: MFMA row index = column (dt*32 + lane&31) of the LDS tile, k index t'
// (0..7) <-> tile row  rowbase + 4*hi + (t'&3) + 8*(t'>>2)   (the permutation pack_b() uses).
template <bool USE_TR>
__device__ __forceinline__ bf16x8 frag_transposed(const char* tile, int rowbase, int dt, int lane) {
    const int hi = lane >> 5;
    bf16x8 out;
    if (USE_TR) {
        const int i = lane & 15, dblk = (lane >> 4) & 1;
        const int chunk = dt * 4 + dblk * 2 + ((i & 3) >> 1);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = rowbase + 4 * hi + 8 * r + (i >> 2);
            const int off = swz_off(row, chunk) + (i & 1) * 8;
            bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)((lds_char*)tile + off));
            out[4 * r + 0] = v[0]; out[4 * r + 1] = v[1]; out[4 * r + 2] = v[2]; out[4 * r + 3] = v[3];
        }
    } else {
        const int col = dt * 32 + (lane & 31);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int row = rowbase + 4 * hi + (t & 3) + 8 * (t >> 2);
            out[t] = *(const bf16_t*)(tile + swz_off(row, col >> 3) + (col & 7) * 2);
        }
    }
    return out;
}

// Per-lane byte offsets that do not depend on the tile index (32-row tiles start at multiples of 16
// rows, which leaves the swz_key(row) swizzle term unchanged): address = tile + row0*128 + offset.
struct FragOffs {
    int rm[4];      // row-major fragment, k-chunk kk:      row0 = first row of the 32-row tile
    int tr[2][2];   // transposed fragment [dt][r]:          row0 = first row of the 16-row k-slice
};
__device__ __forceinline__ FragOffs make_offs(int lane) {
    FragOffs o;
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o.rm[kk] = swz_off(l31, kk * 2 + hi);
    const int i = lane & 15, dblk = (lane >> 4) & 1;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int x = 4 * hi + 8 * r + (i >> 2);
            o.tr[dt][r] = swz_off(x, dt * 4 + dblk * 2 + ((i & 3) >> 1)) + (i & 1) * 8;
        }
    return o;
}
__device__ __forceinline__ bf16x8 frag_rm(const char* tile, int row0, int off) {
    return *(const bf16x8*)(tile + row0 * 128 + off);
}
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int row0, const FragOffs& o, int dt) {
    bf16x8 out;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4*)((lds_char*)tile + row0 * 128 + o.tr[dt][r]));
        out[4 * r + 0] = v[0]; out[4 * r + 1] = v[1]; out[4 * r + 2] = v[2]; out[4 * r + 3] = v[3];
    }
    return out;
}
template <bool USE_TR>
__device__ __forceinline__ bf16x8 frag_t(const char* tile, int row0, const FragOffs& o, int dt, int lane) {
    if (USE_TR) return frag_tr(tile, row0, o, dt);
    return frag_transposed<false>(tile, row0, dt, lane);
}

// accumulator (D layout: row = (reg&3)+8*(reg>>2)+4*hi) -> B operand for k-slice ks (16 rows)
__device__ __forceinline__ bf16x8 pack_b(const f32x16& p, int ks) {
    bf16x8 o;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        o[t] = (bf16_t)p[(2 * ks) * 4 + t];
        o[4 + t] = (bf16_t)p[(2 * ks + 1) * 4 + t];
    }
    return o;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.0f;
    return z;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// raw v_exp_f32 (2^x): arguments here are <= ~6 and results below 2^-126 may flush to 0, which is what
// softmax wants; exp2f() would wrap every call in denormal-range fix-ups (~6 VALU instead of 1)
#define EXP2(x) __builtin_amdgcn_exp2f(x)

// =============================================================================================
// forward
// =============================================================================================
template <bool USE_TR, int MINW>
__global__ void __launch_bounds__(640, MINW)
attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o, long ldo,
                float* __restrict__ lse2, int H, int S, int Sp, int W, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kt = smem;
    char* Vt = smem + (size_t)Sp * 128;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const bf16_t* base = qkv + (long)b * S * ld + h * 64;
    stage_tile(Kt, base + W, ld, S, Sp, w, nw, lane);
    stage_tile(Vt, base + 2 * W, ld, S, Sp, w, nw, lane);
    __syncthreads();

    const int ntiles = Sp / 32;
    const FragOffs fo = make_offs(lane);
    constexpr float RESCALE_THR = 6.0f;   // log2 units: P stays <= 2^6 between rescales (fp32 accumulate)
    for (int qt = w; qt < ntiles; qt += nw) {
        const int q = qt * 32 + l31;
        const int qc = min(q, S - 1);
        bf16x8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = frag_global(base, ld, qc, kk, lane);
        f32x16 oacc[2] = {zero16(), zero16()};
        float m = -INFINITY, l = 0.0f;
        for (int kt = 0; kt < ntiles; ++kt) {
            f32x16 s = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) s = MFMA(frag_rm(Kt, kt * 32, fo.rm[kk]), qf[kk], s);
            if (kt * 32 + 32 > S) {   // tail tile: padded keys contribute exp2(-inf) = 0
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= S) s[r] = -INFINITY;
            }
            float tmax = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * scale_log2;
            if (__any(tmax > m + RESCALE_THR)) {   // wave-uniform; always taken on the first tile (m = -inf)
                const float mnew = fmaxf(m, tmax);
                const float alpha = EXP2(m - mnew);
                l *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m = mnew;
            }
            float psum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = EXP2(fmaf(s[r], scale_log2, -m)); psum += s[r]; }
            l += psum;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pb = pack_b(s, ks);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    oacc[dt] = MFMA(frag_t<USE_TR>(Vt, kt * 32 + ks * 16, fo, dt, lane), pb, oacc[dt]);
            }
        }
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / ltot;
        if (q < S) {
            bf16_t* orow = o + ((long)b * S + q) * ldo + h * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(oacc[dt][g * 4 + e] * inv);
                    *(bf16x4*)(orow + dt * 32 + 8 * g + 4 * hi) = ov;
                }
            if (hi == 0 && lse2) lse2[((long)b * H + h) * Sp + q] = m + log2f(ltot);
        }
    }
}

// =============================================================================================
// forward, S = 32 NK + 1 (257 = 256 patch tokens + the class token): NK waves, no padded tiles.
// The generic kernel pads S to NK + 1 tiles each way: 81 MFMA/softmax blocks for 64.5 blocks of work, and its 9th
// wave (one valid query) makes 9 waves share 4 SIMDs.  Here wave w owns query tile w and walks the NK full key
// tiles; the odd key (token S-1) is folded in with VALU dot products (one score per query: 32 FMAs + a shuffle, no
// MFMA block); the odd query (row S-1) is split over the waves - wave w evaluates it against key tile w - and the NK
// partial (max, sum, O) triples are merged through 2 KiB of LDS.
// =============================================================================================
// One (image, head) of the S = 32 NK + 1 forward with K and V staged in LDS (Kt / Vt) - shared by the one-workgroup-per-head
// kernel and the persistent double-buffered one.  PRE: the wave's own query fragments were requested by the caller
// (qf_pre); otherwise they are loaded here.  Ends with the workgroup barrier + wave 0's merge of the odd query.
template <int NK, bool PRE>
__device__ __forceinline__ void attn_fwd_odd_head(const char* Kt, const char* Vt, float* mrg, const bf16_t* base, long ld,
                                                  bf16_t* obase, long ldo, float* lse_bh, float scale_log2, int lane, int w,
                                                  const bf16x8 (&qf_pre)[4], const bf16x8 (&qo_pre)[4]) {
    constexpr int S = 32 * NK + 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const FragOffs fo = make_offs(lane);
    constexpr float RESCALE_THR = 6.0f;
    // row S-1 = 32 NK of a tile: swz_key(row) == 0, i.e. its chunks are not swizzled
    const char* kl = Kt + (S - 1) * 128;
    const char* vl = Vt + (S - 1) * 128;

    auto rescale = [&](float tmax, float& m, float& l, f32x16 (&oacc)[2]) {
        if (__any(tmax > m + RESCALE_THR)) {
            const float mnew = fmaxf(m, tmax);
            const float alpha = EXP2(m - mnew);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m = mnew;
        }
    };
    auto block = [&](const bf16x8 (&qf)[4], int kt, float& m, float& l, f32x16 (&oacc)[2]) {
        f32x16 sc = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) sc = MFMA(frag_rm(Kt, kt * 32, fo.rm[kk]), qf[kk], sc);
        float tmax = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, sc[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * scale_log2;
        rescale(tmax, m, l, oacc);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = EXP2(fmaf(sc[r], scale_log2, -m)); psum += sc[r]; }
        l += psum;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pb = pack_b(sc, ks);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[dt] = MFMA(frag_tr(Vt, kt * 32 + ks * 16, fo, dt), pb, oacc[dt]);
        }
    };
    // the odd key: one score per query (this lane holds 32 of its query's 64 dims, lane ^ 32 the others)
    auto odd_key = [&](const bf16x8 (&qf)[4], float& m, float& l, f32x16 (&oacc)[2]) {
        float sc = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 kv = *(const bf16x8*)(kl + (kk * 2 + hi) * 16);
#pragma unroll
            for (int e = 0; e < 8; ++e) sc = fmaf((float)qf[kk][e], (float)kv[e], sc);
        }
        sc += __shfl_xor(sc, 32, 64);
        rescale(sc * scale_log2, m, l, oacc);
        const float pj = EXP2(fmaf(sc, scale_log2, -m));
        if (hi == 0) l += pj;                       // l is a per-lane partial (summed over the two halves at the end)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x4 vv = *(const bf16x4*)(vl + (dt * 32 + 8 * g + 4 * hi) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[dt][g * 4 + e] = fmaf(pj, (float)vv[e], oacc[dt][g * 4 + e]);
            }
    };

    {   // ---- this wave's 32 queries ----
        const int q = w * 32 + l31;
        bf16x8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = PRE ? qf_pre[kk] : frag_global(base, ld, q, kk, lane);
        f32x16 oacc[2] = {zero16(), zero16()};
        float m = -INFINITY, l = 0.0f;
        // (Two score tiles in flight - the S MFMAs of key tile kt + 1 issued in front of the softmax arithmetic of tile kt, T15 of
        // the guide - measured SLOWER here, 75-77 against 72-74 us on one box, profiles/r05_attn_ab_fwd_two_tiles_in_flight.log: with
        // four waves per SIMD the partner waves already fill the gaps, and the form costs 13 registers and a recomputed last tile.)
#pragma unroll 1
        for (int kt = 0; kt < NK; ++kt) block(qf, kt, m, l, oacc);
        odd_key(qf, m, l, oacc);
        const float ltot = l + __shfl_xor(l, 32, 64);
        const float inv = 1.0f / ltot;
        bf16_t* orow = obase + (long)q * ldo;
        // A lane holds 4 consecutive d per (dt, g) and its partner lane ^ 32 the next 4: v_permlane32_swap pairs the two
        // 8-byte halves of the even chunk on the lower lanes and of the odd chunk on the upper lanes, so that the row leaves
        // as four 16-byte stores per lane instead of eight 8-byte ones (the store tail of this kernel is issue-bound).
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                bf16x4 ev, od;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ev[e] = (bf16_t)(oacc[dt][(2 * pr) * 4 + e] * inv);
                    od[e] = (bf16_t)(oacc[dt][(2 * pr + 1) * 4 + e] * inv);
                }
                typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                const u32x2v e2 = __builtin_bit_cast(u32x2v, ev), o2 = __builtin_bit_cast(u32x2v, od);
                // (vdst, src) -> vdst keeps its lower lanes and takes src's lower lanes into its upper half; src takes
                // vdst's upper lanes into its lower half
                const auto r0 = __builtin_amdgcn_permlane32_swap(e2[0], o2[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(e2[1], o2[1], false, false);
                const u32x4v out = {r0[0], r1[0], r0[1], r1[1]};
                *(u32x4v*)(orow + dt * 32 + 16 * pr + 8 * hi) = out;
            }
        if (hi == 0 && lse_bh) lse_bh[q] = m + log2f(ltot);
    }
    {   // ---- the odd query against key tile w, in the TRANSPOSED orientation: S = q K^T has the keys on the lanes and the
        // (identical) query rows on the registers, so the softmax costs one exp per lane instead of sixteen; p goes
        // through 64 B of LDS into the B-operand layout of the P.V product ----
        bf16x8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = PRE ? qo_pre[kk] : frag_global(base, ld, S - 1, kk, lane);
        f32x16 st = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) st = MFMA(qf[kk], frag_rm(Kt, w * 32, fo.rm[kk]), st);
        const float sc = st[0] * scale_log2;                  // key w*32 + l31 (same value on both lane halves)
        float m = sc;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const float pj = EXP2(sc - m);
        float l = pj;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) l += __shfl_xor(l, off, 64);
        bf16_t* pst = (bf16_t*)(mrg + NK * 66) + w * 32;
        if (hi == 0) pst[l31] = (bf16_t)pj;
        f32x16 oacc[2] = {zero16(), zero16()};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x4 lo = *(const bf16x4*)(pst + 16 * ks + 4 * hi), up = *(const bf16x4*)(pst + 16 * ks + 8 + 4 * hi);
            bf16x8 pb;
#pragma unroll
            for (int t = 0; t < 4; ++t) { pb[t] = lo[t]; pb[4 + t] = up[t]; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) oacc[dt] = MFMA(frag_tr(Vt, w * 32 + ks * 16, fo, dt), pb, oacc[dt]);
        }
        if (w == 0) odd_key(qf, m, l, oacc);                  // l is the wave total here: lane 0's copy is the one kept
        if (l31 == 0) {
            float* dst = mrg + w * 66;
            if (hi == 0) { dst[0] = m; dst[1] = l; }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[2 + dt * 32 + 8 * g + 4 * hi + e] = oacc[dt][g * 4 + e];
        }
    }
    __syncthreads();
    if (w == 0) {   // merge: lane = head dim
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < NK; ++i) M = fmaxf(M, mrg[i * 66]);
        float L = 0.0f, O = 0.0f;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const float c = EXP2(mrg[i * 66] - M);
            L = fmaf(mrg[i * 66 + 1], c, L);
            O = fmaf(mrg[i * 66 + 2 + lane], c, O);
        }
        obase[(long)(S - 1) * ldo + lane] = (bf16_t)(O / L);
        if (lane == 0 && lse_bh) lse_bh[S - 1] = M + log2f(L);
    }
}

template <int NK>
__global__ void __launch_bounds__(NK * 64, 4)    // 4 waves per SIMD = two 8-wave workgroups per CU (128 VGPRs)
attn_fwd_odd_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o, long ldo,
                    float* __restrict__ lse2, int H, long W, float scale_log2, AttnLayout lay) {
    constexpr int S = 32 * NK + 1, Sp = 32 * NK + 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kt = smem;
    char* Vt = smem + (size_t)Sp * 128;
    float* mrg = (float*)(smem + (size_t)Sp * 256);     // [NK][66]: m, sum, O[64] of the odd query per key tile; then NK x 32
                                                          // bf16: p of the odd query, staged into B-operand order
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bf16_t* base = qkv + (long)b * lay.qkv_b + (long)h * lay.qkv_h;   // (W = element offset from Q to K, K to V)
    stage_tile(Kt, base + W, ld, S, Sp, w, NK, lane);
    stage_tile(Vt, base + 2 * W, ld, S, Sp, w, NK, lane);
    __syncthreads();
    const bf16x8 none[4] = {};
    attn_fwd_odd_head<NK, false>(Kt, Vt, mrg, base, ld, o + (long)b * lay.o_b + (long)h * lay.o_h, ldo,
                                 lse2 ? lse2 + ((long)b * H + h) * Sp : nullptr, scale_log2, lane, w, none, none);
}

#ifdef RVLM_EXPERIMENTAL_GEMM
// PERSISTENT, DOUBLE-BUFFERED forward (round 5; RVLM_ATTN_FWD_PERSIST): <= 256 workgroups of NK waves (one per CU, 2 waves
// per SIMD) walk the (image, head) pairs with TWO K / V slots in LDS (2 x 72 KiB): right after the barrier that opens head
// i, every wave requests its share of head i + 1's K and V into the other slot (inline-asm LDS-DMA: a compiler-visible one
// would make hipcc drain vmcnt in front of every LDS read of the slot in use) and its own query fragments into registers -
// a head's 98 KB of HBM traffic arrives UNDER the previous head's MFMA / softmax work instead of in front of its own.
// MEASURED SLOWER (profiles/r05_attn_ab_bwd_variants_fwd_persistent.log: 90-93 us against 73-76 for the one-workgroup-per-head
// kernel on the same box): with its traffic fully hidden the forward still needs ~24 k cycles per head at 2 waves per SIMD -
// the kernel is bound by its per-wave dependency chains (issue), not by HBM, and the two 8-wave workgroups per CU of the
// shipped kernel (4 waves per SIMD) are what covers them.  Kept in EXPERIMENTAL builds as the measured negative result.
template <int NK>
__global__ void __launch_bounds__(NK * 64)
attn_fwd_odd_pers_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o, long ldo,
                         float* __restrict__ lse2, int H, long W, float scale_log2, AttnLayout lay, int nbh) {
    constexpr int S = 32 * NK + 1, Sp = 32 * NK + 32, SLOT = Sp * 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* mrg = (float*)(smem + 2 * (size_t)SLOT);
    // next head's K and V -> slot `dst` (8 rows = 1 KiB per instruction; the swizzle rides on the per-lane source address)
    auto stage_asm = [&](char* dst, const bf16_t* src, int w, int lane) {
        for (int blk = w; blk < (Sp >> 3); blk += NK) {
            const int row = blk * 8 + (lane >> 3);
            const bf16_t* gp = src + (long)min(row, S - 1) * ld + ((lane & 7) ^ swz_key(row)) * 8;
            const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)(dst + blk * 1024));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(d) : "memory", "m0");
        }
    };
    auto head_base = [&](int bh) { return qkv + (long)(bh / H) * lay.qkv_b + (long)(bh % H) * lay.qkv_h; };
    // (the query fragments - the wave's own tile and the odd query - are requested IN FRONT of the K / V requests: VMEM returns
    // in order, so a load issued behind them would be handed over only after the whole next head has landed)
    bf16x8 qf_next[4], qo_next[4];
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const bf16_t* b0 = head_base(blockIdx.x);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf_next[kk] = frag_global(b0, ld, w * 32 + (lane & 31), kk, lane);
            qo_next[kk] = frag_global(b0, ld, S - 1, kk, lane);
        }
        stage_asm(smem, b0 + W, w, lane);
        stage_asm(smem + Sp * 128, b0 + 2 * W, w, lane);
    }
    int it = 0;
    for (int bh = blockIdx.x; bh < nbh; bh += gridDim.x, ++it) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));       // (opaque per head: lane-derived offsets must not be hoisted across the head loop)
        const int lane = tid & 63, w = tid >> 6;
        char* cur = smem + (it & 1) * SLOT;
        char* nxt = smem + ((it & 1) ^ 1) * SLOT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this head's K / V (and the previous head's O stores)
        __syncthreads();                                       // ... of every wave; the other slot is free now
        bf16x8 qf[4], qo[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { qf[kk] = qf_next[kk]; qo[kk] = qo_next[kk]; }
        const int nb = bh + (int)gridDim.x;
        if (nb < nbh) {
            const bf16_t* b1 = head_base(nb);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                qf_next[kk] = frag_global(b1, ld, w * 32 + (lane & 31), kk, lane);
                qo_next[kk] = frag_global(b1, ld, S - 1, kk, lane);
            }
            stage_asm(nxt, b1 + W, w, lane);
            stage_asm(nxt + Sp * 128, b1 + 2 * W, w, lane);
        }
        const int b = bh / H, h = bh % H;
        attn_fwd_odd_head<NK, true>(cur, cur + Sp * 128, mrg, head_base(bh), ld, o + (long)b * lay.o_b + (long)h * lay.o_h, ldo,
                                    lse2 ? lse2 + ((long)b * H + h) * Sp : nullptr, scale_log2, lane, w, qf, qo);
    }
}
#endif

// =============================================================================================
// backward prep: D[b,h,q] = sum_d dO[q,d] * O[q,d]
// =============================================================================================
__global__ void __launch_bounds__(256)
attn_bwd_prep_kernel(const bf16_t* __restrict__ o, long ldo, const bf16_t* __restrict__ d_o, long lddo,
                     float* __restrict__ dsum, int H, int S, int Sp) {
    const long tok = blockIdx.x;  // b*S + q
    const int b = (int)(tok / S), q = (int)(tok % S);
    const int lane = threadIdx.x & 63;
    for (int h = threadIdx.x >> 6; h < H; h += 4) {
        const float a = (float)o[tok * ldo + h * 64 + lane];
        const float g = (float)d_o[tok * lddo + h * 64 + lane];
        const float v = wave_sum(a * g);
        if (lane == 0) dsum[((long)b * H + h) * Sp + q] = v;
    }
}

// =============================================================================================
// backward dQ: wave owns a query tile, K and V LDS-resident
// =============================================================================================
template <bool USE_TR>
__global__ void __launch_bounds__(640)
attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ o, long ldo,
                   const bf16_t* __restrict__ d_o, long lddo, const float* __restrict__ lse2,
                   float* __restrict__ dsum, bf16_t* __restrict__ dqkv, long lddq, int H, int S, int Sp, int W,
                   float scale, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kt = smem;
    char* Vt = smem + (size_t)Sp * 128;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const bf16_t* base = qkv + (long)b * S * ld + h * 64;
    stage_tile(Kt, base + W, ld, S, Sp, w, nw, lane);
    stage_tile(Vt, base + 2 * W, ld, S, Sp, w, nw, lane);
    __syncthreads();

    const int ntiles = Sp / 32;
    const FragOffs fo = make_offs(lane);
    for (int qt = w; qt < ntiles; qt += nw) {
        const int q = qt * 32 + l31;
        const int qc = min(q, S - 1);
        bf16x8 qf[4], dof[4];
        // D[q] = sum_d dO[q,d] * O[q,d]  (the "prep" pass of flash backward, fused here: this lane holds
        // 32 of the 64 d's of dO as its MFMA fragments, the other half sits in lane ^ 32)
        float dq_sum = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            qf[kk] = frag_global(base, ld, qc, kk, lane);
            dof[kk] = frag_global(d_o + (long)b * S * lddo + h * 64, lddo, qc, kk, lane);
            const bf16x8 of = frag_global(o + (long)b * S * ldo + h * 64, ldo, qc, kk, lane);
#pragma unroll
            for (int e = 0; e < 8; ++e) dq_sum = fmaf((float)dof[kk][e], (float)of[e], dq_sum);
        }
        dq_sum += __shfl_xor(dq_sum, 32, 64);
        if (hi == 0 && q < S) dsum[((long)b * H + h) * Sp + q] = dq_sum;   // consumed by the dK/dV kernel
        const float lq = lse2[((long)b * H + h) * Sp + qc];
        f32x16 acc[2] = {zero16(), zero16()};
        for (int kt = 0; kt < ntiles; ++kt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                s = MFMA(frag_rm(Kt, kt * 32, fo.rm[kk]), qf[kk], s);
                dp = MFMA(frag_rm(Vt, kt * 32, fo.rm[kk]), dof[kk], dp);
            }
            const bool tail = (kt * 32 + 32 > S);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = EXP2(fmaf(s[r], scale_log2, -lq));
                if (tail) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= S) p = 0.0f;
                }
                s[r] = p * (dp[r] - dq_sum);   // dS^T
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 db = pack_b(s, ks);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    acc[dt] = MFMA(frag_t<USE_TR>(Kt, kt * 32 + ks * 16, fo, dt, lane), db, acc[dt]);
            }
        }
        if (q < S) {
            bf16_t* drow = dqkv + ((long)b * S + q) * lddq + h * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(acc[dt][g * 4 + e] * scale);
                    *(bf16x4*)(drow + dt * 32 + 8 * g + 4 * hi) = ov;
                }
        }
    }
}

// =============================================================================================
// backward dK, dV: wave owns a key tile, Q and dO LDS-resident
// =============================================================================================
template <bool USE_TR, bool KV_LDS>
__global__ void __launch_bounds__(640)
attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ d_o, long lddo,
                    const float* __restrict__ lse2, const float* __restrict__ dsum,
                    bf16_t* __restrict__ dqkv, long lddq, int H, int S, int Sp, int W, float scale,
                    float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: Q | dO | [K | V when KV_LDS] | lse2 | D
    char* Qt = smem;
    char* Dt = smem + (size_t)Sp * 128;
    char* Kt = smem + (size_t)Sp * 256;
    char* Vt = smem + (size_t)Sp * 384;
    float* Ls = (float*)(smem + (size_t)Sp * (KV_LDS ? 512 : 256));
    float* Ds = Ls + Sp;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const bf16_t* base = qkv + (long)b * S * ld + h * 64;
    stage_tile(Qt, base, ld, S, Sp, w, nw, lane);
    stage_tile(Dt, d_o + (long)b * S * lddo + h * 64, lddo, S, Sp, w, nw, lane);
    if (KV_LDS) {
        stage_tile(Kt, base + W, ld, S, Sp, w, nw, lane);
        stage_tile(Vt, base + 2 * W, ld, S, Sp, w, nw, lane);
    }
    for (int i = tid; i < Sp; i += blockDim.x) {
        Ls[i] = (i < S) ? lse2[((long)b * H + h) * Sp + i] : INFINITY;   // pad rows -> p = 0
        Ds[i] = (i < S) ? dsum[((long)b * H + h) * Sp + i] : 0.0f;
    }
    __syncthreads();

    const int ntiles = Sp / 32;
    const FragOffs fo = make_offs(lane);
    for (int kt = w; kt < ntiles; kt += nw) {
        const int key = kt * 32 + l31;
        const int kc = min(key, S - 1);
        bf16x8 kf[4], vf[4];
        if (!KV_LDS) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                kf[kk] = frag_global(base + W, ld, kc, kk, lane);
                vf[kk] = frag_global(base + 2 * W, ld, kc, kk, lane);
            }
        }
        f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
        for (int qt = 0; qt < ntiles; ++qt) {
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 kb = KV_LDS ? frag_rm(Kt, kt * 32, fo.rm[kk]) : kf[kk];
                const bf16x8 vb = KV_LDS ? frag_rm(Vt, kt * 32, fo.rm[kk]) : vf[kk];
                s = MFMA(frag_rm(Qt, qt * 32, fo.rm[kk]), kb, s);     // S[q][key]
                dp = MFMA(frag_rm(Dt, qt * 32, fo.rm[kk]), vb, dp);   // dP[q][key]
                __builtin_amdgcn_sched_barrier(0);   // bound the live fragment set (3 waves/SIMD budget)
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 lq = *(const float4*)(Ls + qt * 32 + 8 * g + 4 * hi);
                const float4 dq = *(const float4*)(Ds + qt * 32 + 8 * g + 4 * hi);
                const float lqa[4] = {lq.x, lq.y, lq.z, lq.w};
                const float dqa[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = EXP2(fmaf(s[g * 4 + e], scale_log2, -lqa[e]));
                    s[g * 4 + e] = p;                                   // P   (in place)
                    dp[g * 4 + e] = p * (dp[g * 4 + e] - dqa[e]);       // dS  (in place)
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pb = pack_b(s, ks), db = pack_b(dp, ks);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = MFMA(frag_t<USE_TR>(Dt, qt * 32 + ks * 16, fo, dt, lane), pb, dv[dt]);
                    dk[dt] = MFMA(frag_t<USE_TR>(Qt, qt * 32 + ks * 16, fo, dt, lane), db, dk[dt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (key < S) {
            bf16_t* krow = dqkv + ((long)b * S + key) * lddq + W + h * 64;
            bf16_t* vrow = krow + W;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 ok, ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ok[e] = (bf16_t)(dk[dt][g * 4 + e] * scale);
                        ov[e] = (bf16_t)dv[dt][g * 4 + e];
                    }
                    *(bf16x4*)(krow + dt * 32 + 8 * g + 4 * hi) = ok;
                    *(bf16x4*)(vrow + dt * 32 + 8 * g + 4 * hi) = ov;
                }
        }
    }
}

// =============================================================================================
// backward, FUSED (dQ, dK, dV in one kernel) for sequence lengths S = 32*NK + 1 (CLIP ViTs at 224 px: 257 = 8*32 + 1)
//
// The two-kernel backward above evaluates P and dS twice (once per kernel: exp2 + ~8 VALU per score, the bound of both
// kernels) and runs 9 waves on 4 SIMDs (one SIMD carries 3).  Here NK = 8 waves (2 per SIMD, <= 256 VGPRs each) own one
// 32-key tile each - K, V fragments and the dK, dV accumulators stay in registers - and walk the 9 query tiles in
// lockstep, ONE barrier per tile:
//     S, dP (8 MFMA 32x32x16) -> P, dS once, in registers -> the wave's dS tile [32 keys][32 q] to LDS -> barrier ->
//     dV, dK from the registers (8 MFMA) ; dQ: wave w OWNS the 16 (d) x 16 (q) block (w >> 1, w & 1) of the tile's dQ^T
//     and contracts it over ALL 256 keys - K^T fragments of the NK key tiles held in registers against the NK waves'
//     dS tiles read back through ds_read_b64_tr_b16 (NK v_mfma_f32_16x16x32_bf16) -> 8-byte stores of finished dQ.
// = 5 tile GEMMs per (query tile, key tile), P / dS evaluated once, and NO partial sums cross waves: round 1 summed eight
// fp32 32x64 partials per tile through 64 KiB of LDS with two barriers (3.8 k cycles per tile, now 2.7 k).  The dS tiles
// are double-buffered by tile parity, which is what makes one barrier per tile enough.  The result is a deterministic
// function of the inputs (fixed MFMA accumulation order).  dK / dV leave through a wave-private LDS transpose as 16-byte
// full-line stores (the row-per-lane 8-byte store tail was 10 % of the kernel, now 4 %).
//
// The odd key (index S-1, the "+1") is handled up front (phase 1) in the transposed orientation, where it is ONE
// accumulator register instead of a 32-wide tile: S^T, dP^T by MFMA against the padded last key tile, p and dS for that
// key on one register per lane, dK/dV of that key by MFMA against a one-column B operand; its rank-1 contribution to dQ
// (dS[q] * k) joins the owner's block before the store.  The odd QUERY rides in the ninth query tile (31 padded rows,
// lse = +inf -> P = 0); its rows as 64-wide vectors are requested under the staging.
//
// LDS (S = 257): Q 36 KiB | dO 36 KiB | area 72 KiB (first the K and V tiles, then 2 x NK dS tiles of 2 KiB and NK
// store-staging tiles of 4 KiB) | lse, D, p_odd, dS_odd [288] | k_odd [64] | per-wave dK/dV of the odd key.
// Measured (MI355X, B = 128, 16 heads; profiles/r02_attn_bwd_*.log): 262 -> 206 us per launch.
// =============================================================================================
constexpr int FB2_TILE = 2048;   // one wave's dS tile [32 keys][32 q] bf16

// Cross-lane sums WITHOUT address registers (ds_swizzle bit mode for lane ^ 1 .. 16, v_permlane32_swap for lane ^ 32).  __shfl_xor
// is ds_bpermute: one per-lane address VGPR per distance, derived from the lane id - loop-invariant values that hipcc hoists out of
// the persistent head loop, where the fused backward has no register left for them (they were parked in scratch).
__device__ __forceinline__ float xor32_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0] = the lower half's value everywhere, r[1] the upper's
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float wave_sum_swz(float v) {
#define RVLM_SWZ_ADD(K) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), ((K) << 10) | 0x1f))
    RVLM_SWZ_ADD(1); RVLM_SWZ_ADD(2); RVLM_SWZ_ADD(4); RVLM_SWZ_ADD(8); RVLM_SWZ_ADD(16);
#undef RVLM_SWZ_ADD
    return xor32_sum(v);
}

// PF (round 6, "everything of the next head arrives under this one"): the persistent form above still staged K and V of every
// head in its phase 0 - 72 KB of HBM misses per workgroup at the ~12 B per clock a CU pulls them, 28 % of the kernel, exposed
// (profiles/r05_attn_ab_bwd_variants_fwd_persistent.log).  K and V are only read in phase 1 (fragments to registers), and V only
// row-major by the wave that owns the key tile.  So with PF:
//   * V never touches LDS: a wave's V fragments (32 keys x 64 d = the 4 KiB it owns) are plain global loads to registers, like
//     the O rows; the odd key's V row sits in a 128-B LDS array (Ve) next to its K row;
//   * the K tile keeps its own 36 KiB, dead after phase 1: the NEXT head's K lands there during the query-tile loop (4 more
//     1-KiB DMAs per step next to the 8 of Q / dO);
//   * the dK / dV store staging (8 x 4 KiB) aliases the dS tiles (dead after the loop; one barrier in front of phase 3, which
//     replaces the end-of-head barrier: the count per head stays 12);
//   * the next head's V / O fragments, lse, odd rows are requested into registers at the top of phase 3 and land under the stores.
// LDS: Q 36 | dO 36 | K 36 | dS / staging 32 | small arrays 9.6 KiB = 149.6 KiB.  Phase 0 of every head but a workgroup's first
// is then: commit the small arrays from registers, s_waitcnt vmcnt(0), barrier.
// MEASURED (profiles/r06_attn_bwd_prefetch_ab.log, same box, three alternations): phase 0 12.9 k -> 4.2 k cycles per head as
// designed, but the tile loop 25.6 k -> 30.9 k (12 instead of 8 DMA requests per step through the CU's one texture queue, in front
// of the step's barrier: ~150 cycles of critical path each) and the store phase 1.9 k -> 4.7 k (the register prefetches queue in
// front of the stores): 190.5-192 -> 198-203 us per launch, 44.5 -> 46.9 ms per step in the pipeline.  NOT the shipped form: PF =
// true is only instantiated by make EXPERIMENTAL=1 (RVLM_ATTN_BWD_PF=1).  The bytes of a head have to come through that queue
// somewhere; exposed in phase 0 they at least cost no barrier-synchronised step time.
template <int NK, bool PF>
__global__ void __launch_bounds__(NK * 64)
attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ o, long ldo,
                      const bf16_t* __restrict__ d_o, long lddo, const float* __restrict__ lse2,
                      bf16_t* __restrict__ dqkv, long lddq, int H, int S, long W, float scale, float scale_log2,
                      unsigned long long* __restrict__ trace, int desync, AttnLayout lay, int nbh) {
    // Phase offset: every CU would otherwise stage its head at the same moment (6 TB/s-bound, 22 % of the kernel spent
    // waiting for HBM) and compute at the same moment (HBM idle).  The first workgroup of each CU starts up to 7 x desync
    // kilo-cycles late; the stagger then persists, one CU's staging hides under the others' compute.  (RVLM_ATTN_DESYNC,
    // default 0 since round 3: worth 5 % on the one-head-per-workgroup kernel of round 1, but the persistent kernel requests
    // the next head's Q / dO under its tile loop and then the late start only costs its tail - in-pipeline attention
    // backward 46.5 ms per step at 5, 44.9 at 0, 49.3 at 9; two phases 4-8 k cycles apart: 195 vs 202 us in the micro-benchmark,
    // 44.5 vs 44.7 ms in the pipeline; profiles/r03_ab_attn_desync.log)
    const int trace_mode = desync >> 8;       // (RVLM_ATTN_TRACE=2: per-wave duration of phase 1 instead of the phase stamps)
    desync &= 255;
    if (desync > 0 && blockIdx.x < 256) {
        for (int i = 0; i < (int)((blockIdx.x >> 3) & 7) * desync; ++i) __builtin_amdgcn_s_sleep(16);   // 16 x 64 cycles
    }
    constexpr int NT = NK + 1, Sp = NT * 32, SE = NK * 32;   // query tiles, padded rows, index of the odd key
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qt = smem;
    char* Dt = smem + Sp * 128;
    char* area = smem + Sp * 256;                              // K | V tiles, later NK partial slots (PF: K tile | dS tiles)
    char* Kt = area;
    char* Vt = area + Sp * 128;                                // (not PF)
    char* dsarea = PF ? area + Sp * 128 : area;                // dS tiles (+ store staging behind them; PF: aliased onto them)
    float* Ls = (float*)(PF ? dsarea + 2 * NK * FB2_TILE : area + Sp * 256);
    float* Ds = Ls + Sp;
    float* Pe = Ds + Sp;                                       // p[q][odd key]
    float* De = Pe + Sp;                                       // dS[q][odd key]
    float* Ke = De + Sp;                                       // k[odd key][0..63] as fp32
    float* KVe = Ke + 64;                                      // [NK + 1][2][64] per-wave (+ odd query) dK / dV of the odd key
    // PF: the odd token's rows, raw bf16 [4][64]: dO, O, Q, V.  V's is also the odd KEY's row as an MFMA operand (Ve).  They
    // arrive by LDS-DMA (two 4-byte-per-lane requests of wave 0: lanes 0-31 one row, 32-63 the next), not through registers
    bf16_t* Orow = (bf16_t*)(KVe + (NK + 1) * 128);
    bf16_t* Ve = Orow + 3 * 64;
    // PF: what the next head needs in REGISTERS, requested at the top of phase 3 (a workgroup's first head: in its phase 0)
    bf16x8 vf_n[4], ofr_n[4];
    // (the 16-bit ones stay RAW until they are used: converted where they are loaded, every load got its own s_waitcnt vmcnt(0) -
    // four serial HBM round trips at the top of phase 3)
    float ls_n = INFINITY;
    unsigned short kv_n = 0;
    auto raw16 = [](const bf16_t* q) { return *(const unsigned short*)q; };
    auto cvt16 = [](unsigned short u) { return __uint_as_float((unsigned)u << 16); };
    auto fetch_regs = [&](int bh2, int tid2) __attribute__((always_inline)) {
        const int b2 = bh2 / H, h2 = bh2 % H, w2 = tid2 >> 6, lane2 = tid2 & 63;
        const bf16_t* base2 = qkv + (long)b2 * lay.qkv_b + (long)h2 * lay.qkv_h;
        const bf16_t* dob2 = d_o + (long)b2 * lay.o_b + (long)h2 * lay.o_h;
        const bf16_t* ob2 = o + (long)b2 * lay.o_b + (long)h2 * lay.o_h;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            vf_n[kk] = frag_global(base2 + 2 * W, ld, w2 * 32 + (lane2 & 31), kk, lane2);
            ofr_n[kk] = frag_global(ob2, ldo, w2 * 32 + (lane2 & 31), kk, lane2);
        }
        ls_n = (tid2 < S) ? lse2[((long)b2 * H + h2) * Sp + tid2] : INFINITY;
        if (tid2 < 64) kv_n = raw16(base2 + (long)SE * ld + W + tid2);
        if (w2 == 0) {
            // rows dO | O, then Q | V of token SE -> Orow[0..1], Orow[2..3] (readers: wave 0 at the end of phase 1 and the odd-key
            // operand of every wave in phase 1 - all in front of the barrier that precedes this call)
            const int half = lane2 >> 5, c = (lane2 & 31) * 2;
            const bf16_t* g0 = (half ? ob2 + (long)SE * ldo : dob2 + (long)SE * lddo) + c;
            const bf16_t* g1 = base2 + (long)SE * ld + (half ? 2 * W : 0) + c;
            const unsigned d0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)Orow);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g0), "s"(d0) : "memory", "m0");
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g1), "s"(d0 + 256u) : "memory", "m0");
        }
    };
    // PERSISTENT over the (image, head) pairs bh = blockIdx.x, + gridDim.x, ... (the launcher gives <= 256 workgroups when
    // RVLM_ATTN_PERSIST is on): while a head's query-tile loop runs, the NEXT head's Q and dO tiles are requested into the
    // LDS tile slots the loop has finished with (one 1-KiB DMA per wave and step, through inline asm: a compiler-visible
    // LDS-DMA would make hipcc drain vmcnt in front of every LDS read that might alias it).  A CU pulls HBM misses at only
    // ~12 B per clock, so the 144 KB staging of a head cost 27 % of the kernel; half of it now arrives under the loop.
    bool have_qd = false;       // Q / dO of the head about to start are already in LDS
    const int wave_u = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (int bh = blockIdx.x; bh < nbh; bh += gridDim.x) {
    // (opaque per iteration: otherwise every lane-derived offset of the phases below is hoisted out of this loop, lives
    // across the whole head - 256 VGPRs, 52 spilled dwords - and the kernel ran 40 % slower than its one-head form)
    // (PF: rebuilt from the wave index (an SGPR) and mbcnt instead of from threadIdx.x, whose register would have to survive the
    // whole head next to 256 live ones - it was the one value hipcc parked in scratch and reloaded at the top of every head)
    int tid = threadIdx.x;
    if (PF) {    // (inside a volatile asm: as builtins the two mbcnt are loop-invariant and get hoisted - and spilled - again)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(tid));
        tid += wave_u * 64;
    }
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // optional phase timeline (RVLM_ATTN_TRACE=1: 5 s_memtime stamps per head into the dsum scratch buffer)
    auto stamp = [&](int k) {
        if (trace && trace_mode < 2 && tid == 0) trace[(long)bh * 8 + k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const int b = bh / H, h = bh % H;
    const int nxt = bh + (int)gridDim.x;
    const bool prefetch_next = nxt < nbh;
    const bf16_t* base = qkv + (long)b * lay.qkv_b + (long)h * lay.qkv_h;
    const bf16_t* dob = d_o + (long)b * lay.o_b + (long)h * lay.o_h;
    const bf16_t* ob = o + (long)b * lay.o_b + (long)h * lay.o_h;
    // the next head's Q / dO blocks (8 rows = one DMA instruction): wave w < 4 takes block 4 t + w of Q, the others of dO
    const bf16_t* nsrc = nullptr;
    if (prefetch_next) {
        const int nb = nxt / H, nh = nxt % H;
        nsrc = w < 4 ? qkv + (long)nb * lay.qkv_b + (long)nh * lay.qkv_h : d_o + (long)nb * lay.o_b + (long)nh * lay.o_h;
    }
    const long nld = w < 4 ? ld : lddo;
    // ---- phase 0: stage Q, dO (unless they came in under the previous head's loop), K, V; lse ----------------------
    if (!have_qd) {
        stage_tile(Qt, base, ld, S, Sp, w, NK, lane);
        stage_tile(Dt, dob, lddo, S, Sp, w, NK, lane);
        if (PF) { stage_tile(Kt, base + W, ld, S, Sp, w, NK, lane); fetch_regs(bh, tid); }
    }
    bf16x8 ofr[4];   // O rows of this wave's query tile (for D = rowsum(dO * O)): requested under the staging
    if (!PF) {
        stage_tile(Kt, base + W, ld, S, Sp, w, NK, lane);
        stage_tile(Vt, base + 2 * W, ld, S, Sp, w, NK, lane);
        for (int i = tid; i < Sp; i += NK * 64) Ls[i] = (i < S) ? lse2[((long)b * H + h) * Sp + i] : INFINITY;
        if (tid < 64) Ke[tid] = (float)base[(long)SE * ld + W + tid];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ofr[kk] = frag_global(ob, ldo, w * 32 + (lane & 31), kk, lane);
    } else {
        // commit what came in registers (every wave is past the previous head's loop: the barrier in front of its phase 3)
        if (tid < Sp) Ls[tid] = ls_n;
        if (tid < 64) Ke[tid] = cvt16(kv_n);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) ofr[kk] = ofr_n[kk];
    }
    // the odd query's rows as 64-wide vectors (ONE wave), requested under the staging as well: fetched in phase 1 they put
    // an HBM round trip on the path to the phase's barrier.  That wave is wave 0: per-wave stamps (RVLM_ATTN_TRACE=2) put
    // the first-dispatched waves 0-3 at the phase's barrier after 4.3 k cycles, waves 4-6 after 5.6 k, and the wave that
    // carries these 1.7 k cycles of serial shuffles on top - it used to be the last one - after 7.3 k
    constexpr int ODD_Q_WAVE = 0;
    float odd_do = 0.0f, odd_o = 0.0f, odd_q = 0.0f, odd_v = 0.0f;
    if (!PF && w == ODD_Q_WAVE) {
        odd_do = (float)dob[(long)SE * lddo + lane]; odd_o = (float)ob[(long)SE * ldo + lane];
        odd_q = (float)base[(long)SE * ld + lane]; odd_v = (float)base[(long)SE * ld + 2 * W + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the inline-asm requests of the previous head's loop)
    __syncthreads();
    stamp(1);
    const unsigned long long t_p1 = (trace && trace_mode == 2) ? __builtin_amdgcn_s_memtime() : 0ull;

    const FragOffs fo = make_offs(lane);
    // ---- phase 1: this wave's key tile in registers; D for its query tile(s); the odd key -------------------
    // (PF: D = rowsum(dO * O) of this wave's query tile FIRST - the prefetched O fragments are the 16 registers this phase has
    // no room for next to the K^T fragments; with the sum placed where it stood, behind them, hipcc parked 12 of the
    // prefetched dwords in scratch right behind their loads, i.e. waited for them in phase 3)
    float dsum_pf = 0.0f;
    if (PF) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 dof0 = frag_rm(Dt, w * 32, fo.rm[kk]), of = ofr[kk];
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum_pf = fmaf((float)dof0[e], (float)of[e], dsum_pf);
        }
        dsum_pf = xor32_sum(dsum_pf);
        __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { kf[kk] = frag_rm(Kt, w * 32, fo.rm[kk]); vf[kk] = PF ? vf_n[kk] : frag_rm(Vt, w * 32, fo.rm[kk]); }
    // dQ ownership: wave w computes the 16 (d) x 16 (q) block (db, qb) of every query tile's dQ^T over ALL keys, so no
    // partial sums cross waves.  Its A operands K^T[16 d][32 keys] of the NK key tiles stay in registers
    // (v_mfma_f32_16x16x32_bf16: lane <-> d = 16 db + (lane & 15), k = 8 (lane >> 4) + 0..7 <-> key).
    const int db = w >> 1, qb = w & 1, G = lane >> 4, i16 = lane & 15;
    bf16x8 kq[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = j * 32 + 8 * G + 4 * r + (i16 >> 2);
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)((lds_char*)Kt + swz_off(row, 2 * db + ((i16 & 3) >> 1)) + (i16 & 1) * 8));
            kq[j][4 * r + 0] = v[0]; kq[j][4 * r + 1] = v[1]; kq[j][4 * r + 2] = v[2]; kq[j][4 * r + 3] = v[3];
        }
    if (trace) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(5); }     // K / V / K^T fragments in registers
    {
        f32x16 dke[2] = {zero16(), zero16()}, dve[2] = {zero16(), zero16()};
        {
            const int qe = w;   // query tiles 0..NK-1: one per wave (the odd query, tile NK, takes the vector path below)
            const int q = qe * 32 + l31;
            bf16x8 qf[4], dof[4];
            float dsum = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                qf[kk] = frag_rm(Qt, qe * 32, fo.rm[kk]);
                dof[kk] = frag_rm(Dt, qe * 32, fo.rm[kk]);
                if (!PF) {
                    const bf16x8 of = ofr[kk];
#pragma unroll
                    for (int e = 0; e < 8; ++e) dsum = fmaf((float)dof[kk][e], (float)of[e], dsum);
                }
            }
            if (PF) dsum = dsum_pf; else
            dsum += __shfl_xor(dsum, 32, 64);
            // S^T / dP^T against the last (padded) key tile: lane <-> query, register 0 of the hi = 0 lanes <-> odd key
            f32x16 sT = zero16(), dpT = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                sT = MFMA(frag_rm(Kt, SE, fo.rm[kk]), qf[kk], sT);
                if (PF) {     // the padded last key tile of V: row 0 (lanes 0 and 32) = the odd key's row, the others zero
                    bf16x8 ve = *(const bf16x8*)(Ve + (kk * 2 + hi) * 8);
                    if (l31 != 0) ve = bf16x8{};
                    dpT = MFMA(ve, dof[kk], dpT);
                } else
                dpT = MFMA(frag_rm(Vt, SE, fo.rm[kk]), dof[kk], dpT);
            }
            const float pe = EXP2(fmaf(sT[0], scale_log2, -Ls[q]));
            if (hi == 0) { Ds[q] = dsum; Pe[q] = pe; De[q] = pe * (dpT[0] - dsum); }
            if (trace) stamp(6);                                                       // D, p and dS of the odd key
            // dV, dK of the odd key: B operand with the single column n = 0 (lanes 0 and 32), k <-> the 32 queries
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pb, db;
#pragma unroll
                for (int half = 0; half < 2; ++half) {     // k index t = 4 half + e <-> query qe*32 + ks*16 + 4 hi + 8 half + e
                    const int q0 = qe * 32 + ks * 16 + 4 * hi + 8 * half;
                    const float4 p4 = *(const float4*)(Pe + q0), d4 = *(const float4*)(De + q0);
                    const float pa[4] = {p4.x, p4.y, p4.z, p4.w}, da[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pb[4 * half + e] = (bf16_t)(l31 == 0 ? pa[e] : 0.0f);
                        db[4 * half + e] = (bf16_t)(l31 == 0 ? da[e] : 0.0f);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dve[dt] = MFMA(frag_tr(Dt, qe * 32 + ks * 16, fo, dt), pb, dve[dt]);
                    dke[dt] = MFMA(frag_tr(Qt, qe * 32 + ks * 16, fo, dt), db, dke[dt]);
                }
            }
        }
        if (l31 == 0) {   // column 0: rows d = dt*32 + 8*g + 4*hi + 0..3 in registers 4 g .. 4 g + 3
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * hi;
                    *(float4*)(KVe + (w * 2 + 0) * 64 + d) = make_float4(dke[dt][4 * g], dke[dt][4 * g + 1], dke[dt][4 * g + 2], dke[dt][4 * g + 3]);
                    *(float4*)(KVe + (w * 2 + 1) * 64 + d) = make_float4(dve[dt][4 * g], dve[dt][4 * g + 1], dve[dt][4 * g + 2], dve[dt][4 * g + 3]);
                }
        }
    }
    if (w == ODD_Q_WAVE) {
        // the odd QUERY (row SE) against the odd key, as 64-wide vectors (lane <-> d): D, p, dS, and its dK / dV terms
        if (PF) { odd_do = (float)Orow[lane]; odd_o = (float)Orow[64 + lane]; odd_q = (float)Orow[128 + lane]; odd_v = (float)Orow[192 + lane]; }
        const float dov = odd_do, ov = odd_o, qv = odd_q, kv = Ke[lane], vv = odd_v;
        const float dsum = PF ? wave_sum_swz(dov * ov) : wave_sum(dov * ov), sc = PF ? wave_sum_swz(qv * kv) : wave_sum(qv * kv),
                    dpe = PF ? wave_sum_swz(dov * vv) : wave_sum(dov * vv);
        const float pe = EXP2(fmaf(sc, scale_log2, -Ls[SE]));
        const float de = pe * (dpe - dsum);
        if (lane < 32) { Ds[SE + lane] = (lane == 0) ? dsum : 0.0f; Pe[SE + lane] = (lane == 0) ? pe : 0.0f; De[SE + lane] = (lane == 0) ? de : 0.0f; }
        KVe[(NK * 2 + 0) * 64 + lane] = de * qv;
        KVe[(NK * 2 + 1) * 64 + lane] = pe * dov;
    }
    if (trace) stamp(7);                                                               // this wave's part of phase 1 done
    if (trace && trace_mode == 2 && lane == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        trace[(long)bh * 8 + w] = __builtin_amdgcn_s_memtime() - t_p1;
    }
    __syncthreads();   // K / V tiles are dead: the area becomes the partial slots; Ds / Pe / De are complete
    stamp(2);
    // ---- phase 2: lockstep walk over the query tiles -------------------------------------------------------------
    // Per tile: S, dP (8 MFMA) -> P, dS in registers -> dS tile to LDS -> ONE barrier -> dV, dK from the registers
    // (8 MFMA) and this wave's dQ^T block from the NK waves' dS tiles (NK v_mfma_f32_16x16x32_bf16) -> dQ stored.
    // The dS tiles are double-buffered by tile parity: a wave that writes tile qt + 1 has passed barrier qt, which
    // every wave reaches only after its reads of tile qt - 1 - one barrier per tile is enough.
    lds_char* dsb = (lds_char*)dsarea;                        // [2][NK] tiles of FB2_TILE bytes: [32 keys][32 q] bf16
    // write: this lane's key row (64-B rows), 8-B chunk (4 consecutive q) index XOR ((key >> 1) & 7): the 8 same-parity
    // rows of a 16-lane ds_write_b64 group get 8 different keys, and the two rows r, r + 8 that share a bank phase in
    // the transposing read below differ in key bit 2 = the other 32-B half of the row (round 2 XORed (key >> 2) & 7: 2-way
    // conflicts on every write and every read, 24 extra LDS cycles per query tile and wave in scripts/lds_bank_model.py)
    const int st_w = w * FB2_TILE + l31 * 64;
    const int st_sw = (l31 >> DS_KEY_SHIFT) & 7;
    // read (B operand of the 16x16x32 MFMA, lane <-> q = 16 qb + (lane & 15), k <-> key = 8 G + 0..7): transposing read of
    // the 4 (keys) x 16 (q) block; lane i of a 16-lane group addresses key row 8 G + 4 r + (i >> 2), 8-B chunk 4 qb + (i & 3)
    int rd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * G + 4 * r + (i16 >> 2);
        rd[r] = row * 64 + (((4 * qb + (i16 & 3)) ^ ((row >> DS_KEY_SHIFT) & 7)) << 3);
    }
    const float4 ke4 = *(const float4*)(Ke + 16 * db + 4 * G);        // k[odd key][d], d = 16 db + 4 G + 0..3
    bf16_t* dq_out = dqkv + (long)b * lay.qkv_b + (long)h * lay.qkv_h + 16 * db + 4 * G;
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    // the next head's tile t of Q / dO into the slot of this head's tile t: legal once every wave has passed the barrier of
    // step t (its fragment reads of the tile are waited for before that barrier).  Issued BEFORE the barrier of step t + 1:
    // the request stalls its wave for a couple of hundred cycles in the texture queue, which the early waves would
    // otherwise spend waiting at that barrier
    auto prefetch_tile = [&](int t) {
        const int blk = 4 * t + (w & 3), row = blk * 8 + (lane >> 3);
        const bf16_t* gp = nsrc + (long)min(row, S - 1) * nld + ((lane & 7) ^ swz_key(row)) * 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)((w < 4 ? Qt : Dt) + blk * 1024));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(dst) : "memory", "m0");
    };
    // PF: the next head's K into the K tile, dead since phase 1: blocks 4 t .. 4 t + 3 (8 rows each) at step t, by the waves of
    // parity t (one DMA each, so that a step adds one request to half of the waves only)
    const bf16_t* nk_src = nullptr;
    if (PF && prefetch_next) nk_src = qkv + (long)(nxt / H) * lay.qkv_b + (long)(nxt % H) * lay.qkv_h + W;
    auto prefetch_k = [&](int t) {
        if ((w & 1) != (t & 1)) return;
        const int blk = 4 * t + (w >> 1), row = blk * 8 + (lane >> 3);
        const bf16_t* gp = nk_src + (long)min(row, S - 1) * ld + ((lane & 7) ^ swz_key(row)) * 8;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)(Kt + blk * 1024));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gp), "s"(dst) : "memory", "m0");
    };
    // PF: RUNNING request pointers for steps t < NT - 1.  The row a lane requests at step t is 32 t + 8 (block of its wave) +
    // (lane >> 3) - never clamped below the last tile - and the swizzle key only sees row bits 1-3, so pointer(t) = pointer(0) +
    // 32 t rows: one 64-bit add and one SALU add per request instead of the ~20 VALU (64-bit multiply, min, swizzle) hipcc emitted
    // for each call of the two lambdas above, in a loop that is bound by instruction issue.  The last tile (clamped rows) keeps
    // the generic form, outside the loop.
    const bf16_t* pq = nullptr;
    const bf16_t* pk = nullptr;
    unsigned dq_dst = 0, dk_dst = 0;
    if (PF && prefetch_next) {
        const int rq = (w & 3) * 8 + (lane >> 3), rk = (w & 1) * 32 + (w >> 1) * 8 + (lane >> 3);
        pq = nsrc + (long)rq * nld + ((lane & 7) ^ swz_key(rq)) * 8;
        pk = nk_src + (long)rk * ld + ((lane & 7) ^ swz_key(rk)) * 8;
        dq_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)((w < 4 ? Qt : Dt) + (w & 3) * 1024));
        dk_dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)(Kt + ((w & 1) * 4 + (w >> 1)) * 1024));
    }
    const long pq_step = 32 * nld, pk_step = 64 * ld;
    // which steps request: bit t of a mask that is shifted along (a loop-CARRIED condition: written as tests of qt and of the wave's
    // parity hipcc unswitched / peeled the tile loop into three copies and spilled 31 registers).  Q / dO tile qt - 1 at steps 1 ..
    // NT - 1; K blocks 4 qt .. 4 qt + 3 at steps qt < NT - 1 of the wave's parity
    unsigned qmask = (PF && prefetch_next) ? ((1u << NT) - 2u) : 0u;
    unsigned kmask = (PF && prefetch_next) ? (((wave_u & 1) ? 0xAAAAAAAAu : 0x55555555u) & ((1u << (NT - 1)) - 1u)) : 0u;
    asm volatile("" : "+s"(qmask), "+s"(kmask));
    for (int qt = 0; qt < NT; ++qt) {
        f32x16 s = zero16(), dp = zero16();
        // Round 5: the tile's 32 lse / D values (8 x 16 B per lane, two distinct addresses per instruction: broadcasts) are
        // requested HERE, in front of the S / dP MFMAs.  hipcc placed each of the eight reads directly in front of its four
        // exp / dS evaluations, followed by its own s_waitcnt lgkmcnt(0): eight exposed LDS round trips (~100+ cycles each
        // with eight waves on the LDS) in the middle of the softmax arithmetic of every tile and wave - a third of the
        // 3.0 k cycles a tile took?  Measured (profiles/r05_attn_ab_bwd_variants_fwd_persistent.log, same box, three
        // alternations): tile loop 27.3 k -> 25.5 k cycles per head (-6 %), kernel 199-203 -> 191-199 us.  The sched_barrier pins
        // them; the MFMA phase (>= 256 cycles) covers their latency, and the 32 registers are free at this point of the step
        // (dO^T / Q^T / dS fragments are not live yet).  Also requesting the step's eight row-major Q / dO fragments in one batch
        // behind them measured no better (196-201 us): removed.
        float4 lq4[4], dq4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            lq4[g] = *(const float4*)(Ls + qt * 32 + 8 * g + 4 * hi);
            dq4[g] = *(const float4*)(Ds + qt * 32 + 8 * g + 4 * hi);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            s = MFMA(frag_rm(Qt, qt * 32, fo.rm[kk]), kf[kk], s);      // S[q][key]: lane <-> key, regs <-> q
            dp = MFMA(frag_rm(Dt, qt * 32, fo.rm[kk]), vf[kk], dp);    // dP[q][key]
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 lq = lq4[g], dq = dq4[g];
            const float lqa[4] = {lq.x, lq.y, lq.z, lq.w};
            const float dqa[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = EXP2(fmaf(s[g * 4 + e], scale_log2, -lqa[e]));
                s[g * 4 + e] = pv;                                  // P   (in place)
                dp[g * 4 + e] = pv * (dp[g * 4 + e] - dqa[e]);      // dS  (in place)
            }
        }
        lds_char* buf = dsb + (qt & 1) * (NK * FB2_TILE);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)dp[g * 4 + e];
            *(__attribute__((address_space(3))) bf16x4*)(buf + st_w + (((2 * g + hi) ^ st_sw) << 3)) = v4;
        }
        // operands of dV / dK (dO^T, Q^T of this query tile): independent of the other waves.  (Requesting them ahead of
        // the softmax arithmetic measured 1 % slower: 256 VGPRs and a spill.)
        bf16x8 dot[2][2], qtr[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dot[ks][dt] = frag_tr(Dt, qt * 32 + ks * 16, fo, dt);
                qtr[ks][dt] = frag_tr(Qt, qt * 32 + ks * 16, fo, dt);
            }
        if (PF) {
            if (qmask & 1u) {                                       // Q / dO tile qt - 1 (< NT - 1)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(pq), "s"(dq_dst) : "memory", "m0");
                pq += pq_step; dq_dst += 4096;
            }
            if (kmask & 1u) {                                       // K blocks 4 qt .. 4 qt + 3: the waves of parity qt
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(pk), "s"(dk_dst) : "memory", "m0");
                pk += pk_step; dk_dst += 8192;
            }
            qmask >>= 1; kmask >>= 1;
        } else
        if (prefetch_next && qt > 0) prefetch_tile(qt - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the dS tile is in LDS
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 dsq[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(buf + j * FB2_TILE + rd[r]));
                dsq[j][4 * r + 0] = v[0]; dsq[j][4 * r + 1] = v[1]; dsq[j][4 * r + 2] = v[2]; dsq[j][4 * r + 3] = v[3];
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 pb = pack_b(s, ks), dbv = pack_b(dp, ks);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = MFMA(dot[ks][dt], pb, dv[dt]);
                dk[dt] = MFMA(qtr[ks][dt], dbv, dk[dt]);
            }
        }
        f32x4 dq0 = {0.0f, 0.0f, 0.0f, 0.0f}, dq1 = {0.0f, 0.0f, 0.0f, 0.0f};   // dQ^T block: lane <-> q, regs <-> d
#pragma unroll
        for (int j = 0; j < NK; j += 2) {
            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kq[j], dsq[j], dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kq[j + 1], dsq[j + 1], dq1, 0, 0, 0);
        }
        {
            const int q = qt * 32 + 16 * qb + i16;
            const float de = De[q];                                 // dS[q][odd key]: its rank-1 term dS * k joins here
            if (q < S) {
                bf16x4 ov;
                ov[0] = (bf16_t)(fmaf(de, ke4.x, dq0[0] + dq1[0]) * scale); ov[1] = (bf16_t)(fmaf(de, ke4.y, dq0[1] + dq1[1]) * scale);
                ov[2] = (bf16_t)(fmaf(de, ke4.z, dq0[2] + dq1[2]) * scale); ov[3] = (bf16_t)(fmaf(de, ke4.w, dq0[3] + dq1[3]) * scale);
                *(bf16x4*)(dq_out + (long)q * lddq) = ov;
            }
        }
    }

    if (prefetch_next) prefetch_tile(NT - 1);
    if (PF && prefetch_next) prefetch_k(NT - 1);                    // (Sp / 8 = 4 NT blocks: steps 0 .. NT - 1 cover the K tile)
    if (PF) {
        // every wave is done with the dS tiles (the store staging below aliases them) and with Ls / Ds / De (the next head's
        // phase 0 rewrites them); then the next head's register operands go out, to land under the stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (unconditional: a conditional definition would keep the PREVIOUS values live through the whole head, tile loop included;
        // a workgroup's last head re-reads its own operands and drops them)
        fetch_regs(prefetch_next ? nxt : bh, tid);
    }
    stamp(3);
    // ---- phase 3: dK, dV of this wave's keys; the odd key ------------------------------------------------------
    // Through a wave-private 4 KiB LDS tile ([32 keys][64 d] bf16, 16-B chunk index XOR (key & 7)), so that the global
    // stores are 16 B per lane and 8 whole 128-B rows per instruction instead of row-per-lane 8-byte stores (4 + 4
    // dwordx4 stores instead of 16 + 16 dwordx2; the store tail was 10 % of the kernel, issue-bound).  The tile lives
    // behind the dS buffers, which slower waves may still be reading.
    {
        lds_char* tile = (lds_char*)dsarea + (PF ? 0 : 2 * NK * FB2_TILE) + w * 4096;
        const int r8 = lane >> 3, c8 = lane & 7;
        bf16_t* kbase = dqkv + (long)b * lay.qkv_b + (long)h * lay.qkv_h + (long)(w * 32) * lddq + W + c8 * 8;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ov[e] = which == 0 ? (bf16_t)(dk[dt][g * 4 + e] * scale) : (bf16_t)dv[dt][g * 4 + e];
                    *(__attribute__((address_space(3))) bf16x4*)(tile + l31 * 128 + (((dt * 4 + g) ^ (l31 & 7)) << 4) + hi * 8) = ov;
                }
            bf16x8 t[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + r8;
                t[it] = *(const __attribute__((address_space(3))) bf16x8*)(tile + row * 128 + ((c8 ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int it = 0; it < 4; ++it)
                *(bf16x8*)(kbase + (long)(it * 8 + r8) * lddq + which * W) = t[it];
        }
    }
    if (w == 0) {
        float ak = 0.0f, av = 0.0f;
#pragma unroll
        for (int ww = 0; ww <= NK; ++ww) { ak += KVe[(ww * 2 + 0) * 64 + lane]; av += KVe[(ww * 2 + 1) * 64 + lane]; }
        bf16_t* krow = dqkv + (long)b * lay.qkv_b + (long)h * lay.qkv_h + (long)SE * lddq + W;
        krow[lane] = (bf16_t)(ak * scale);
        krow[W + lane] = (bf16_t)av;
    }
    stamp(4);
    have_qd = prefetch_next;
    // the next head's staging overwrites the K / V area (dS tiles, store transposes) and the small arrays.  (PF: nothing is staged
    // there any more - the small arrays are rewritten behind the barrier in front of phase 3, KVe behind the next phase-0 barrier,
    // which wave 0 reaches after its reads below... above: the odd key's row is summed by wave 0 before it leaves this phase.)
    if (!PF) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
static bool g_use_tr = true;
void attn_set_use_tr(int on) { g_use_tr = on != 0; }

static int attn_block_threads(int ntiles) {
    // one 32-row tile per wave and round, up to 10 waves: the fewest rounds, then the fewest waves that give them
    // (S = 257 -> 9 tiles -> 9 waves; S = 577 -> 19 tiles -> 2 rounds of 10 waves - with 9 waves it took 3 rounds, 27 wave
    // slots for 19 tiles of work; 10 waves still are at most 3 per SIMD, the register budget of 9)
    const int rounds = (ntiles + 9) / 10;
    return ((ntiles + rounds - 1) / rounds) * 64;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(RVLM_ERR_UNSUPPORTED, "attention: sequence too long for LDS-resident K/V");
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    return RVLM_OK;
}

int attn_fwd_bf16(const bf16_t* qkv, long ldqkv, bf16_t* o, long ldo, float* lse, int B, int H,
                  int S, hipStream_t s) {
    const int Sp = (int)round_up(S, 32), W = H * 64;
    const size_t lds_bytes = (size_t)Sp * 256;
    const float sl2 = 0.125f * 1.4426950408889634f;
    const int nt = attn_block_threads(Sp / 32);
    int rc;
    // MINW = 5 caps the kernel at 96 VGPRs so that two 9-wave workgroups (2 x 74 KiB LDS) share a CU:
    // one workgroup's K/V staging then hides under the other's MFMA/softmax phase.
    static int occ = -1, odd = -1;
    if (occ < 0) { const char* e = getenv("RVLM_ATTN_OCC"); occ = e ? atoi(e) : 5; }
    if (odd < 0) { const char* e = getenv("RVLM_ATTN_FWD_ODD"); odd = e ? atoi(e) : 1; }
    if (odd && g_use_tr && S == 257) {   // 8 waves, no padded tiles (see attn_fwd_odd_kernel)
        constexpr int NK = 8;
        const size_t lds_o = (size_t)(32 * NK + 32) * 256 + (size_t)NK * 66 * sizeof(float) + (size_t)NK * 64;
#ifdef RVLM_EXPERIMENTAL_GEMM
        static int fpers = -1;
        if (fpers < 0) { const char* e = getenv("RVLM_ATTN_FWD_PERSIST"); fpers = e ? atoi(e) : 0; }
        if (fpers) {      // persistent workgroups, two K / V slots (see attn_fwd_odd_pers_kernel)
            const size_t lds_p = 2 * (size_t)(32 * NK + 32) * 256 + (size_t)NK * 66 * sizeof(float) + (size_t)NK * 64;
            if ((rc = set_lds(attn_fwd_odd_pers_kernel<NK>, lds_p))) return rc;
            const AttnLayout lay = {(long)S * ldqkv, 64L, (long)S * ldo, 64L};
            const int nbh = B * H;
            hipLaunchKernelGGL((attn_fwd_odd_pers_kernel<NK>), dim3(std::min(nbh, 256)), dim3(NK * 64), lds_p, s, qkv, ldqkv, o,
                               ldo, lse, H, (long)W, sl2, lay, nbh);
            RVLM_CHECK_LAUNCH();
            return RVLM_OK;
        }
#endif
        if ((rc = set_lds(attn_fwd_odd_kernel<NK>, lds_o))) return rc;
        static int hm = -1;      // timing probe: read the same buffers as head-blocked [3][B*H][S][64] / [B*H][S][64]
        if (hm < 0) { const char* e = getenv("RVLM_ATTN_HM"); hm = e ? atoi(e) : 0; }
        if (hm) {
            const AttnLayout lay = {(long)H * S * 64, (long)S * 64, (long)H * S * 64, (long)S * 64};
            hipLaunchKernelGGL((attn_fwd_odd_kernel<NK>), dim3(B * H), dim3(NK * 64), lds_o, s, qkv, 64L, o, 64L, lse, H,
                               (long)B * H * S * 64, sl2, lay);
        } else {
            const AttnLayout lay = {(long)S * ldqkv, 64L, (long)S * ldo, 64L};
            hipLaunchKernelGGL((attn_fwd_odd_kernel<NK>), dim3(B * H), dim3(NK * 64), lds_o, s, qkv, ldqkv, o, ldo, lse, H,
                               (long)W, sl2, lay);
        }
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
#define LAUNCH_FWD(TR, MW)                                                                                   \
    do {                                                                                                      \
        if ((rc = set_lds(attn_fwd_kernel<TR, MW>, lds_bytes))) return rc;                                   \
        hipLaunchKernelGGL((attn_fwd_kernel<TR, MW>), dim3(B * H), dim3(nt), lds_bytes, s, qkv, ldqkv, o,    \
                           ldo, lse, H, S, Sp, W, sl2);                                                      \
    } while (0)
    // (two workgroups share a CU only if their K / V tiles fit twice: at S = 577 the 96-register cap would buy nothing)
    if (g_use_tr) { if (occ == 5 && 2 * lds_bytes <= 160 * 1024) LAUNCH_FWD(true, 5); else LAUNCH_FWD(true, 1); }
    else LAUNCH_FWD(false, 1);
#undef LAUNCH_FWD
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

int attn_bwd_bf16(const bf16_t* qkv, long ldqkv, const bf16_t* o, long ldo, const bf16_t* d_o,
                  long lddo, const float* lse, float* dsum_scratch, bf16_t* dqkv, long lddqkv,
                  int B, int H, int S, hipStream_t s) {
    const int Sp = (int)round_up(S, 32), W = H * 64;
    const float scale = 0.125f, sl2 = 0.125f * 1.4426950408889634f;
    const int nt = attn_block_threads(Sp / 32);
    const size_t lds_q = (size_t)Sp * 256;
    const bool kv_lds = ((size_t)Sp * 520 <= 160 * 1024);     // Q, dO, K, V all LDS-resident (S <= 315)
    const size_t lds_kv = kv_lds ? (size_t)Sp * 520 : (size_t)Sp * 264;
    int rc;
    static int fused = -1;
    if (fused < 0) { const char* e = getenv("RVLM_ATTN_FUSED"); fused = e ? atoi(e) : 1; }
    if (fused && g_use_tr && S == 257) {   // one kernel: dQ, dK, dV (see attn_bwd_fused_kernel)
        constexpr int NK = 8, SpF = (NK + 1) * 32;
        // RVLM_ATTN_BWD_PF=1 (make EXPERIMENTAL=1 builds only; measured SLOWER, profiles/r06_attn_bwd_prefetch_ab.log): K / V / O /
        // lse of the next head arrive under the current one (template parameter PF above)
#ifdef RVLM_EXPERIMENTAL_GEMM
        static int pf = -1;
        if (pf < 0) { const char* e = getenv("RVLM_ATTN_BWD_PF"); pf = e ? atoi(e) : 0; }
#else
        constexpr int pf = 0;
#endif
        const size_t smalls = (4 * SpF + 64 + (NK + 1) * 128) * sizeof(float) + 4 * 128;
        const size_t lds_f = pf ? (size_t)SpF * 384 + 2 * NK * FB2_TILE + smalls : (size_t)SpF * 512 + smalls;
#ifdef RVLM_EXPERIMENTAL_GEMM
        if ((rc = pf ? set_lds(attn_bwd_fused_kernel<NK, true>, lds_f) : set_lds(attn_bwd_fused_kernel<NK, false>, lds_f))) return rc;
#else
        if ((rc = set_lds(attn_bwd_fused_kernel<NK, false>, lds_f))) return rc;
#endif
        static int trace = -1, desync = -1;
        if (trace < 0) { const char* e = getenv("RVLM_ATTN_TRACE"); trace = e ? atoi(e) : 0; }
        if (desync < 0) { const char* e = getenv("RVLM_ATTN_DESYNC"); desync = e ? atoi(e) : 0; }
        static int hm = -1, persist = -1;
        if (hm < 0) { const char* e = getenv("RVLM_ATTN_HM"); hm = e ? atoi(e) : 0; }
        if (persist < 0) { const char* e = getenv("RVLM_ATTN_PERSIST"); persist = e ? atoi(e) : 1; }
        const int nbh = B * H, grid = persist ? std::min(nbh, 256) : nbh;
#ifdef RVLM_EXPERIMENTAL_GEMM
#define LAUNCH_FUSED(PFV, ...) hipLaunchKernelGGL((attn_bwd_fused_kernel<NK, PFV>), dim3(grid), dim3(NK * 64), lds_f, s, __VA_ARGS__)
#else       // (the shipped library instantiates the measured form only)
#define LAUNCH_FUSED(PFV, ...) hipLaunchKernelGGL((attn_bwd_fused_kernel<NK, false>), dim3(grid), dim3(NK * 64), lds_f, s, __VA_ARGS__)
#endif
        unsigned long long* tr = trace ? (unsigned long long*)dsum_scratch : nullptr;
        if (hm) {
            const AttnLayout lay = {(long)H * S * 64, (long)S * 64, (long)H * S * 64, (long)S * 64};
            if (pf) LAUNCH_FUSED(true, qkv, 64L, o, 64L, d_o, 64L, lse, dqkv, 64L, H, S, (long)B * H * S * 64, scale, sl2, tr, desync | (trace << 8), lay, nbh);
            else LAUNCH_FUSED(false, qkv, 64L, o, 64L, d_o, 64L, lse, dqkv, 64L, H, S, (long)B * H * S * 64, scale, sl2, tr, desync | (trace << 8), lay, nbh);
        } else {
            const AttnLayout lay = {(long)S * ldqkv, 64L, (long)S * ldo, 64L};
            if (pf) LAUNCH_FUSED(true, qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv, lddqkv, H, S, (long)W, scale, sl2, tr, desync | (trace << 8), lay, nbh);
            else LAUNCH_FUSED(false, qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv, lddqkv, H, S, (long)W, scale, sl2, tr, desync | (trace << 8), lay, nbh);
        }
#undef LAUNCH_FUSED
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
#define LAUNCH_BWD(TR)                                                                                         \
    do {                                                                                                        \
        if ((rc = set_lds(attn_bwd_dq_kernel<TR>, lds_q))) return rc;                                          \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<TR>), dim3(B * H), dim3(nt), lds_q, s, qkv, ldqkv, o, ldo, d_o, \
                           lddo, lse, dsum_scratch, dqkv, lddqkv, H, S, Sp, W, scale, sl2);                    \
        RVLM_CHECK_LAUNCH();                                                                                    \
        if (kv_lds) {                                                                                           \
            if ((rc = set_lds(attn_bwd_dkv_kernel<TR, true>, lds_kv))) return rc;                              \
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<TR, true>), dim3(B * H), dim3(nt), lds_kv, s, qkv, ldqkv,  \
                               d_o, lddo, lse, dsum_scratch, dqkv, lddqkv, H, S, Sp, W, scale, sl2);           \
        } else {                                                                                                \
            if ((rc = set_lds(attn_bwd_dkv_kernel<TR, false>, lds_kv))) return rc;                             \
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<TR, false>), dim3(B * H), dim3(nt), lds_kv, s, qkv, ldqkv, \
                               d_o, lddo, lse, dsum_scratch, dqkv, lddqkv, H, S, Sp, W, scale, sl2);           \
        }                                                                                                       \
    } while (0)
    if (g_use_tr) LAUNCH_BWD(true); else LAUNCH_BWD(false);
#undef LAUNCH_BWD
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Class-token attention of the LAST transformer block.  The encoder's output is ln_post(x[:, 0]) @ proj
// (open_clip VisionTransformer.forward with pool_type 'tok'), so in the last block only the class-token query
// contributes: every other query row of that block (and everything downstream of it) is dead.  One wave per
// (image, head): q is the class token's query, the keys / values are all S tokens.  fp32 softmax, natural-log lse.
//   fwd: o[b, h*64 + d] = sum_j softmax_j(scale * q.k_j) v_j[d]
//   bwd: dV_j = p_j dO, dK_j = dS_j q, dQ_0 = sum_j dS_j k_j with dS_j = scale * p_j (dO.v_j - dO.o); dQ_j = 0 (j > 0)
// HBM-bound: K and V of the block are read once (fwd) / K twice, V once (bwd); dqkv is written in full.
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16_t x = (bf16_t)a, y = (bf16_t)b;
    return (unsigned)*(const unsigned short*)&x | ((unsigned)*(const unsigned short*)&y << 16);
}
__device__ __forceinline__ float dot8(const float (&a)[8], const float (&b)[8]) {
    float acc = a[0] * b[0];
#pragma unroll
    for (int e = 1; e < 8; ++e) acc = fmaf(a[e], b[e], acc);
    return acc;
}
// sum over the 8 lanes that share lane >> 3 (the 8 sixteen-byte chunks of one 64-wide head row)
__device__ __forceinline__ float chunk_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}
// Lane (g, c) = (lane >> 3, lane & 7): key j = g + 8 i, 16-byte chunk c of its 128-byte head row, so that one wave load
// covers 8 whole rows.  Dot products are reduced over c, the per-group running results over g at the end.

__global__ void __launch_bounds__(256)
attn_cls_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, bf16_t* __restrict__ o, long ldo, float* __restrict__ lse,
                    int H, int S, int W, int total, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= total) return;
    const int b = bh / H, hd = bh - b * H, g = lane >> 3, c = lane & 7;
    const bf16_t* base = qkv + (long)b * S * ld + hd * 64 + c * 8;
    float qc[8];
    unpack8(*(const uint4*)base, qc);
#pragma unroll
    for (int e = 0; e < 8; ++e) qc[e] *= scale;
    float m = -INFINITY, l = 0.0f, acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    const int iters = (S + 7) >> 3;
    for (int i = 0; i < iters; ++i) {
        const int j = g + 8 * i;
        const bool valid = j < S;
        const bf16_t* row = base + (long)(valid ? j : S - 1) * ld;
        float kf[8], vf[8];
        unpack8(*(const uint4*)(row + W), kf);
        unpack8(*(const uint4*)(row + 2 * W), vf);
        const float sc = chunk_sum(dot8(kf, qc));
        if (valid) {
            const float mn = fmaxf(m, sc);
            const float corr = __expf(m - mn), pj = __expf(sc - mn);
            l = fmaf(l, corr, pj);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(acc[e], corr, pj * vf[e]);
            m = mn;
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off), l2 = __shfl_xor(l, off);
        const float mn = fmaxf(m, m2);
        const float ca = (m == -INFINITY) ? 0.0f : __expf(m - mn), cb = (m2 == -INFINITY) ? 0.0f : __expf(m2 - mn);
        l = l * ca + l2 * cb;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * ca + __shfl_xor(acc[e], off) * cb;
        m = mn;
    }
    if (g == 0) {
        const float inv = 1.0f / l;
        uint4 ov;
        ov.x = pack2(acc[0] * inv, acc[1] * inv); ov.y = pack2(acc[2] * inv, acc[3] * inv);
        ov.z = pack2(acc[4] * inv, acc[5] * inv); ov.w = pack2(acc[6] * inv, acc[7] * inv);
        *(uint4*)(o + (long)b * ldo + hd * 64 + c * 8) = ov;
        if (c == 0) lse[bh] = m + __logf(l);
    }
}

__global__ void __launch_bounds__(256)
attn_cls_bwd_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ o, long ldo,
                    const bf16_t* __restrict__ d_o, long lddo, const float* __restrict__ lse,
                    bf16_t* __restrict__ dqkv, long lddq, int H, int S, int W, int total, float scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= total) return;
    const int b = bh / H, hd = bh - b * H, g = lane >> 3, c = lane & 7;
    const bf16_t* base = qkv + (long)b * S * ld + hd * 64 + c * 8;
    bf16_t* dbase = dqkv + (long)b * S * lddq + hd * 64 + c * 8;
    float qc[8], doc[8], oc[8];
    unpack8(*(const uint4*)base, qc);
    unpack8(*(const uint4*)(d_o + (long)b * lddo + hd * 64 + c * 8), doc);
    unpack8(*(const uint4*)(o + (long)b * ldo + hd * 64 + c * 8), oc);
    const float Dsum = chunk_sum(dot8(doc, oc));
    const float Lse = lse[bh];
    float accq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) accq[e] = 0.0f;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    const int iters = (S + 7) >> 3;
    for (int i = 0; i < iters; ++i) {
        const int j = g + 8 * i;
        const bool valid = j < S;
        const long rj = valid ? j : S - 1;
        float kf[8], vf[8];
        unpack8(*(const uint4*)(base + rj * ld + W), kf);
        unpack8(*(const uint4*)(base + rj * ld + 2 * W), vf);
        const float sc = chunk_sum(dot8(kf, qc));
        const float dp = chunk_sum(dot8(vf, doc));
        const float pj = valid ? __expf(scale * sc - Lse) : 0.0f;
        const float dsj = pj * (dp - Dsum) * scale;
#pragma unroll
        for (int e = 0; e < 8; ++e) accq[e] = fmaf(dsj, kf[e], accq[e]);
        if (valid) {
            uint4 kk, vv;
            kk.x = pack2(dsj * qc[0], dsj * qc[1]); kk.y = pack2(dsj * qc[2], dsj * qc[3]);
            kk.z = pack2(dsj * qc[4], dsj * qc[5]); kk.w = pack2(dsj * qc[6], dsj * qc[7]);
            vv.x = pack2(pj * doc[0], pj * doc[1]); vv.y = pack2(pj * doc[2], pj * doc[3]);
            vv.z = pack2(pj * doc[4], pj * doc[5]); vv.w = pack2(pj * doc[6], pj * doc[7]);
            bf16_t* drow = dbase + rj * lddq;
            *(uint4*)(drow + W) = kk;
            *(uint4*)(drow + 2 * W) = vv;
            if (j > 0) *(uint4*)drow = zero;
        }
    }
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) accq[e] += __shfl_xor(accq[e], off);
    if (g == 0) {
        uint4 qq;
        qq.x = pack2(accq[0], accq[1]); qq.y = pack2(accq[2], accq[3]);
        qq.z = pack2(accq[4], accq[5]); qq.w = pack2(accq[6], accq[7]);
        *(uint4*)dbase = qq;
    }
}

int attn_cls_fwd_bf16(const bf16_t* qkv, long ldqkv, bf16_t* o, long ldo, float* lse, int B, int H, int S,
                      hipStream_t s) {
    if (!qkv || !o || !lse || B <= 0 || H <= 0 || S <= 0 || ldqkv % 8 != 0 || ldo % 8 != 0)
        return fail(RVLM_ERR_ARG, "attn_cls_fwd_bf16: bad arguments");
    const int total = B * H;
    hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3(cdiv(total, 4)), dim3(256), 0, s, qkv, ldqkv, o, ldo, lse, H, S, H * 64,
                       total, 0.125f);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
int attn_cls_bwd_bf16(const bf16_t* qkv, long ldqkv, const bf16_t* o, long ldo, const bf16_t* d_o, long lddo,
                      const float* lse, bf16_t* dqkv, long lddqkv, int B, int H, int S, hipStream_t s) {
    if (!qkv || !o || !d_o || !lse || !dqkv || B <= 0 || H <= 0 || S <= 0 || ldqkv % 8 != 0 || lddqkv % 8 != 0 ||
        ldo % 8 != 0 || lddo % 8 != 0)
        return fail(RVLM_ERR_ARG, "attn_cls_bwd_bf16: bad arguments");
    const int total = B * H;
    hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3(cdiv(total, 4)), dim3(256), 0, s, qkv, ldqkv, o, ldo, d_o, lddo, lse, dqkv,
                       lddqkv, H, S, H * 64, total, 0.125f);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// occupancy (workgroups per CU) the runtime reports for the three kernels at sequence length S
int attn_occupancy(int S, int* out3) {
    const int Sp = (int)round_up(S, 32);
    const int nt = attn_block_threads(Sp / 32);
    const size_t lds_f = (size_t)Sp * 256, lds_kv = ((size_t)Sp * 520 <= 160 * 1024) ? (size_t)Sp * 520 : (size_t)Sp * 264;
    int rc;
    if ((rc = set_lds(attn_fwd_kernel<true, 5>, lds_f))) return rc;
    if ((rc = set_lds(attn_fwd_kernel<true, 1>, lds_f))) return rc;
    if ((rc = set_lds(attn_bwd_dq_kernel<true>, lds_f))) return rc;
    RVLM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[0], attn_fwd_kernel<true, 5>, nt, lds_f));
    RVLM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[1], attn_fwd_kernel<true, 1>, nt, lds_f));
    RVLM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[2], attn_bwd_dq_kernel<true>, nt, lds_f));
    (void)lds_kv;
    return RVLM_OK;
}

}  // namespace rvlm
