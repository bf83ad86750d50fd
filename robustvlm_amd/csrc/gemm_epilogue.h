// Shared epilogue of the bf16 MFMA GEMM kernels.
//
// Accumulator layout (operand-swapped 32x32x16 MFMA): for sub-tile (mi, ni) lane holds, for g = 0..3,
// the 4 consecutive columns n = n_base + ni*32 + 8*g + 4*hi + {0..3} of row m = m_base + mi*32 + (lane&31).
//
// Full tiles (the common case) go through a wave-private LDS transpose so that every global access of
// the epilogue is a 16-byte-per-lane, full-128-byte-line access (8 or 16 lanes per output row):
// measured on MI355X, the direct 8-byte-per-lane row-strided stores of a 32 896 x 3072 bf16 output cost
// 55-80 us (2.5-3.6 TB/s, transaction-bound) and did not overlap with the partner workgroup's MFMAs.
// Residual (fp32) and h_pre (bf16) are read in the same coalesced pattern after the transpose.
// Edge tiles (M or N not a multiple of the wave tile) take the predicated per-lane path.
#pragma once
#include "kernels.h"

namespace rvlm {

constexpr int EPI_LDS_BYTES_PER_WAVE = 32 * 272;   // 32 rows x (64 fp32 + 16 B pad)

// ---- predicated per-lane path (edge tiles) ------------------------------------------------------
template <int EPI, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue_edge(const f32x16 (&acc)[MI][NI], const GemmBf16& p, int m_base,
                                                   int n_base, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m_base + mi * 32 + l31;
        if (m >= p.M) continue;
        const long rowoff = (long)m * p.ldo;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_base + ni * 32 + 8 * g + 4 * hi;
                if (n >= p.N) continue;
                const long o = rowoff + n;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e] + (p.bias ? p.bias[n + e] : 0.0f);
                if (EPI == EPI_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ((bf16_t*)p.out)[o + e] = (bf16_t)v[e];
                } else if (EPI == EPI_F32_RESID) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ((float*)p.out)[o + e] = v[e] + (p.residual ? p.residual[o + e] : 0.0f);
                } else if (EPI == EPI_BF16_ACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float av, dv;
                        act_pair(v[e], p.act, av, dv);
                        if (p.out_pre) p.out_pre[o + e] = (bf16_t)dv;
                        ((bf16_t*)p.out)[o + e] = (bf16_t)av;
                    }
                } else if (EPI == EPI_BF16_DACT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ((bf16_t*)p.out)[o + e] = (bf16_t)(v[e] * (float)p.h_pre[o + e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ((float*)p.out)[o + e] = v[e];
                }
            }
    }
}

// ---- LDS-transposed path (full tiles) -------------------------------------------------------------
// stage one 32 x 64 block of fp32 values: row stride 272 B
__device__ __forceinline__ void stage_f32(char* buf, const float (&v)[2][4][4], int l31, int hi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(buf + l31 * 272 + (ni * 32 + 8 * g + 4 * hi) * 4) =
                make_float4(v[ni][g][0], v[ni][g][1], v[ni][g][2], v[ni][g][3]);
}
// stage one 32 x 64 block as bf16: row stride 144 B
__device__ __forceinline__ void stage_bf16(char* buf, const float (&v)[2][4][4], int l31, int hi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[ni][g][e];
            *(bf16x4*)(buf + l31 * 144 + (ni * 32 + 8 * g + 4 * hi) * 2) = o;
        }
}
// write the staged bf16 block to global: 8 lanes x 16 B per row, 8 rows per wave instruction
__device__ __forceinline__ void flush_bf16(const char* buf, bf16_t* dst, long ldo, int lane) {
    uint4 t[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) t[it] = *(const uint4*)(buf + (it * 8 + (lane >> 3)) * 144 + (lane & 7) * 16);
#pragma unroll
    for (int it = 0; it < 4; ++it) *(uint4*)(dst + (long)(it * 8 + (lane >> 3)) * ldo + (lane & 7) * 8) = t[it];
}

template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_full(const f32x16 (&acc)[MI][2], const GemmBf16& p, int m_base,
                                                   int n_base, int lane, char* buf) {
    const int l31 = lane & 31, hi = lane >> 5;
    float4 bv[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[ni][g] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv[ni][g] = *(const float4*)(p.bias + n_base + ni * 32 + 8 * g + 4 * hi);
        }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        float v[2][4][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v[ni][g][0] = acc[mi][ni][g * 4 + 0] + bv[ni][g].x;
                v[ni][g][1] = acc[mi][ni][g * 4 + 1] + bv[ni][g].y;
                v[ni][g][2] = acc[mi][ni][g * 4 + 2] + bv[ni][g].z;
                v[ni][g][3] = acc[mi][ni][g * 4 + 3] + bv[ni][g].w;
            }
        const long row0 = (long)(m_base + mi * 32) * p.ldo + n_base;
        if (EPI == EPI_BF16) {
            stage_bf16(buf, v, l31, hi);
            flush_bf16(buf, (bf16_t*)p.out + row0, p.ldo, lane);
        } else if (EPI == EPI_BF16_ACT) {
            float d[2][4][4];   // act'(h) -> out_pre, act(h) -> out
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) act_pair(v[ni][g][e], p.act, v[ni][g][e], d[ni][g][e]);
            if (p.out_pre) {
                stage_bf16(buf, d, l31, hi);
                flush_bf16(buf, p.out_pre + row0, p.ldo, lane);
            }
            stage_bf16(buf, v, l31, hi);
            flush_bf16(buf, (bf16_t*)p.out + row0, p.ldo, lane);
        } else {
            // fp32 staging: 16 lanes x 16 B per row (64 fp32), 4 rows per wave instruction
            stage_f32(buf, v, l31, hi);
            const int r4 = lane >> 4, c16 = lane & 15;
            if (EPI == EPI_BF16_DACT) {
                // out bf16 = v * act'(h): 8 values per lane -> pair two staged float4 per lane
                // lane handles row it*8 + (lane>>3), cols (lane&7)*8 .. +7
                uint4 hv[4];
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    hv[it] = *(const uint4*)(p.h_pre + row0 + (long)(it * 8 + (lane >> 3)) * p.ldo + (lane & 7) * 8);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const char* src = buf + (it * 8 + (lane >> 3)) * 272 + (lane & 7) * 32;
                    const float4 a = *(const float4*)src, b = *(const float4*)(src + 16);
                    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    const bf16x8 h8 = *(const bf16x8*)&hv[it];
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(f[e] * (float)h8[e]);
                    *(bf16x8*)((bf16_t*)p.out + row0 + (long)(it * 8 + (lane >> 3)) * p.ldo + (lane & 7) * 8) = o;
                }
            } else {
                float4 rv[8];
                if (EPI == EPI_F32_RESID && p.residual) {
#pragma unroll
                    for (int it = 0; it < 8; ++it)
                        rv[it] = *(const float4*)(p.residual + row0 + (long)(it * 4 + r4) * p.ldo + c16 * 4);
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    float4 a = *(const float4*)(buf + (it * 4 + r4) * 272 + c16 * 16);
                    if (EPI == EPI_F32_RESID && p.residual) {
                        a.x += rv[it].x; a.y += rv[it].y; a.z += rv[it].z; a.w += rv[it].w;
                    }
                    *(float4*)((float*)p.out + row0 + (long)(it * 4 + r4) * p.ldo + c16 * 4) = a;
                }
            }
        }
    }
}

// buf: wave-private LDS region of EPI_LDS_BYTES_PER_WAVE bytes; the caller has already synchronised
// the workgroup after its last read of the pipeline LDS.
template <int EPI, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[MI][NI], const GemmBf16& p, int m_base,
                                              int n_base, int lane, char* buf) {
    static_assert(NI == 2, "wave tile is 64 columns");
    const bool full = (m_base + MI * 32 <= p.M) && (n_base + NI * 32 <= p.N);   // wave-uniform
    if (full) gemm_epilogue_full<EPI, MI>(acc, p, m_base, n_base, lane, buf);
    else gemm_epilogue_edge<EPI, MI, NI>(acc, p, m_base, n_base, lane);
}

}  // namespace rvlm
