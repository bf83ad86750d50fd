// Remainder rows of the persistent 256x256 bf16 GEMM kernels, computed inside the same launch (shared by
// gemm_bf16_256p.hip and gemm_bf16_256x.hip).
#pragma once
#include "kernels.h"
#include "gemm_persist.h"
#include <type_traits>

namespace rvlm {

// ---- remainder rows inside the same launch --------------------------------------------------------------------
// The tile loop covers floor(M / 256) * 256 rows.  The <= 255 rows left (128 at the encoder's M = 128 x 257) used to
// take two more launches per GEMM (a split-K 128x128 kernel + its reduce, ~13 us of latency-bound work, 5 % of a PGD
// step).  Here every workgroup, once its tiles are done, computes one unit of the strip: 32 rows x (u x 32) columns,
// K split over its 8 waves (fragments straight from global / L2 into registers, no LDS pipeline), partial sums reduced
// through LDS - no data crosses a workgroup, so no device-scope fences, and the epilogue is the tile loop's.
template <int EPI, int ACT>
__device__ __forceinline__ void strip_tail(const GemmBf16& p, int m_done, int m_total, char* lds, int w, int lane) {
    constexpr int UMAX = 2;
    const int rem = m_total - m_done;
    const int col_groups = p.N >> 5;
    // 32-row groups; 16-row groups (the upper MFMA rows duplicate the lower ones) when that is what it takes to give
    // every workgroup a unit - the strip is bound by the bytes one CU can pull, not by MFMA work
    const int RG = (((rem + 31) >> 5) * col_groups < (int)gridDim.x) ? 16 : 32;
    const int row_groups = (rem + RG - 1) / RG;
    int u = (row_groups * col_groups + (int)gridDim.x - 1) / (int)gridDim.x;
    u = u < 1 ? 1 : (u > UMAX ? UMAX : u);
    const int units_per_row = (col_groups + u - 1) / u, total_units = row_groups * units_per_row;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kw = p.K >> 3;                       // K % 128 == 0: every wave gets a multiple of 16
    float* red = (float*)lds;                       // [8 waves][UMAX][64 lanes][16]
    for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
        const int rg = unit / units_per_row, cu = unit - rg * units_per_row;
        const int m0 = m_done + rg * RG, n0 = cu * u * 32;
        const int nsub = min(u, col_groups - cu * u);
        const bf16_t* ap = p.A + (long)min(m0 + (l31 & (RG - 1)), m_total - 1) * p.lda + w * kw + hi * 8;
        const bf16_t* bp = p.Bw + (long)(n0 + l31) * p.ldb + w * kw + hi * 8;
        f32x16 acc[UMAX];
#pragma unroll
        for (int j = 0; j < UMAX; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
        // 64-deep groups (kw is a multiple of 16, not of 64: the last group may be short)
        auto load_group = [&](int k, bf16x8 (&a)[4], auto& b, auto nbmax, int nb) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ko = (k + kk * 16 < kw) ? k + kk * 16 : k;
                a[kk] = *(const bf16x8*)(ap + ko);
#pragma unroll
                for (int j = 0; j < decltype(nbmax)::value; ++j)
                    if (j < nb) b[j][kk] = *(const bf16x8*)(bp + (long)j * 32 * p.ldb + ko);
            }
        };
        auto mfma_group = [&](int k, const bf16x8 (&a)[4], const auto& b, auto nbmax, int nb) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (k + kk * 16 < kw) {
#pragma unroll
                    for (int j = 0; j < decltype(nbmax)::value; ++j)
                        if (j < nb) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][kk], a[kk], acc[j], 0, 0, 0);
                }
            }
        };
        {
            // two register stages: the loads of the next group are in flight under this group's MFMAs (a serial
            // load -> MFMA loop exposes the full memory latency 8 times at K = 4096)
            bf16x8 a0[4], a1[4], b0[UMAX][4], b1[UMAX][4];
            const std::integral_constant<int, UMAX> all;
            load_group(0, a0, b0, all, nsub);
            for (int k = 0; k < kw; k += 128) {
                if (k + 64 < kw) load_group(k + 64, a1, b1, all, nsub);
                mfma_group(k, a0, b0, all, nsub);
                if (k + 128 < kw) load_group(k + 128, a0, b0, all, nsub);
                if (k + 64 < kw) mfma_group(k + 64, a1, b1, all, nsub);
            }
        }
        __syncthreads();      // LDS free (previous unit's reads / the last tile's staging)
#pragma unroll
        for (int j = 0; j < UMAX; ++j)
            if (j < nsub) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    *(float4*)(red + (((w * UMAX + j) * 64 + lane) * 16 + q4 * 4)) =
                        make_float4(acc[j][q4 * 4], acc[j][q4 * 4 + 1], acc[j][q4 * 4 + 2], acc[j][q4 * 4 + 3]);
            }
        __syncthreads();
        // wave w combines accumulator registers 2w, 2w+1 (two consecutive columns) of every sub-tile
        const int m = m0 + l31;
#pragma unroll
        for (int j = 0; j < UMAX; ++j) {
            if (j >= nsub) continue;
            float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float2 t = *(const float2*)(red + (((ww * UMAX + j) * 64 + lane) * 16 + 2 * w));
                v0 += t.x; v1 += t.y;
            }
            const int r = 2 * w;
            const int n = n0 + j * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (m >= m_total || l31 >= RG) continue;
            if (p.bias) { v0 += p.bias[n]; v1 += p.bias[n + 1]; }
            const long o = (long)m * p.ldo + n;
            if (EPI == EPI_BF16) {
                ((bf16_t*)p.out)[o] = (bf16_t)v0; ((bf16_t*)p.out)[o + 1] = (bf16_t)v1;
            } else if (EPI == EPI_F32_RESID) {
                ((float*)p.out)[o] = v0 + p.residual[o]; ((float*)p.out)[o + 1] = v1 + p.residual[o + 1];
            } else if (EPI == EPI_BF16_ACT) {
                float a0v, d0v, a1v, d1v;
                actp_pair<ACT>(v0, a0v, d0v); actp_pair<ACT>(v1, a1v, d1v);
                if (p.out_pre) { p.out_pre[o] = (bf16_t)d0v; p.out_pre[o + 1] = (bf16_t)d1v; }
                ((bf16_t*)p.out)[o] = (bf16_t)a0v; ((bf16_t*)p.out)[o + 1] = (bf16_t)a1v;
            } else if (EPI == EPI_BF16_DACT) {
                ((bf16_t*)p.out)[o] = (bf16_t)(v0 * (float)p.h_pre[o]);
                ((bf16_t*)p.out)[o + 1] = (bf16_t)(v1 * (float)p.h_pre[o + 1]);
            } else {
                ((float*)p.out)[o] = v0; ((float*)p.out)[o + 1] = v1;
            }
        }
    }
}

}  // namespace rvlm
