// Generic strided / batched fp32 GEMM on the VALU (k-ordered fmaf chain, fp32 accumulate).
// This is the fp32 PARITY path of the encoder (config C1: embeddings within 1e-4 relative of the
// reference's fp32 CPU path) and the home of the tiny GEMMs (head projection, CE logits).  The
// throughput path is gemm_bf16.hip.
#include "kernels.h"

namespace rvlm {

constexpr int FT = 64;   // tile
// FK = k step: 16 in general; 64 for launches that cannot fill the chip (the head projection: 24 workgroups walking
// K = 1024 were latency-bound at ~3 us per load -> LDS -> barrier round, 196 us per call; 4x fewer rounds with 4x the
// loads in flight).  The fmaf chain over k is the same for every FK, so results do not depend on it.
template <bool A_KC, bool B_KC, int FK>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32 p) {
    __shared__ float As[FK][FT + 4];
    __shared__ float Bs[FK][FT + 4];
    const int bz = blockIdx.z;
    const int b1 = bz / p.nb2, b2 = bz % p.nb2;
    const float* __restrict__ A = p.A + b1 * p.sab1 + b2 * p.sab2;
    const float* __restrict__ Bm = p.B + b1 * p.sbb1 + b2 * p.sbb2;
    const long coff = b1 * p.scb1 + b2 * p.scb2;
    const int m0 = blockIdx.y * FT, n0 = blockIdx.x * FT;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;

    for (int k0 = 0; k0 < p.K; k0 += FK) {
#pragma unroll
        for (int j = 0; j < FK / 4; ++j) {
            int idx = threadIdx.x + j * 256;
            int m, k;
            if (A_KC) { m = idx / FK; k = idx % FK; } else { m = idx & 63; k = idx >> 6; }
            float v = 0.0f;
            if (m0 + m < p.M && k0 + k < p.K) v = A[(long)(m0 + m) * p.sam + (long)(k0 + k) * p.sak];
            As[k][m] = v;
        }
#pragma unroll
        for (int j = 0; j < FK / 4; ++j) {
            int idx = threadIdx.x + j * 256;
            int n, k;
            if (B_KC) { n = idx / FK; k = idx % FK; } else { n = idx & 63; k = idx >> 6; }
            float v = 0.0f;
            if (n0 + n < p.N && k0 + k < p.K) v = Bm[(long)(n0 + n) * p.sbn + (long)(k0 + k) * p.sbk];
            Bs[k][n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            float a[4], b[4];
            *(float4*)a = *(const float4*)&As[k][ty * 4];
            *(float4*)b = *(const float4*)&Bs[k][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            const long o = coff + (long)m * p.scm + (long)n * p.scn;
            float v = p.alpha * acc[i][j];
            if (p.bias) v += p.bias[n];
            if (p.C_pre) p.C_pre[o] = v;
            if (p.act >= 0) v = act_fwd_precise(v, p.act);
            if (p.dact_h) v *= act_bwd_precise(p.dact_h[o], p.dact_kind);
            if (p.residual) v += p.residual[o];
            p.C[o] = v;
        }
    }
}

int gemm_f32(const GemmF32& p, hipStream_t s) {
    if (!p.A || !p.B || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nb1 <= 0 || p.nb2 <= 0)
        return fail(RVLM_ERR_ARG, "gemm_f32: bad arguments");
    dim3 grid(cdiv(p.N, FT), cdiv(p.M, FT), p.nb1 * p.nb2);
    const bool akc = (p.sak == 1), bkc = (p.sbk == 1);
    const bool deep = (long)grid.x * grid.y * grid.z < 256 && p.K >= 256;
#define RVLM_F32_LAUNCH(FKV)                                                                                       \
    do {                                                                                                           \
        if (akc && bkc) hipLaunchKernelGGL((gemm_f32_kernel<true, true, FKV>), grid, dim3(256), 0, s, p);          \
        else if (akc) hipLaunchKernelGGL((gemm_f32_kernel<true, false, FKV>), grid, dim3(256), 0, s, p);           \
        else if (bkc) hipLaunchKernelGGL((gemm_f32_kernel<false, true, FKV>), grid, dim3(256), 0, s, p);           \
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false, FKV>), grid, dim3(256), 0, s, p);                   \
    } while (0)
    if (deep) RVLM_F32_LAUNCH(64);
    else RVLM_F32_LAUNCH(16);
#undef RVLM_F32_LAUNCH
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

}  // namespace rvlm
