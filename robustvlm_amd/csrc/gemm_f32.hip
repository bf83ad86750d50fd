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

// ---- few-row problems (the projection head: M = batch).  The 64x64-tile kernel gives them a couple of dozen
// workgroups (220 us for 128 x 768 x 1024); here every wave owns a 4-row strip and the chip is covered.
// NN: C[m, n] = alpha * sum_k A[m, k] B[k, n]  (A row-major, B with n contiguous).  A workgroup owns 4 rows x 64
// columns; its 4 waves split K (the chip holds ~1.5 of these waves per CU, so the loop is bound by load latency: K/4
// steps per wave instead of K), lanes = 64 consecutive n, the A values are wave-uniform scalar loads; the four partial
// sums are combined in wave order through LDS (deterministic; associates differently from the tiled kernel).
__global__ void __launch_bounds__(256) gemm_f32_skinny_nn_kernel(GemmF32 p) {
    __shared__ float part[4][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * 4;
    const int nc = min(n, p.N - 1);
    const int kc = (p.K + 3) >> 2, k0 = wave * kc, k1 = min(p.K, k0 + kc);
    const float* __restrict__ Bp = p.B + (long)nc * p.sbn;
    const float* __restrict__ A0 = p.A + (long)min(m0, p.M - 1) * p.sam;
    const float* __restrict__ A1 = p.A + (long)min(m0 + 1, p.M - 1) * p.sam;
    const float* __restrict__ A2 = p.A + (long)min(m0 + 2, p.M - 1) * p.sam;
    const float* __restrict__ A3 = p.A + (long)min(m0 + 3, p.M - 1) * p.sam;
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;
    int k = k0;
    for (; k + 32 <= k1; k += 32) {
        float b[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) b[u] = Bp[(long)(k + u) * p.sbk];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            c0 = fmaf(A0[k + u], b[u], c0); c1 = fmaf(A1[k + u], b[u], c1);
            c2 = fmaf(A2[k + u], b[u], c2); c3 = fmaf(A3[k + u], b[u], c3);
        }
    }
    for (; k < k1; ++k) {
        const float b = Bp[(long)k * p.sbk];
        c0 = fmaf(A0[k], b, c0); c1 = fmaf(A1[k], b, c1); c2 = fmaf(A2[k], b, c2); c3 = fmaf(A3[k], b, c3);
    }
    part[wave][0][lane] = c0; part[wave][1][lane] = c1; part[wave][2][lane] = c2; part[wave][3][lane] = c3;
    __syncthreads();
    const int m = m0 + wave;               // wave w finishes row w
    if (n < p.N && m < p.M) {
        const float v = ((part[0][wave][lane] + part[1][wave][lane]) + part[2][wave][lane]) + part[3][wave][lane];
        p.C[(long)m * p.scm + n] = p.alpha * v;
    }
}
// NT: C[m, n] = alpha * sum_k A[m, k] B[n, k]  (both K-contiguous).  A wave owns a 4 x 4 output block, lanes stride K.
__global__ void __launch_bounds__(256) gemm_f32_skinny_nt_kernel(GemmF32 p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (blockIdx.x * 4 + wave) * 4, m0 = blockIdx.y * 4;
    if (n0 >= p.N) return;
    const float* __restrict__ Ar[4];
    const float* __restrict__ Br[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        Ar[i] = p.A + (long)min(m0 + i, p.M - 1) * p.sam;
        Br[i] = p.B + (long)min(n0 + i, p.N - 1) * p.sbn;
    }
    float c[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = 0.0f;
    for (int k = lane; k < p.K; k += 64) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = Ar[i][k]; b[i] = Br[i][k]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = fmaf(a[i], b[j], c[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = wave_sum(c[i][j]);
            if (lane == 0 && m0 + i < p.M && n0 + j < p.N) p.C[(long)(m0 + i) * p.scm + n0 + j] = p.alpha * v;
        }
}

int gemm_f32(const GemmF32& p, hipStream_t s) {
    if (!p.A || !p.B || !p.C || p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nb1 <= 0 || p.nb2 <= 0)
        return fail(RVLM_ERR_ARG, "gemm_f32: bad arguments");
    const bool plain = !p.bias && p.act < 0 && !p.C_pre && !p.dact_h && !p.residual && p.nb1 == 1 && p.nb2 == 1 &&
                       p.scn == 1;
    if (plain && p.M <= 512 && p.K >= 256 && p.sak == 1 && (long)cdiv(p.N, 64) * cdiv(p.M, 64) < 128) {
        if (p.sbn == 1) {
            hipLaunchKernelGGL(gemm_f32_skinny_nn_kernel, dim3(cdiv(p.N, 64), cdiv(p.M, 4)), dim3(256), 0, s, p);
            RVLM_CHECK_LAUNCH();
            return RVLM_OK;
        }
        if (p.sbk == 1) {
            hipLaunchKernelGGL(gemm_f32_skinny_nt_kernel, dim3(cdiv(p.N, 16), cdiv(p.M, 4)), dim3(256), 0, s, p);
            RVLM_CHECK_LAUNCH();
            return RVLM_OK;
        }
    }
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
