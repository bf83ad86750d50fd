// Generic strided / batched fp32 GEMM: C = alpha * A B (+ epilogue), every output element a k-ORDERED fp32 fmaf chain.
// This is the fp32 PARITY path of the encoder (config C1: embeddings within 1e-4 relative of the
// reference's fp32 CPU path; the reference's hot path is fp32 everywhere, train/pgd_train.py:30-38) and the home of
// the tiny GEMMs (head projection, CE logits).  The throughput path is gemm_bf16*.hip.
//
// Round 5: the tiles run on the matrix pipe.  v_mfma_f32_32x32x2_f32 takes one fp32 A and one fp32 B value per lane
// (lane l: A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31]) and computes D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)):
// one rounding per product, no wider internal sum - bit for bit the chain the VALU kernel below evaluates
// (MI355X_MICROARCH.md "Matrix cores", cdna_hip_programming.md section 3 "FP32-input MFMA"; asserted here by
// tests/test_gpu_kernels.py::test_gemm_f32_mfma_bit_identical_to_valu_chain through rvlm_k_gemm_f32_set_valu), at 64
// cycles per 32x32x2 block and SIMD = the fp32 vector peak (157 TFLOP/s) with none of the VALU kernel's LDS-read and
// issue overheads: the 64x64x16 VALU tile ran at ~40 TFLOP/s.
#include "kernels.h"

namespace rvlm {

// ---- the VALU tile (64 x 64 x FK): the explicit fmaf chain; since round 5 the test hook's comparison arm only ------------
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


// ---- fp32 tiles on the matrix pipe ----------------------------------------------------------------------------------
// Workgroup = 4 waves (2 x 2), tile BM x BN (128 or 64 each), wave tile (BM/2) x (BN/2) = TM x TN blocks of 32 x 32, BK = 16.
// LDS holds both operand tiles k-major ([16 k][rows], double-buffered): the MFMA operand read is then ONE ds_read_b32 per
// block and k-pair, lanes 0-31 on 32 consecutive floats of row k, lanes 32-63 of row k + 1 (conflict-free).  Global ->
// register -> LDS with the next k-tile's loads issued before the MFMAs of the current one (one barrier per k-tile).
// A k-contiguous operand (activations, [N, K] weights) is read as 16-byte chunks of 4 k and scattered to 4 k-rows of the
// LDS tile (lanes = consecutive rows: conflict-free ds_write_b32); a row-contiguous operand ([K, N] weights, V, the
// transposed score matrices) as 16-byte chunks of 4 rows written whole.  vec_a / vec_b = the operand's base, strides and
// extent allow 16-byte accesses (else scalar, guarded loads: S = 257 score matrices).
#ifndef RVLM_F32_BK
#define RVLM_F32_BK 16      // k-tile depth of the fp32 MFMA tiles (A/B knob: 32 = half the barriers, 64 KiB of LDS, two workgroups per CU)
#endif
template <int ROWS, bool KC>
struct F32TileLoader {
    static constexpr int NV = ROWS * RVLM_F32_BK / 4 / 256;   // 16-byte chunks per thread (BK = 16: 2 at ROWS = 128, 1 at ROWS = 64)
    float4 v[NV];
    __device__ __forceinline__ void load(const float* __restrict__ base, long s_row, long s_k, int row0, int nrows, int k0,
                                         int K, int tid, bool VEC) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + j * 256;
            float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (KC) {
                const int r = idx % ROWS, k = k0 + 4 * (idx / ROWS);
                const bool rv = row0 + r < nrows;
                const float* g = base + (long)(row0 + r) * s_row + (long)k * s_k;
                if (VEC) {
                    if (rv && k < K) t = *(const float4*)g;
                } else {
                    if (rv && k + 0 < K) t.x = g[0];
                    if (rv && k + 1 < K) t.y = g[s_k];
                    if (rv && k + 2 < K) t.z = g[2 * s_k];
                    if (rv && k + 3 < K) t.w = g[3 * s_k];
                }
            } else {
                const int r = row0 + 4 * (idx % (ROWS / 4)), k = k0 + idx / (ROWS / 4);
                const float* g = base + (long)r * s_row + (long)k * s_k;
                if (VEC) {
                    if (r < nrows && k < K) t = *(const float4*)g;
                } else {
                    if (k < K && r + 0 < nrows) t.x = g[0];
                    if (k < K && r + 1 < nrows) t.y = g[s_row];
                    if (k < K && r + 2 < nrows) t.z = g[2 * s_row];
                    if (k < K && r + 3 < nrows) t.w = g[3 * s_row];
                }
            }
            v[j] = t;
        }
    }
    __device__ __forceinline__ void store(float (*tile)[ROWS], int tid) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = tid + j * 256;
            if (KC) {
                const int r = idx % ROWS, k = 4 * (idx / ROWS);
                tile[k + 0][r] = v[j].x; tile[k + 1][r] = v[j].y; tile[k + 2][r] = v[j].z; tile[k + 3][r] = v[j].w;
            } else {
                *(float4*)&tile[idx / (ROWS / 4)][4 * (idx % (ROWS / 4))] = v[j];
            }
        }
    }
};

// PLAIN: alpha and bias only (the instantiations with the activation / act' / residual epilogue are 5x the code)
template <int BM, int BN, bool A_KC, bool B_KC, bool PLAIN>
__global__ void __launch_bounds__(256) gemm_f32_mfma_kernel(GemmF32 p, int vec_a, int vec_b) {
    constexpr int BK = RVLM_F32_BK, TM = BM / 64, TN = BN / 64;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];
    const int bz = blockIdx.z;
    const int b1 = bz / p.nb2, b2 = bz % p.nb2;
    const float* __restrict__ A = p.A + b1 * p.sab1 + b2 * p.sab2;
    const float* __restrict__ Bm = p.B + b1 * p.sbb1 + b2 * p.sbb2;
    const long coff = b1 * p.scb1 + b2 * p.scb2;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = (w >> 1) * (BM / 2), wn = (w & 1) * (BN / 2);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const bool va = vec_a != 0, vb = vec_b != 0;     // (kernel arguments: wave-uniform branches)
    // loader bounds: exact, or rounded up to 4 along the operand's contiguous dimension when its padding is readable
    // (GemmF32::pad4: zeros along k; along the rows the extra rows only feed outputs that are never stored)
    const int Ka = (A_KC && (p.pad4 & 1)) ? ((p.K + 3) & ~3) : p.K, Ma = (!A_KC && (p.pad4 & 1)) ? ((p.M + 3) & ~3) : p.M;
    const int Kb = (B_KC && (p.pad4 & 2)) ? ((p.K + 3) & ~3) : p.K, Nb = (!B_KC && (p.pad4 & 2)) ? ((p.N + 3) & ~3) : p.N;
    F32TileLoader<BM, A_KC> la;
    F32TileLoader<BN, B_KC> lb;
    la.load(A, p.sam, p.sak, m0, Ma, 0, Ka, tid, va);
    lb.load(Bm, p.sbn, p.sbk, n0, Nb, 0, Kb, tid, vb);
    la.store(As[0], tid);
    lb.store(Bs[0], tid);
    __syncthreads();
    const int nk = (p.K + BK - 1) / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(A, p.sam, p.sak, m0, Ma, (kt + 1) * BK, Ka, tid, va);
            lb.load(Bm, p.sbn, p.sbk, n0, Nb, (kt + 1) * BK, Kb, tid, vb);
        }
        // the operands of k-pair kk + 1 are requested BEFORE the MFMAs of pair kk (round 5, PMC: with read -> wait -> 4 MFMAs per
        // pair the matrix pipe was busy 70 % of the kernel, the waves parked at s_waitcnt 17-25 % of their cycles)
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = As[cur][hi][wm + 32 * i + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = Bs[cur][hi][wn + 32 * j + l31];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {     // k increases: the sum order of every output element is k = 0, 1, 2, ...
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = As[cur][2 * kk + 2 + hi][wm + 32 * i + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = Bs[cur][2 * kk + 2 + hi][wn + 32 * j + l31];
            }
            __builtin_amdgcn_sched_barrier(0);     // (hipcc otherwise sinks the requests behind the MFMAs, into the same registers)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {     // the other buffer: every wave left it at the barrier that closed step kt - 1
            la.store(As[cur ^ 1], tid);
            lb.store(Bs[cur ^ 1], tid);
        }
        __syncthreads();
    }
    // D layout of the 32x32 blocks: column n = lane & 31, row m = (e & 3) + 8 (e >> 2) + 4 (lane >> 5): a 32-lane half
    // stores 128 contiguous bytes of one output row per register when scn == 1
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + 32 * j + l31;
            if (n >= p.N) continue;
            const float bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (m >= p.M) continue;
                const long o = coff + (long)m * p.scm + (long)n * p.scn;
                float v = p.alpha * acc[i][j][e];
                if (p.bias) v += bias;
                if (!PLAIN) {
                    if (p.C_pre) p.C_pre[o] = v;
                    if (p.act >= 0) v = act_fwd_precise(v, p.act);
                    if (p.dact_h) v *= act_bwd_precise(p.dact_h[o], p.dact_kind);
                    if (p.residual) v += p.residual[o];
                }
                p.C[o] = v;
            }
        }
}

static bool g_f32_valu = false;     // test hook: the VALU chain kernel for every shape (the bit-identity check's other side)
void gemm_f32_set_valu(int on) { g_f32_valu = on != 0; }

template <int BM, int BN>
static void launch_f32_mfma(const GemmF32& p, bool akc, bool bkc, bool va, bool vb, hipStream_t s) {
    const dim3 grid(cdiv(p.N, BN), cdiv(p.M, BM), p.nb1 * p.nb2);
    const bool plain = p.act < 0 && !p.C_pre && !p.dact_h && !p.residual;
#define RVLM_F32M(AK, BK_)                                                                                                    \
    do {                                                                                                                      \
        if (plain) hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, AK, BK_, true>), grid, dim3(256), 0, s, p, (int)va, (int)vb);   \
        else hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, AK, BK_, false>), grid, dim3(256), 0, s, p, (int)va, (int)vb);        \
    } while (0)
    if (akc && bkc) RVLM_F32M(true, true);
    else if (akc) RVLM_F32M(true, false);
    else if (bkc) RVLM_F32M(false, true);
    else RVLM_F32M(false, false);
#undef RVLM_F32M
}

// 16-byte accesses to one operand: contiguous along k (kc) or along its rows, every other stride, the batch offsets and
// the base a multiple of 4 floats, and the contiguous extent a multiple of 4 (a chunk is then all inside or all outside)
static bool f32_vec_ok(const float* base, bool kc, long s_row, long s_k, long sb1, long sb2, int nrows, int K, bool pad) {
    if (((uintptr_t)base & 15) || (sb1 & 3) || (sb2 & 3)) return false;
    if (kc) return s_k == 1 && !(s_row & 3) && (pad || !(K & 3));
    return s_row == 1 && !(s_k & 3) && (pad || !(nrows & 3));
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
    const bool akc = (p.sak == 1), bkc = (p.sbk == 1);
    if (!g_f32_valu) {
        // matrix-pipe tiles: the tile shape with the least padded volume, the larger one on a tie if it still gives
        // every CU a workgroup (S = 257 attention products: 64-row tiles pad to 320, 128-row tiles to 384)
        const bool va = f32_vec_ok(p.A, akc, p.sam, p.sak, p.sab1, p.sab2, p.M, p.K, p.pad4 & 1);
        const bool vb = f32_vec_ok(p.B, bkc, p.sbn, p.sbk, p.sbb1, p.sbb2, p.N, p.K, p.pad4 & 2);
        const long nb = (long)p.nb1 * p.nb2;
        auto vol = [&](int bm, int bn) { return (long)cdiv(p.M, bm) * bm * ((long)cdiv(p.N, bn) * bn); };
        auto wgs = [&](int bm, int bn) { return (long)cdiv(p.M, bm) * cdiv(p.N, bn) * nb; };
        int bm = 64, bn = 64;
        if (vol(128, 64) <= vol(bm, bn) && wgs(128, 64) >= 256) { bm = 128; bn = 64; }
        if (vol(128, 128) <= vol(bm, bn) && wgs(128, 128) >= 256) { bm = 128; bn = 128; }
        if (bm == 128 && bn == 128) {
            // Tail split: 128 x 128 workgroups are resident three per CU (139 registers), i.e. the launch runs in rounds of 768
            // equal tiles and a last round that is mostly empty costs a whole tile time - M = 32 896: the QKV projection is
            // 8.03 rounds, the N = 1024 products 2.68.  The main launch therefore takes the largest prefix of row tiles that
            // fills whole rounds, and the remaining rows go in 64 x 64 tiles (a quarter of the tile time, more slots).  Every
            // output element is the same k-ordered chain under any tiling, so the split changes no bit.
            const long slots = 3 * 256, tm = cdiv(p.M, 128), tn = cdiv(p.N, 128), tiles = tm * tn * nb;
            const long rounds = tiles / slots, rem = tiles % slots;
            const long tm_main = nb == 1 ? (rounds * slots) / tn : tm;
            if (nb == 1 && rounds >= 1 && rem > 0 && rem < slots * 3 / 4 && tm_main >= 1 && tm_main < tm) {
                GemmF32 a = p, b = p;
                a.M = (int)(tm_main * 128);
                const long r0 = tm_main * 128;
                b.M = p.M - (int)r0;
                b.A = p.A + r0 * p.sam;
                b.C = p.C + r0 * p.scm;
                if (p.C_pre) b.C_pre = p.C_pre + r0 * p.scm;
                if (p.dact_h) b.dact_h = p.dact_h + r0 * p.scm;
                if (p.residual) b.residual = p.residual + r0 * p.scm;
                launch_f32_mfma<128, 128>(a, akc, bkc, va, vb, s);
                RVLM_CHECK_LAUNCH();
                const bool va_t = f32_vec_ok(b.A, akc, b.sam, b.sak, b.sab1, b.sab2, b.M, b.K, b.pad4 & 1);
                launch_f32_mfma<64, 64>(b, akc, bkc, va_t, vb, s);
            } else {
                launch_f32_mfma<128, 128>(p, akc, bkc, va, vb, s);
            }
        } else if (bm == 128) launch_f32_mfma<128, 64>(p, akc, bkc, va, vb, s);
        else launch_f32_mfma<64, 64>(p, akc, bkc, va, vb, s);
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
    dim3 grid(cdiv(p.N, FT), cdiv(p.M, FT), p.nb1 * p.nb2);
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
