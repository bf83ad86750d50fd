// Split-bf16 ("x3") linears for the fp32-storage engine (round 6; precision RVLM_PREC_F32X3).
//
// The reference runs the attack in fp32 (train/pgd_train.py:30-38, no autocast).  The engine's fp32 mode computes every linear on
// the fp32 matrix pipe, 1/16 of the bf16 MFMA rate.  A split-bf16 product
//     a = a_hi + a_lo,  a_hi = bf16(a),  a_lo = bf16(a - a_hi)        a w ~ a_hi w_hi + a_hi w_lo + a_lo w_hi        (fp32 accumulate)
// carries ~16 mantissa bits (the dropped a_lo w_lo term is 2^-18 of the product) at three bf16 MFMAs per product, and it needs NO new
// GEMM kernel: the three products are ONE contraction of length 3 K,
//     A3 = [A_hi | A_hi | A_lo]   ([M, 3K], written by x3_split_rows below from the fp32 activation)
//     W3 = [W_hi | W_lo | W_hi]   ([N, 3K], built once per weight by x3_prepare_weight)
// which the persistent bf16 GEMM (gemm_bf16_256p.hip) runs as it stands, with its fp32 epilogues (bias, bias + residual), large
// term first.  What the emulation says this buys: oracle/split_bf16_emulation.py, profiles/r06_split_bf16_emulation.log (first FARE
// iteration: gradient-sign agreement with the fp32 oracle 0.9996-0.9998, bf16: 0.82-0.89).
// Activations stay fp32 in memory; LayerNorm, softmax, the attention products and the losses are the fp32 engine's own kernels.
#include "kernels.h"

namespace rvlm {

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t h = (bf16_t)v[e];
        hi[e] = h;
        lo[e] = (bf16_t)(v[e] - (float)h);
    }
}

// A fp32 [M, K] (row stride lda) -> A3 bf16 [rows_out, 3K]: [hi | hi | lo]; rows M .. rows_out - 1 are zero-filled (the GEMM's
// tiles read whole 128-row blocks).  8 columns per lane: two 16-byte loads, three 16-byte stores.
__global__ void __launch_bounds__(256)
x3_split_rows_kernel(const float* __restrict__ A, long lda, bf16_t* __restrict__ A3, int M, int rows_out, int K) {
    const int chunks = K >> 3;
    const long total = (long)rows_out * chunks;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / chunks), c = (int)(i - (long)r * chunks) * 8;
        bf16x8 hi = {}, lo = {};
        if (r < M) {
            float v[8];
            *(float4*)&v[0] = *(const float4*)(A + (long)r * lda + c);
            *(float4*)&v[4] = *(const float4*)(A + (long)r * lda + c + 4);
            split8(v, hi, lo);
        }
        bf16_t* o = A3 + (long)r * 3 * K + c;
        *(bf16x8*)o = hi; *(bf16x8*)(o + K) = hi; *(bf16x8*)(o + 2 * K) = lo;
    }
}
int x3_split_rows(const float* A, long lda, bf16_t* A3, int M, int rows_out, int K, hipStream_t s) {
    if (K % 8 != 0 || lda % 4 != 0 || (((size_t)A | (size_t)A3) & 15)) return fail(RVLM_ERR_ARG, "x3_split_rows: alignment");
    const long total = (long)rows_out * (K >> 3);
    const int grid = (int)std::min<long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(x3_split_rows_kernel, dim3(grid), dim3(256), 0, s, A, lda, A3, M, rows_out, K);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// weight fp32 [rows, cols] -> nk3 [rows, 3 cols] = [hi | lo | hi] and t3 [cols, 3 rows] = [hi^T | lo^T | hi^T] (the B operands
// of the forward and of the dgrad GEMM, both in the NT form).  Runs once per weight upload: no tuning.
__global__ void __launch_bounds__(256)
x3_weight_kernel(const float* __restrict__ src, bf16_t* __restrict__ nk3, bf16_t* __restrict__ t3, int rows, int cols) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        const float v = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.0f;
        tile[ty + 8 * k][tx] = v;
        if (r < rows && c < cols) {
            const bf16_t h = (bf16_t)v, l = (bf16_t)(v - (float)h);
            bf16_t* o = nk3 + (long)r * 3 * cols + c;
            o[0] = h; o[cols] = l; o[2 * cols] = h;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (r < rows && c < cols) {
            const float v = tile[tx][ty + 8 * k];
            const bf16_t h = (bf16_t)v, l = (bf16_t)(v - (float)h);
            bf16_t* o = t3 + (long)c * 3 * rows + r;
            o[0] = h; o[rows] = l; o[2 * rows] = h;
        }
    }
}
int x3_prepare_weight(const float* src, int rows, int cols, bf16_t* nk3, bf16_t* t3, hipStream_t s) {
    hipLaunchKernelGGL(x3_weight_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, s, src, nk3, t3, rows, cols);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// fc1's epilogue as its own pass (the GEMM leaves h = x W^T + b in fp32): mode 0: act(h); mode 1: out * act'(h) (fc2's dgrad
// through the activation).  The same precise functions as gemm_f32's epilogue (common.h).  The result goes to `out` (fp32) or, with
// A3 != null, STRAIGHT into the split copy [hi | hi | lo] the next linear reads ([rows_out, 3N], rows >= M zero) - the fp32
// round trip of the [M, 4W] activation (4 B written + 4 B read back by x3_split_rows) is skipped: ~0.2 ms per block and pass.
template <int MODE, bool SPLIT, bool DACT = false>
__global__ void __launch_bounds__(256)
x3_act_kernel(const float* __restrict__ hbuf, long ldh, float* __restrict__ out, long ldo, bf16_t* __restrict__ A3, int M, int rows_out,
              int N, int act, bf16_t* __restrict__ dact_bf = nullptr, long ld_dact = 0) {
    const int chunks = N >> 3;
    const long total = (long)(SPLIT ? rows_out : M) * chunks;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / chunks), c = (int)(i - (long)r * chunks) * 8;
        float v[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (r < M) {
            float hv[8];
            *(float4*)&hv[0] = *(const float4*)(hbuf + (long)r * ldh + c);
            *(float4*)&hv[4] = *(const float4*)(hbuf + (long)r * ldh + c + 4);
            if (MODE == 1) {
                *(float4*)&v[0] = *(const float4*)(out + (long)r * ldo + c);
                *(float4*)&v[4] = *(const float4*)(out + (long)r * ldo + c + 4);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = MODE == 0 ? act_fwd_precise(hv[e], act) : v[e] * act_bwd_precise(hv[e], act);
            if (DACT) {      // the handoff: act'(h) in bf16 for the bf16 handle's fc2 dgrad epilogue
                bf16x8 d;
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = (bf16_t)act_bwd_precise(hv[e], act);
                *(bf16x8*)(dact_bf + (long)r * ld_dact + c) = d;
            }
        }
        if (SPLIT) {
            bf16x8 hi, lo;
            split8(v, hi, lo);
            bf16_t* o = A3 + (long)r * 3 * N + c;
            *(bf16x8*)o = hi; *(bf16x8*)(o + N) = hi; *(bf16x8*)(o + 2 * N) = lo;
        } else {
            *(float4*)(out + (long)r * ldo + c) = *(const float4*)&v[0];
            *(float4*)(out + (long)r * ldo + c + 4) = *(const float4*)&v[4];
        }
    }
}
int x3_act(const float* hbuf, long ldh, float* out, long ldo, bf16_t* A3, int M, int rows_out, int N, int act, int mode, hipStream_t s,
           bf16_t* dact_bf, long ld_dact) {
    if (N % 8 != 0 || ldh % 4 != 0 || ldo % 4 != 0 || (dact_bf && (mode != 0 || ld_dact % 8 != 0))) return fail(RVLM_ERR_ARG, "x3_act: alignment");
    const long total = (long)(A3 ? rows_out : M) * (N >> 3);
    const int grid = (int)std::min<long>((total + 255) / 256, 256 * 32);
#define RVLM_X3_ACT(MODE, SPLIT) hipLaunchKernelGGL((x3_act_kernel<MODE, SPLIT>), dim3(grid), dim3(256), 0, s, hbuf, ldh, out, ldo, A3, M, rows_out, N, act)
    if (mode == 0 && dact_bf) {
        if (A3) hipLaunchKernelGGL((x3_act_kernel<0, true, true>), dim3(grid), dim3(256), 0, s, hbuf, ldh, out, ldo, A3, M, rows_out, N, act, dact_bf, ld_dact);
        else hipLaunchKernelGGL((x3_act_kernel<0, false, true>), dim3(grid), dim3(256), 0, s, hbuf, ldh, out, ldo, A3, M, rows_out, N, act, dact_bf, ld_dact);
    } else
    if (mode == 0) { if (A3) RVLM_X3_ACT(0, true); else RVLM_X3_ACT(0, false); }
    else { if (A3) RVLM_X3_ACT(1, true); else RVLM_X3_ACT(1, false); }
#undef RVLM_X3_ACT
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// ---- handoff to the bf16 engine (round 6): the forward ran here (fp32 storage), the input gradient runs on the bf16 handle's
// kernels (engine.hip, vit_backward_from) - FARE's first-iteration noise sits in the forward DIFFERENCE phi(x + d0) - phi(x), not in
// the cotangent's way back (oracle/split_bf16_emulation.py, arm x3fwd-bf16bwd-flash: gradient-sign agreement with the fp32 oracle
// 0.998 against 0.9996 for a split-bf16 backward and 0.82 for the bf16 forward).  What the bf16 backward reads in bf16 - qkv, the
// attention output, act'(h) - is exported from the fp32 tensors the forward kept; 8 columns per lane, 16-byte stores.
template <bool DACT>
__global__ void __launch_bounds__(256)
x3_export_bf16_kernel(const float* __restrict__ A, long lda, bf16_t* __restrict__ O, long ldo, int M, int N, int act) {
    const int chunks = N >> 3;
    const long total = (long)M * chunks;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / chunks), c = (int)(i - (long)r * chunks) * 8;
        float v[8];
        *(float4*)&v[0] = *(const float4*)(A + (long)r * lda + c);
        *(float4*)&v[4] = *(const float4*)(A + (long)r * lda + c + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(DACT ? act_bwd_precise(v[e], act) : v[e]);
        *(bf16x8*)(O + (long)r * ldo + c) = o;
    }
}
// dact = 0: out = bf16(A); 1: out = bf16(act'(A)) (A = the fp32 pre-activation fc1 left: what the bf16 forward's fc1 epilogue stores)
int x3_export_bf16(const float* A, long lda, bf16_t* out, long ldo, int M, int N, int act, int dact, hipStream_t s) {
    if (N % 8 != 0 || lda % 4 != 0 || ldo % 8 != 0 || (((size_t)A | (size_t)out) & 15)) return fail(RVLM_ERR_ARG, "x3_export_bf16: alignment");
    const long total = (long)M * (N >> 3);
    const int grid = (int)std::min<long>((total + 255) / 256, 256 * 32);
    if (dact) hipLaunchKernelGGL((x3_export_bf16_kernel<true>), dim3(grid), dim3(256), 0, s, A, lda, out, ldo, M, N, act);
    else hipLaunchKernelGGL((x3_export_bf16_kernel<false>), dim3(grid), dim3(256), 0, s, A, lda, out, ldo, M, N, act);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// LayerNorm forward whose output goes STRAIGHT into the split copy [hi | hi | lo] of the linear behind it (W = NV * 512; the same
// row arithmetic as layernorm_fwd8_kernel in vit_kernels.hip, fp32 result split instead of stored): the fp32 LayerNorm output -
// 4 B written + 4 B read back per element by x3_split_rows - never exists.  Rows M .. rows_out - 1 of A3 are zero-filled.
template <int NV>
__global__ void __launch_bounds__(256)
x3_layernorm_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                        bf16_t* __restrict__ A3, float* __restrict__ mean, float* __restrict__ rstd, int M, int rows_out) {
    constexpr int W = NV * 512;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows_out) return;
    bf16_t* o = A3 + (long)row * 3 * W + lane * 8;
    if (row >= M) {
#pragma unroll
        for (int it = 0; it < NV; ++it) { *(bf16x8*)(o + it * 512) = bf16x8{}; *(bf16x8*)(o + it * 512 + W) = bf16x8{}; *(bf16x8*)(o + it * 512 + 2 * W) = bf16x8{}; }
        return;
    }
    const float* xr = x + (long)row * ldx + lane * 8;
    float v[NV][8];
    float s = 0.0f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        *(float4*)&v[it][0] = *(const float4*)(xr + it * 512);
        *(float4*)&v[it][4] = *(const float4*)(xr + it * 512 + 4);
        s += ((v[it][0] + v[it][1]) + (v[it][2] + v[it][3])) + ((v[it][4] + v[it][5]) + (v[it][6] + v[it][7]));
    }
    const float mu = wave_sum(s) / (float)W;
    float q = 0.0f;
#pragma unroll
    for (int it = 0; it < NV; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[it][e] - mu; q = fmaf(d, d, q); }
    const float rs = rsqrtf(wave_sum(q) / (float)W + 1e-5f);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        float g[8], b[8], y[8];
        *(float4*)&g[0] = *(const float4*)(gamma + it * 512 + lane * 8);
        *(float4*)&g[4] = *(const float4*)(gamma + it * 512 + lane * 8 + 4);
        *(float4*)&b[0] = *(const float4*)(beta + it * 512 + lane * 8);
        *(float4*)&b[4] = *(const float4*)(beta + it * 512 + lane * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (v[it][e] - mu) * rs * g[e] + b[e];
        bf16x8 hi, lo;
        split8(y, hi, lo);
        *(bf16x8*)(o + it * 512) = hi; *(bf16x8*)(o + it * 512 + W) = hi; *(bf16x8*)(o + it * 512 + 2 * W) = lo;
    }
}
// false: width not covered (the caller runs the fp32 LayerNorm and lets the linear split its output)
bool x3_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, bf16_t* A3, float* mean, float* rstd, int M,
                      int rows_out, int W, hipStream_t s) {
    if (W % 512 != 0 || W > 2048 || ldx % 4 != 0) return false;
    const dim3 grid(cdiv(rows_out, 4)), block(256);
    switch (W / 512) {
        case 1: hipLaunchKernelGGL((x3_layernorm_fwd_kernel<1>), grid, block, 0, s, x, ldx, gamma, beta, A3, mean, rstd, M, rows_out); break;
        case 2: hipLaunchKernelGGL((x3_layernorm_fwd_kernel<2>), grid, block, 0, s, x, ldx, gamma, beta, A3, mean, rstd, M, rows_out); break;
        case 3: hipLaunchKernelGGL((x3_layernorm_fwd_kernel<3>), grid, block, 0, s, x, ldx, gamma, beta, A3, mean, rstd, M, rows_out); break;
        default: hipLaunchKernelGGL((x3_layernorm_fwd_kernel<4>), grid, block, 0, s, x, ldx, gamma, beta, A3, mean, rstd, M, rows_out); break;
    }
    return true;
}

}  // namespace rvlm
