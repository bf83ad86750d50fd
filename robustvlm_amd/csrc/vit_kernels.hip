// HBM-bound glue kernels of the encoder: LayerNorm fwd/bwd (wave-shuffle row reductions), patch
// im2col (+delta, Normalize fused into the load) / col2im, CLS+pos+ln_pre, L2-normalise, softmax
// rows for the fp32 path, dtype conversion.  One wave (64 lanes) per row, float4 / bf16x4 accesses.
#include "kernels.h"

namespace rvlm {

// ---- 4-wide row accessors ----------------------------------------------------------------------
__device__ __forceinline__ void load4(const float* p, float (&v)[4]) { *(float4*)v = *(const float4*)p; }
__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
    bf16x4 t = *(const bf16x4*)p;
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) { *(float4*)p = *(const float4*)v; }
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    bf16x4 t;
    t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
    *(bf16x4*)p = t;
}

constexpr int LN_MAXV = 8;  // register-cached float4 chunks per lane: W <= 2048

// =============================================================================================
// LayerNorm forward
// =============================================================================================
template <typename TO>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                     const float* __restrict__ beta, TO* __restrict__ y, long ldy,
                     float* __restrict__ mean, float* __restrict__ rstd, int M, int W) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long)row * ldx;
    float v[LN_MAXV][4];
    float s = 0.0f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) { load4(xr + c, v[it]); s += (v[it][0] + v[it][1]) + (v[it][2] + v[it][3]); }
    }
    const float mu = wave_sum(s) / (float)W;
    float q = 0.0f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[it][e] - mu; q = fmaf(d, d, q); }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)W + 1e-5f);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    TO* yr = y + (long)row * ldy;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float g[4], b[4], o[4];
            load4(gamma + c, g); load4(beta + c, b);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mu) * rs * g[e] + b[e];
            store4(yr + c, o);
        }
    }
}

// bf16 output, W = NV * 512: every lane owns 8 consecutive columns per 512-column slab (two 16-byte loads, one 16-byte
// bf16 store - the generic kernel's 8-byte stores reach 4.9 TB/s on the encoder's [32 896, 1024] rows).  Same arithmetic
// per element; the row sums associate differently from the generic kernel (fp32, ~1 ulp of the mean / variance).
template <int NV>
__global__ void __launch_bounds__(256)
layernorm_fwd8_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                      const float* __restrict__ beta, bf16_t* __restrict__ y, long ldy,
                      float* __restrict__ mean, float* __restrict__ rstd, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (long)row * ldx + lane * 8;
    float v[NV][8];
    float s = 0.0f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        *(float4*)&v[it][0] = *(const float4*)(xr + it * 512);
        *(float4*)&v[it][4] = *(const float4*)(xr + it * 512 + 4);
        s += ((v[it][0] + v[it][1]) + (v[it][2] + v[it][3])) + ((v[it][4] + v[it][5]) + (v[it][6] + v[it][7]));
    }
    const float mu = wave_sum(s) / (float)(NV * 512);
    float q = 0.0f;
#pragma unroll
    for (int it = 0; it < NV; ++it)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[it][e] - mu; q = fmaf(d, d, q); }
    const float rs = rsqrtf(wave_sum(q) / (float)(NV * 512) + 1e-5f);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    bf16_t* yr = y + (long)row * ldy + lane * 8;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        float g[8], b[8];
        *(float4*)&g[0] = *(const float4*)(gamma + it * 512 + lane * 8);
        *(float4*)&g[4] = *(const float4*)(gamma + it * 512 + lane * 8 + 4);
        *(float4*)&b[0] = *(const float4*)(beta + it * 512 + lane * 8);
        *(float4*)&b[4] = *(const float4*)(beta + it * 512 + lane * 8 + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((v[it][e] - mu) * rs * g[e] + b[e]);
        *(bf16x8*)(yr + it * 512) = o;
    }
}
template <typename TO>
static bool layernorm_fwd8(const float*, long, const float*, const float*, TO*, long, float*, float*, int, int, hipStream_t) {
    return false;
}
template <>
bool layernorm_fwd8<bf16_t>(const float* x, long ldx, const float* gamma, const float* beta, bf16_t* y, long ldy,
                            float* mean, float* rstd, int M, int W, hipStream_t s) {
    if (W % 512 != 0 || W > 2048 || ldx % 4 != 0 || ldy % 8 != 0) return false;
    const dim3 grid(cdiv(M, 4)), block(256);
    switch (W / 512) {
        case 1: hipLaunchKernelGGL((layernorm_fwd8_kernel<1>), grid, block, 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M); break;
        case 2: hipLaunchKernelGGL((layernorm_fwd8_kernel<2>), grid, block, 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M); break;
        case 3: hipLaunchKernelGGL((layernorm_fwd8_kernel<3>), grid, block, 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M); break;
        default: hipLaunchKernelGGL((layernorm_fwd8_kernel<4>), grid, block, 0, s, x, ldx, gamma, beta, y, ldy, mean, rstd, M); break;
    }
    return true;
}

template <typename TO>
int layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, TO* y, long ldy,
                  float* mean, float* rstd, int M, int W, hipStream_t s) {
    if (W % 4 != 0 || W > LN_MAXV * 256) return fail(RVLM_ERR_UNSUPPORTED, "layernorm: width");
    if (layernorm_fwd8<TO>(x, ldx, gamma, beta, y, ldy, mean, rstd, M, W, s)) {
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
    hipLaunchKernelGGL((layernorm_fwd_kernel<TO>), dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, gamma,
                       beta, y, ldy, mean, rstd, M, W);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int layernorm_fwd<float>(const float*, long, const float*, const float*, float*, long,
                                  float*, float*, int, int, hipStream_t);
template int layernorm_fwd<bf16_t>(const float*, long, const float*, const float*, bf16_t*, long,
                                   float*, float*, int, int, hipStream_t);

// =============================================================================================
// LayerNorm backward (input gradient only): dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),
// g = dy*gamma.  Accumulates into the fp32 residual-stream gradient and emits its low-precision
// copy (the A operand of the next dgrad GEMM).
// =============================================================================================
template <typename TI, typename TB>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const TI* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                     const float* __restrict__ gamma, const float* __restrict__ mean,
                     const float* __restrict__ rstd, float* __restrict__ dres, long lddres,
                     TB* __restrict__ dres_lp, long ldlp, int accumulate, int M, int W) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    if (accumulate < 0) accumulate = (row % -accumulate) == 0;     // -S: only the class-token rows hold a gradient yet
    const float mu = mean[row], rs = rstd[row];
    const TI* dyr = dy + (long)row * lddy;
    const float* xr = x + (long)row * ldx;
    float g[LN_MAXV][4], xh[LN_MAXV][4];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float d[4], xv[4], gm[4];
            load4(dyr + c, d); load4(xr + c, xv); load4(gamma + c, gm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[it][e] = d[e] * gm[e];
                xh[it][e] = (xv[e] - mu) * rs;
                s1 += g[it][e];
                s2 = fmaf(g[it][e], xh[it][e], s2);
            }
        }
    }
    const float c1 = wave_sum(s1) / (float)W;
    const float c2 = wave_sum(s2) / (float)W;
    // dres == null: the low-precision copy IS the residual-gradient stream (read to accumulate, written back)
    float* dr = dres ? dres + (long)row * lddres : nullptr;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float o[4];
            if (!accumulate) { o[0] = o[1] = o[2] = o[3] = 0.0f; }
            else if (dr) load4(dr + c, o);
            else load4(dres_lp + (long)row * ldlp + c, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += rs * (g[it][e] - c1 - xh[it][e] * c2);
            if (dr) store4(dr + c, o);
            if (dres_lp) store4(dres_lp + (long)row * ldlp + c, o);
        }
    }
}

// bf16 dy / dres_lp, W = NV * 512: 8 consecutive columns per lane and slab (16-byte bf16 accesses), as in the forward
template <int NV, bool LPONLY = false>
__global__ void __launch_bounds__(256)
layernorm_bwd8_kernel(const bf16_t* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                      const float* __restrict__ gamma, const float* __restrict__ mean,
                      const float* __restrict__ rstd, float* __restrict__ dres, long lddres,
                      bf16_t* __restrict__ dres_lp, long ldlp, int accumulate, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    if (accumulate < 0) accumulate = (row % -accumulate) == 0;
    const float mu = mean[row], rs = rstd[row];
    const bf16_t* dyr = dy + (long)row * lddy + lane * 8;
    const float* xr = x + (long)row * ldx + lane * 8;
    float g[NV][8], xh[NV][8];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const bf16x8 d = *(const bf16x8*)(dyr + it * 512);
        float xv[8], gm[8];
        *(float4*)&xv[0] = *(const float4*)(xr + it * 512);
        *(float4*)&xv[4] = *(const float4*)(xr + it * 512 + 4);
        *(float4*)&gm[0] = *(const float4*)(gamma + it * 512 + lane * 8);
        *(float4*)&gm[4] = *(const float4*)(gamma + it * 512 + lane * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            g[it][e] = (float)d[e] * gm[e];
            xh[it][e] = (xv[e] - mu) * rs;
            s1 += g[it][e];
            s2 = fmaf(g[it][e], xh[it][e], s2);
        }
    }
    const float c1 = wave_sum(s1) / (float)(NV * 512);
    const float c2 = wave_sum(s2) / (float)(NV * 512);
    // LPONLY: the bf16 copy IS the residual-gradient stream - read to accumulate, written back: 10 instead of 16 B per element
    float* dr = LPONLY ? nullptr : dres + (long)row * lddres + lane * 8;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        float o[8];
        if (!accumulate) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.0f;
        } else if (LPONLY) {
            const bf16x8 a = *(const bf16x8*)(dres_lp + (long)row * ldlp + lane * 8 + it * 512);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (float)a[e];
        } else {
            *(float4*)&o[0] = *(const float4*)(dr + it * 512);
            *(float4*)&o[4] = *(const float4*)(dr + it * 512 + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += rs * (g[it][e] - c1 - xh[it][e] * c2);
        if (!LPONLY) {
            *(float4*)(dr + it * 512) = *(const float4*)&o[0];
            *(float4*)(dr + it * 512 + 4) = *(const float4*)&o[4];
        }
        if (dres_lp) {
            bf16x8 ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) ol[e] = (bf16_t)o[e];
            *(bf16x8*)(dres_lp + (long)row * ldlp + lane * 8 + it * 512) = ol;
        }
    }
}
template <typename TI, typename TB>
static bool layernorm_bwd8(const TI*, long, const float*, long, const float*, const float*, const float*, float*, long, TB*,
                           long, int, int, int, hipStream_t) {
    return false;
}
template <>
bool layernorm_bwd8<bf16_t, bf16_t>(const bf16_t* dy, long lddy, const float* x, long ldx, const float* gamma,
                                    const float* mean, const float* rstd, float* dres, long lddres, bf16_t* dres_lp,
                                    long ldlp, int accumulate, int M, int W, hipStream_t s) {
    if (W % 512 != 0 || W > 2048 || lddy % 8 != 0 || ldx % 4 != 0 || lddres % 4 != 0 || ldlp % 8 != 0) return false;
    if (!dres && !dres_lp) return false;
    const dim3 grid(cdiv(M, 4)), block(256);
#define RVLM_LNB8(NVV) do { if (dres) hipLaunchKernelGGL((layernorm_bwd8_kernel<NVV, false>), grid, block, 0, s, dy, lddy, x, ldx, gamma, mean, \
                                                          rstd, dres, lddres, dres_lp, ldlp, accumulate, M);                                  \
                            else hipLaunchKernelGGL((layernorm_bwd8_kernel<NVV, true>), grid, block, 0, s, dy, lddy, x, ldx, gamma, mean,     \
                                                    rstd, dres, lddres, dres_lp, ldlp, accumulate, M); } while (0)
    switch (W / 512) {
        case 1: RVLM_LNB8(1); break;
        case 2: RVLM_LNB8(2); break;
        case 3: RVLM_LNB8(3); break;
        default: RVLM_LNB8(4); break;
    }
#undef RVLM_LNB8
    return true;
}

template <typename TI, typename TB>
int layernorm_bwd(const TI* dy, long lddy, const float* x, long ldx, const float* gamma,
                  const float* mean, const float* rstd, float* dres, long lddres, TB* dres_lp,
                  long ldlp, int accumulate, int M, int W, hipStream_t s) {
    if (W % 4 != 0 || W > LN_MAXV * 256) return fail(RVLM_ERR_UNSUPPORTED, "layernorm: width");
    if (layernorm_bwd8<TI, TB>(dy, lddy, x, ldx, gamma, mean, rstd, dres, lddres, dres_lp, ldlp, accumulate, M, W, s)) {
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
    hipLaunchKernelGGL((layernorm_bwd_kernel<TI, TB>), dim3(cdiv(M, 4)), dim3(256), 0, s, dy, lddy,
                       x, ldx, gamma, mean, rstd, dres, lddres, dres_lp, ldlp, accumulate, M, W);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int layernorm_bwd<float, float>(const float*, long, const float*, long, const float*,
                                         const float*, const float*, float*, long, float*, long,
                                         int, int, int, hipStream_t);
template int layernorm_bwd<bf16_t, bf16_t>(const bf16_t*, long, const float*, long, const float*,
                                           const float*, const float*, float*, long, bf16_t*, long,
                                           int, int, int, hipStream_t);
template int layernorm_bwd<float, bf16_t>(const float*, long, const float*, long, const float*,
                                          const float*, const float*, float*, long, bf16_t*, long,
                                          int, int, int, hipStream_t);

// =============================================================================================
// Patch im2col with (x + delta) and Normalize fused:  A0[row, col]
// =============================================================================================
// I = index type: int whenever every linear index fits 31 bits (64-bit div / mod cost ~100 instructions each here)
template <typename T, typename I>
__global__ void __launch_bounds__(256)
im2col_kernel(const float* __restrict__ x, const float* __restrict__ delta, int B, int img, int P,
              float m0, float m1, float m2, float s0, float s1, float s2, T* __restrict__ A0,
              long lda, int Kpad) {
    const int g = img / P, PP = P * P, K = 3 * PP;
    const I total = (I)B * g * g * Kpad;
    for (I idx = (I)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (I)gridDim.x * blockDim.x) {
        const int col = (int)(idx % Kpad);
        const I row = idx / Kpad;
        float v = 0.0f;
        if (col < K) {
            const int c = col / PP, r = col - c * PP, i = r / P, j = r - i * P;
            const int px = (int)(row % g);
            const I t = row / g;
            const int py = (int)(t % g);
            const I b = t / g;
            const I xi = ((b * 3 + c) * img + (I)py * P + i) * img + (I)px * P + j;
            float pix = x[xi];
            if (delta) pix = pix + delta[xi];                 // pgd_train.py:32  data_clean + perturbation
            const float mu = c == 0 ? m0 : (c == 1 ? m1 : m2);
            const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            v = (pix - mu) / sd;                              // Normalize, …clip.py:116,254
        }
        A0[row * lda + col] = from_f32<T>(v);
    }
}

template <typename T>
int im2col_normalize(const float* x, const float* delta, int B, int img, int P, const float* mean3,
                     const float* std3, T* A0, long lda, int Kpad, hipStream_t s) {
    const long total = (long)B * (img / P) * (img / P) * Kpad;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    const bool small = total < (1L << 30) && (long)B * 3 * img * img < (1L << 30) &&
                       (long)B * (img / P) * (img / P) * lda < (1L << 30);
    if (small)
        hipLaunchKernelGGL((im2col_kernel<T, int>), dim3(grid), dim3(256), 0, s, x, delta, B, img, P,
                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], A0, lda, Kpad);
    else
        hipLaunchKernelGGL((im2col_kernel<T, long>), dim3(grid), dim3(256), 0, s, x, delta, B, img, P,
                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], A0, lda, Kpad);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int im2col_normalize<float>(const float*, const float*, int, int, int, const float*,
                                     const float*, float*, long, int, hipStream_t);
template int im2col_normalize<bf16_t>(const float*, const float*, int, int, int, const float*,
                                      const float*, bf16_t*, long, int, hipStream_t);

template <typename T, typename I>
__global__ void __launch_bounds__(256)
col2im_kernel(const T* __restrict__ dA0, long lda, int B, int img, int P, float s0, float s1,
              float s2, float* __restrict__ grad_x) {
    const int g = img / P, PP = P * P;
    const I total = (I)B * 3 * img * img;
    for (I idx = (I)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (I)gridDim.x * blockDim.x) {
        const int xx = (int)(idx % img);
        I t = idx / img;
        const int yy = (int)(t % img);
        t /= img;
        const int c = (int)(t % 3);
        const I b = t / 3;
        const int py = yy / P, i = yy - py * P, px = xx / P, j = xx - px * P;
        const I row = (b * g + py) * g + px;
        const int col = c * PP + i * P + j;
        const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
        grad_x[idx] = to_f32(dA0[row * lda + col]) / sd;
    }
}

template <typename T>
int col2im_grad(const T* dA0, long lda, int B, int img, int P, const float* std3, float* grad_x,
                hipStream_t s) {
    const long total = (long)B * 3 * img * img;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    const bool small = total < (1L << 30) && (long)B * (img / P) * (img / P) * lda < (1L << 30);
    if (small)
        hipLaunchKernelGGL((col2im_kernel<T, int>), dim3(grid), dim3(256), 0, s, dA0, lda, B, img, P, std3[0],
                           std3[1], std3[2], grad_x);
    else
        hipLaunchKernelGGL((col2im_kernel<T, long>), dim3(grid), dim3(256), 0, s, dA0, lda, B, img, P, std3[0],
                           std3[1], std3[2], grad_x);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int col2im_grad<float>(const float*, long, int, int, int, const float*, float*, hipStream_t);
template int col2im_grad<bf16_t>(const bf16_t*, long, int, int, int, const float*, float*,
                                 hipStream_t);

// =============================================================================================
// tokens = [cls ; patch_out] + pos ; x0 = ln_pre(tokens)     (Appendix B steps 2-3)
// =============================================================================================
template <typename T>
__global__ void __launch_bounds__(256)
embed_lnpre_fwd_kernel(const T* __restrict__ patch_out, long ldp, const float* __restrict__ cls,
                       const float* __restrict__ pos, const float* __restrict__ gamma,
                       const float* __restrict__ beta, float* __restrict__ x0, long ldx,
                       float* __restrict__ mean, float* __restrict__ rstd, int B, int S, int W,
                       float* __restrict__ tokens_out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B * S) return;
    const int b = row / S, sidx = row - b * S;
    const T* pr = patch_out + ((long)b * (S - 1) + (sidx - 1)) * ldp;
    float v[LN_MAXV][4];
    float sum = 0.0f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float a[4], p[4];
            if (sidx == 0) load4(cls + c, a); else load4(pr + c, a);
            load4(pos + (long)sidx * W + c, p);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] = a[e] + p[e]; sum += v[it][e]; }
            if (tokens_out) store4(tokens_out + (long)row * ldx + c, v[it]);
        }
    }
    const float mu = wave_sum(sum) / (float)W;
    float q = 0.0f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[it][e] - mu; q = fmaf(d, d, q); }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)W + 1e-5f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float g[4], bb[4], o[4];
            load4(gamma + c, g); load4(beta + c, bb);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[it][e] - mu) * rs * g[e] + bb[e];
            store4(x0 + (long)row * ldx + c, o);
        }
    }
}

template <typename T>
int embed_lnpre_fwd(const T* patch_out, long ldp, const float* cls, const float* pos,
                    const float* gamma, const float* beta, float* x0, long ldx, float* mean,
                    float* rstd, int B, int S, int W, hipStream_t s, float* tokens_out) {
    if (W % 4 != 0 || W > LN_MAXV * 256) return fail(RVLM_ERR_UNSUPPORTED, "embed: width");
    hipLaunchKernelGGL((embed_lnpre_fwd_kernel<T>), dim3(cdiv((long)B * S, 4)), dim3(256), 0, s,
                       patch_out, ldp, cls, pos, gamma, beta, x0, ldx, mean, rstd, B, S, W, tokens_out);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int embed_lnpre_fwd<float>(const float*, long, const float*, const float*, const float*,
                                    const float*, float*, long, float*, float*, int, int, int,
                                    hipStream_t, float*);

template <typename TP, typename T>
__global__ void __launch_bounds__(256)
embed_lnpre_bwd_kernel(const float* __restrict__ dx0, long lddx, const TP* __restrict__ patch_out,
                       long ldp, const float* __restrict__ pos, const float* __restrict__ gamma,
                       const float* __restrict__ mean, const float* __restrict__ rstd,
                       T* __restrict__ d_patch, long lddp, int B, int S, int W) {
    // one wave per patch token (CLS rows have no image gradient)
    const int prow = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (prow >= B * (S - 1)) return;
    const int b = prow / (S - 1), sidx = prow - b * (S - 1) + 1;
    const long row = (long)b * S + sidx;
    const float mu = mean[row], rs = rstd[row];
    float g[LN_MAXV][4], xh[LN_MAXV][4];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float d[4], a[4], p[4], gm[4];
            load4(dx0 + row * lddx + c, d);
            load4(patch_out + (long)prow * ldp + c, a);
            load4(pos + (long)sidx * W + c, p);
            load4(gamma + c, gm);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[it][e] = d[e] * gm[e];
                xh[it][e] = ((a[e] + p[e]) - mu) * rs;
                s1 += g[it][e];
                s2 = fmaf(g[it][e], xh[it][e], s2);
            }
        }
    }
    const float c1 = wave_sum(s1) / (float)W, c2 = wave_sum(s2) / (float)W;
#pragma unroll
    for (int it = 0; it < LN_MAXV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < W) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (g[it][e] - c1 - xh[it][e] * c2);
            store4(d_patch + (long)prow * lddp + c, o);
        }
    }
}

template <typename T>
int embed_lnpre_bwd(const float* dx0, long lddx, const float* patch_out, long ldp, const float* cls,
                    const float* pos, const float* gamma, const float* mean, const float* rstd,
                    T* d_patch, long lddp, int B, int S, int W, hipStream_t s) {
    (void)cls;
    if (W % 4 != 0 || W > LN_MAXV * 256) return fail(RVLM_ERR_UNSUPPORTED, "embed: width");
    hipLaunchKernelGGL((embed_lnpre_bwd_kernel<float, T>), dim3(cdiv((long)B * (S - 1), 4)),
                       dim3(256), 0, s, dx0, lddx, patch_out, ldp, pos, gamma, mean, rstd, d_patch,
                       lddp, B, S, W);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int embed_lnpre_bwd<float>(const float*, long, const float*, long, const float*,
                                    const float*, const float*, const float*, const float*, float*,
                                    long, int, int, int, hipStream_t);
template int embed_lnpre_bwd<bf16_t>(const float*, long, const float*, long, const float*,
                                     const float*, const float*, const float*, const float*,
                                     bf16_t*, long, int, int, int, hipStream_t);

// =============================================================================================
// F.normalize(dim=-1) forward / backward (…clip.py:255-256), one wave per sample
// =============================================================================================
__global__ void __launch_bounds__(256)
l2norm_fwd_kernel(const float* __restrict__ e, float* __restrict__ out, float* __restrict__ inv_norm,
                  int B, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float q = 0.0f;
    for (int c = lane; c < D; c += 64) { float v = e[(long)row * D + c]; q = fmaf(v, v, q); }
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    if (lane == 0) inv_norm[row] = 1.0f / nrm;
    for (int c = lane; c < D; c += 64) out[(long)row * D + c] = e[(long)row * D + c] / nrm;
}
__global__ void __launch_bounds__(256)
l2norm_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ e_raw,
                  const float* __restrict__ inv_norm, float* __restrict__ d_raw, int B, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float inv = inv_norm[row];
    float dot = 0.0f;
    for (int c = lane; c < D; c += 64)
        dot = fmaf(e_raw[(long)row * D + c] * inv, d_out[(long)row * D + c], dot);
    dot = wave_sum(dot);
    for (int c = lane; c < D; c += 64) {
        const float eh = e_raw[(long)row * D + c] * inv;
        d_raw[(long)row * D + c] = (d_out[(long)row * D + c] - eh * dot) * inv;
    }
}
int l2_normalize_fwd(const float* e, float* out, float* inv_norm, int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, e, out, inv_norm, B, D);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
int l2_normalize_bwd(const float* d_out, const float* e_raw, const float* inv_norm, float* d_raw,
                     int B, int D, hipStream_t s) {
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, d_out, e_raw, inv_norm,
                       d_raw, B, D);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// =============================================================================================
// Softmax over rows (fp32 attention path, scores materialised), one wave per row, in place
// =============================================================================================
__global__ void __launch_bounds__(256) softmax_fwd_kernel(float* __restrict__ s, long rows, int cols, int ld, float* __restrict__ lse2,
                                                          int lse_ld) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = s + row * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, r[c]);
    m = wave_max(m);
    float sum = 0.0f;
    for (int c = lane; c < cols; c += 64) { float e = expf(r[c] - m); r[c] = e; sum += e; }
    sum = wave_sum(sum);
    for (int c = lane; c < cols; c += 64) r[c] = r[c] / sum;
    if (lse2 && lane == 0) lse2[(row / cols) * lse_ld + row % cols] = (m + logf(sum)) * 1.4426950408889634f;
}
__global__ void __launch_bounds__(256)
softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows, int cols, int ld,
                   float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pr = p + row * ld;
    float* dr = dp + row * ld;
    float dot = 0.0f;
    for (int c = lane; c < cols; c += 64) dot = fmaf(pr[c], dr[c], dot);
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) dr[c] = pr[c] * (dr[c] - dot) * scale;
}
// Rows of 257 .. 260 columns (S = 257: the ViT-L/14 and ViT-B/16 score matrices, ld = 260), round 6: every element is read ONCE
// (16 bytes per lane: lane l holds columns 4 l .. 4 l + 3, lane 0 also the tail 256 .. 259) and written once, four rows per wave with
// their loads in flight together.  The kernels above walk a row three times through memory with 4-byte accesses: 470 / 405 us per
// launch on the [526 336, 260] matrices of ViT-L/14 at B = 128, 11 % of the fp32 / split-bf16 engines' step (profiles/
// r06_x3_kernel_stats.csv).  Same arithmetic per element; the row sums associate differently from the generic kernels' (a lane
// holds 4 consecutive columns instead of columns l, l + 64, ..: fp32, ~1 ulp of the row sum - like layernorm_fwd8 above).
template <bool BWD>
__global__ void __launch_bounds__(256)
softmax257_kernel(const float* __restrict__ p, float* __restrict__ s, long rows, int cols, int ld, float scale,
                  float* __restrict__ lse2 = nullptr, int lse_ld = 0) {
    constexpr int R = 4;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    const int lane = threadIdx.x & 63;
    if (row0 >= rows) return;
    float4 v[R], t[R], q[R], u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long row = min(row0 + r, rows - 1);
        float* sr = s + row * ld;
        v[r] = *(const float4*)(sr + 4 * lane);
        t[r] = lane == 0 ? *(const float4*)(sr + 256) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (BWD) {
            const float* pr = p + row * ld;
            q[r] = *(const float4*)(pr + 4 * lane);
            u[r] = lane == 0 ? *(const float4*)(pr + 256) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int ntail = cols - 256;      // 1 .. 4 valid tail columns (lane 0)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= rows) break;
        float* sr = s + (row0 + r) * ld;
        float a[4] = {v[r].x, v[r].y, v[r].z, v[r].w}, b[4] = {t[r].x, t[r].y, t[r].z, t[r].w};
        if (!BWD) {
            float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
            if (lane == 0)
                _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < ntail) m = fmaxf(m, b[e]);
            m = wave_max(m);
            float sum = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = expf(a[e] - m); sum += a[e]; }
            if (lane == 0)
                _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < ntail) { b[e] = expf(b[e] - m); sum += b[e]; }
            sum = wave_sum(sum);
            if (lse2 && lane == 0) lse2[((row0 + r) / cols) * lse_ld + (row0 + r) % cols] = (m + logf(sum)) * 1.4426950408889634f;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = a[e] / sum;
            if (lane == 0)
                _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < ntail) b[e] = b[e] / sum;
        } else {
            const float pa[4] = {q[r].x, q[r].y, q[r].z, q[r].w}, pb[4] = {u[r].x, u[r].y, u[r].z, u[r].w};
            float dot = 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dot = fmaf(pa[e], a[e], dot);
            if (lane == 0)
                _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < ntail) dot = fmaf(pb[e], b[e], dot);
            dot = wave_sum(dot);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = pa[e] * (a[e] - dot) * scale;
            if (lane == 0)
                _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < ntail) b[e] = pb[e] * (b[e] - dot) * scale;
        }
        *(float4*)(sr + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
        if (lane == 0)
            _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < ntail) sr[256 + e] = b[e];
    }
}
static bool softmax257_ok(const void* a, const void* b, int cols, int ld) {
    return cols > 256 && cols <= 260 && ld >= 260 && ld % 4 == 0 && (((size_t)a | (size_t)b) & 15) == 0;
}
int softmax_rows_fwd(float* s, long rows, int cols, int ld, hipStream_t st, float* lse2, int lse_ld) {
    if (softmax257_ok(s, s, cols, ld))
        hipLaunchKernelGGL((softmax257_kernel<false>), dim3(cdiv(rows, 16)), dim3(256), 0, st, nullptr, s, rows, cols, ld, 1.0f, lse2, lse_ld);
    else
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, s, rows, cols, ld, lse2, lse_ld);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
int softmax_rows_bwd(const float* p, float* dp, long rows, int cols, int ld, float scale, hipStream_t st) {
    if (softmax257_ok(p, dp, cols, ld))
        hipLaunchKernelGGL((softmax257_kernel<true>), dim3(cdiv(rows, 16)), dim3(256), 0, st, p, dp, rows, cols, ld, scale, (float*)nullptr, 0);
    else
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, p, dp, rows, cols, ld,
                       scale);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// =============================================================================================
// misc
// =============================================================================================
__global__ void __launch_bounds__(256)
convert_kernel(const float* __restrict__ src, long lds_, bf16_t* __restrict__ dst, long ldd,
               int rows, int cols, int transpose) {
    const long total = (long)rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        if (!transpose) {
            const long r = idx / cols, c = idx - r * cols;
            dst[r * ldd + c] = (bf16_t)src[r * lds_ + c];
        } else {  // write-coalesced: dst[c, r]
            const long c = idx / rows, r = idx - c * rows;
            dst[c * ldd + r] = (bf16_t)src[r * lds_ + c];
        }
    }
}
// fp32 [rows, cols] -> bf16 copy nk [rows, cols] AND transposed copy t [cols, rows] in one pass over 64x64 tiles (the
// weight refresh after every optimizer step: the element-wise version above spends 12 us per matrix, 4.7 ms per step).
__global__ void __launch_bounds__(256)
convert_pair_kernel(const float* __restrict__ src, long lds_, bf16_t* __restrict__ nk, long ld_nk,
                    bf16_t* __restrict__ t, long ld_t) {
    __shared__ unsigned short tile[64 * 66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 32 + wv * 8 + (lane >> 3), c8 = (lane & 7) * 8;
        const float* sp = src + (long)(r0 + row) * lds_ + c0 + c8;
        const float4 a = *(const float4*)sp, b = *(const float4*)(sp + 4);
        bf16x8 v;
        v[0] = (bf16_t)a.x; v[1] = (bf16_t)a.y; v[2] = (bf16_t)a.z; v[3] = (bf16_t)a.w;
        v[4] = (bf16_t)b.x; v[5] = (bf16_t)b.y; v[6] = (bf16_t)b.z; v[7] = (bf16_t)b.w;
        *(bf16x8*)(nk + (long)(r0 + row) * ld_nk + c0 + c8) = v;
        const unsigned short* u = (const unsigned short*)&v;
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[row * 66 + c8 + e] = u[e];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = pass * 32 + wv * 8 + (lane >> 3), rc = (lane & 7) * 8;
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = tile[(rc + i) * 66 + c];
        uint4 o;
        o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
        o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
        *(uint4*)(t + (long)(c0 + c) * ld_t + r0 + rc) = o;
    }
}
// returns false when the shape / leading dimensions do not fit the tiled kernel (caller falls back)
bool convert_f32_to_bf16_pair(const float* src, long lds_, bf16_t* nk, long ld_nk, bf16_t* t, long ld_t, int rows,
                              int cols, hipStream_t s) {
    if (rows % 64 != 0 || cols % 64 != 0 || lds_ % 4 != 0 || ld_nk % 8 != 0 || ld_t % 8 != 0) return false;
    if (((size_t)src | (size_t)nk | (size_t)t) & 15) return false;
    hipLaunchKernelGGL(convert_pair_kernel, dim3(cols / 64, rows / 64), dim3(256), 0, s, src, lds_, nk, ld_nk, t, ld_t);
    return true;
}

int convert_f32_to_bf16(const float* src, long lds_, bf16_t* dst, long ldd, int rows, int cols,
                        int transpose, hipStream_t s) {
    const long total = (long)rows * cols;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(convert_kernel, dim3(grid), dim3(256), 0, s, src, lds_, dst, ldd, rows, cols,
                       transpose);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

__global__ void __launch_bounds__(256)
scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n, float alpha) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        dst[i] = alpha * src[i];
}
int scale_copy_f32(const float* src, float* dst, size_t n, float alpha, hipStream_t s) {
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(scale_copy_kernel, dim3(grid), dim3(256), 0, s, src, dst, n, alpha);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ dst, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        dst[i] = v;
}
int fill_f32(float* dst, size_t n, float v, hipStream_t s) {
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, s, dst, n, v);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

}  // namespace rvlm
