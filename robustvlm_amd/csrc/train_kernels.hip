// Kernels of the outer training step (train_one_epoch's
// loss.backward() / optimizer.step(), train/adversarial_training_clip.py:356-364): weight-gradient
// support (transposes feeding the NT MFMA GEMM, column sums for biases, LayerNorm affine gradients,
// positional / class-embedding gradients) and the AdamW update.  HBM-bound helpers; the wgrad FLOPs
// themselves run on gemm_bf16_nt (contraction over the token dimension on transposed operands).
#include "kernels.h"
#include <type_traits>

namespace rvlm {

constexpr int RED_NCH = 128;

// out[c, r] = in[r, c] for r < R, zero for R <= r < Rp (the GEMM's K padding).  64x64 tiles via LDS.
template <typename T>
__global__ void __launch_bounds__(256)
transpose_kernel(const T* __restrict__ in, long ldi, int R, int C, T* __restrict__ out, long ldo, int Rp) {
    __shared__ T tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty * 16 + i, c = c0 + tx;
        tile[ty * 16 + i][tx] = (r < R && c < C) ? in[(long)r * ldi + c] : from_f32<T>(0.0f);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty * 16 + i, r = r0 + tx;
        if (c < C && r < Rp) out[(long)c * ldo + r] = tile[tx][ty * 16 + i];
    }
}
template <typename T>
int transpose_pad(const T* in, long ldi, int R, int C, T* out, long ldo, int Rp, hipStream_t s) {
    hipLaunchKernelGGL((transpose_kernel<T>), dim3(cdiv(C, 64), cdiv(Rp, 64)), dim3(256), 0, s, in, ldi, R, C, out,
                       ldo, Rp);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int transpose_pad<bf16_t>(const bf16_t*, long, int, int, bf16_t*, long, int, hipStream_t);

// Split-K layout for the persistent weight-gradient GEMM: out[((r / Kc) * C + c) * Kc + r % Kc] = in[r, c] for r < R,
// zero for R <= r < Rp (Rp = splits * Kc).  C % 64 == 0, Kc % 64 == 0 (a 64-row tile never straddles two chunks).
// Every global access is 16 B per lane, 128 B per row (the 2-byte-per-lane version above reaches 2.4 TB/s).
// colpart != null: also emits the column sums of the tile's 64 rows, colpart[blockIdx.y][c] (bias gradient).
__global__ void __launch_bounds__(256)
transpose_split_kernel(const bf16_t* __restrict__ in, long ldi, int R, int C, bf16_t* __restrict__ out, int Kc,
                       float* __restrict__ colpart) {
    __shared__ unsigned tile[64 * 33];   // 64 rows x 64 bf16, row pitch 33 dwords
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 32 + wv * 8 + (lane >> 3), r = r0 + row;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r < R) v = *(const uint4*)(in + (long)r * ldi + c0 + (lane & 7) * 8);
        unsigned* d = tile + row * 33 + (lane & 7) * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const unsigned short* t16 = (const unsigned short*)tile;
    const int split = r0 / Kc;
    const long obase = (long)split * C * Kc + (r0 - split * Kc);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int c = pass * 32 + wv * 8 + (lane >> 3), rc = (lane & 7) * 8;
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = t16[(rc + i) * 66 + c];
        uint4 o;
        o.x = e[0] | ((unsigned)e[1] << 16); o.y = e[2] | ((unsigned)e[3] << 16);
        o.z = e[4] | ((unsigned)e[5] << 16); o.w = e[6] | ((unsigned)e[7] << 16);
        *(uint4*)(out + obase + (long)(c0 + c) * Kc + rc) = o;
        if (colpart) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += __uint_as_float((unsigned)e[i] << 16);
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4);
            if ((lane & 7) == 0) colpart[(long)blockIdx.y * C + c0 + c] = sum;
        }
    }
}
// partial column sums -> out (+)=: 16 columns x 16 partial lanes per workgroup
__global__ void __launch_bounds__(256)
reduce_partials16_kernel(const float* __restrict__ partial, int nch, int C, float* __restrict__ out, int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, kl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    float a0 = 0.0f, a1 = 0.0f;
    if (c < C) {
        int k = kl;
        for (; k + 16 < nch; k += 32) {
            a0 += partial[(long)k * C + c];
            a1 += partial[(long)(k + 16) * C + c];
        }
        if (k < nch) a0 += partial[(long)k * C + c];
    }
    red[kl][cl] = a0 + a1;
    __syncthreads();
    if (kl == 0 && c < C) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += red[i][cl];
        out[c] = (accumulate ? out[c] : 0.0f) + acc;
    }
}
// transposes one wgrad operand into the split layout; dbias != null: dbias[c] (+)= sum_r in[r, c] on the way
int transpose_split(const bf16_t* in, long ldi, int R, int C, bf16_t* out, int Kc, int splits, float* dbias,
                    int accumulate, float* red, size_t red_floats, hipStream_t s) {
    if (C % 64 != 0 || Kc % 64 != 0 || ldi % 8 != 0) return fail(RVLM_ERR_ARG, "transpose_split: C, Kc % 64, ldi % 8");
    const int rt = splits * Kc / 64;
    float* part = nullptr;
    if (dbias) {
        if (!red || (size_t)rt * C > red_floats) return fail(RVLM_ERR_STATE, "transpose_split: no reduce scratch");
        part = red;
    }
    hipLaunchKernelGGL(transpose_split_kernel, dim3(C / 64, rt), dim3(256), 0, s, in, ldi, R, C, out, Kc, part);
    RVLM_CHECK_LAUNCH();
    if (dbias) {
        hipLaunchKernelGGL(reduce_partials16_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, part, rt, C, dbias, accumulate);
        RVLM_CHECK_LAUNCH();
    }
    return RVLM_OK;
}

// ---- column reductions over the token dimension (bias and LayerNorm-affine gradients) ---------------
// Two deterministic passes: grid (C/64, NCH) blocks each reduce a 64-column x (R/NCH)-row slab into
// partial[chunk][c]; a second tiny kernel sums the NCH partials.  (A single pass with C/64 workgroups
// left 94 % of the chip idle: 3.1 ms per call at M = 32 896.)
template <typename T>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const T* __restrict__ in, long ld, int R, int C, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float acc = 0.0f;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += 4) acc += to_f32(in[(long)r * ld + c]);
    red[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int t = threadIdx.x;
        partial[(long)blockIdx.y * C + c] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    }
}
template <typename T>
__global__ void __launch_bounds__(256)
ln_param_partial_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                        const float* __restrict__ mean, const float* __restrict__ rstd, int R, int C,
                        float* __restrict__ pg, float* __restrict__ pb) {
    __shared__ float rg[4][64], rb[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float ag = 0.0f, ab = 0.0f;
    if (c < C)
        for (int r = r0 + rl; r < r1; r += 4) {
            const float d = to_f32(dy[(long)r * lddy + c]);
            const float xh = (x[(long)r * ldx + c] - mean[r]) * rstd[r];
            ag = fmaf(d, xh, ag);
            ab += d;
        }
    rg[rl][threadIdx.x & 63] = ag;
    rb[rl][threadIdx.x & 63] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int t = threadIdx.x;
        pg[(long)blockIdx.y * C + c] = (rg[0][t] + rg[1][t]) + (rg[2][t] + rg[3][t]);
        pb[(long)blockIdx.y * C + c] = (rb[0][t] + rb[1][t]) + (rb[2][t] + rb[3][t]);
    }
}
// bf16 rows of C % 256 == 0 columns, 16-B aligned (the linear layers' dY next to the copy-free weight-gradient GEMM): 16 B per
// lane - a workgroup = 32 column groups of 8 x 8 row lanes, each lane 8 fp32 sums over its rows, LDS sum over the row lanes.
__global__ void __launch_bounds__(256)
colsum8_partial_kernel(const bf16_t* __restrict__ in, long ld, int R, int C, float* __restrict__ partial) {
    __shared__ float red[8][256 + 8];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + cg * 8;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 8) {
        const uint4 v = *(const uint4*)(in + (long)r * ld + c);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] += __uint_as_float(u[i] << 16);
            acc[2 * i + 1] += __uint_as_float(u[i] & 0xffff0000u);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][cg * 8 + i] = acc[i];
    __syncthreads();
    const int t = threadIdx.x;
    partial[(long)blockIdx.y * C + blockIdx.x * 256 + t] =
        ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) + ((red[4][t] + red[5][t]) + (red[6][t] + red[7][t]));
}
template <typename T>
int colsum(const T* in, long ld, int R, int C, float* out, int accumulate, float* red, size_t red_floats, hipStream_t s) {
    const int nch = R >= 4096 ? RED_NCH : (R >= 256 ? 16 : 1);
    if (!red || (size_t)nch * C > red_floats) return fail(RVLM_ERR_STATE, "colsum: no reduce scratch");
    if (std::is_same<T, bf16_t>::value && C % 256 == 0 && ld % 8 == 0 && ((size_t)in & 15) == 0) {
        hipLaunchKernelGGL(colsum8_partial_kernel, dim3(C / 256, nch), dim3(256), 0, s, (const bf16_t*)in, ld, R, C, red);
        RVLM_CHECK_LAUNCH();
        hipLaunchKernelGGL(reduce_partials16_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, red, nch, C, out, accumulate);
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
    hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3(cdiv(C, 64), nch), dim3(256), 0, s, in, ld, R, C, red);
    RVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials16_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, red, nch, C, out, accumulate);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int colsum<float>(const float*, long, int, int, float*, int, float*, size_t, hipStream_t);
template int colsum<bf16_t>(const bf16_t*, long, int, int, float*, int, float*, size_t, hipStream_t);

// LayerNorm affine gradients: dgamma[c] (+)= sum_r dy[r,c] * (x[r,c]-mean[r])*rstd[r]; dbeta[c] (+)= sum_r dy[r,c]
template <typename T>
int ln_param_grad(const T* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd, int R,
                  int C, float* dgamma, float* dbeta, int accumulate, float* red, size_t red_floats, hipStream_t s) {
    const int nch = R >= 4096 ? RED_NCH : (R >= 256 ? 16 : 1);
    if (!red || (size_t)2 * nch * C > red_floats) return fail(RVLM_ERR_STATE, "ln_param_grad: no reduce scratch");
    float* pg = red;
    float* pb = red + (size_t)nch * C;
    hipLaunchKernelGGL((ln_param_partial_kernel<T>), dim3(cdiv(C, 64), nch), dim3(256), 0, s, dy, lddy, x, ldx, mean,
                       rstd, R, C, pg, pb);
    RVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials16_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, pg, nch, C, dgamma, accumulate);
    RVLM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials16_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, pb, nch, C, dbeta, accumulate);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int ln_param_grad<float>(const float*, long, const float*, long, const float*, const float*, int, int,
                                  float*, float*, int, float*, size_t, hipStream_t);
template int ln_param_grad<bf16_t>(const bf16_t*, long, const float*, long, const float*, const float*, int, int,
                                   float*, float*, int, float*, size_t, hipStream_t);

// positional-embedding gradient: dpos[s, c] (+)= sum_b dtok[b*S + s, c]; class embedding = row s = 0
__global__ void __launch_bounds__(256)
pos_grad_kernel(const float* __restrict__ dtok, long ld, int B, int S, int W, float* __restrict__ dpos,
                float* __restrict__ dcls, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)S * W) return;
    const int sidx = (int)(idx / W), c = (int)(idx - (long)sidx * W);
    float acc = 0.0f;
    for (int b = 0; b < B; ++b) acc += dtok[((long)b * S + sidx) * ld + c];
    dpos[idx] = (accumulate ? dpos[idx] : 0.0f) + acc;
    if (sidx == 0) dcls[c] = (accumulate ? dcls[c] : 0.0f) + acc;
}
int pos_cls_grad(const float* dtok, long ld, int B, int S, int W, float* dpos, float* dcls, int accumulate,
                 hipStream_t s) {
    hipLaunchKernelGGL(pos_grad_kernel, dim3(cdiv((long)S * W, 256)), dim3(256), 0, s, dtok, ld, B, S, W, dpos, dcls,
                       accumulate);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// d_patch[b*(S-1) + s-1, :] = (T) dtok[b*S + s, :]   (drop the CLS rows, cast for the conv wgrad / dgrad GEMM)
template <typename T>
__global__ void __launch_bounds__(256)
gather_patch_rows_kernel(const float* __restrict__ dtok, long ld, int B, int S, int W, T* __restrict__ d_patch,
                         long ldp) {
    const long total = (long)B * (S - 1) * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long prow = idx / W;
        const int c = (int)(idx - prow * W);
        const long b = prow / (S - 1), sidx = prow - b * (S - 1) + 1;
        d_patch[prow * ldp + c] = from_f32<T>(dtok[(b * S + sidx) * ld + c]);
    }
}
template <typename T>
int gather_patch_rows(const float* dtok, long ld, int B, int S, int W, T* d_patch, long ldp, hipStream_t s) {
    const long total = (long)B * (S - 1) * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL((gather_patch_rows_kernel<T>), dim3(grid), dim3(256), 0, s, dtok, ld, B, S, W, d_patch, ldp);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template int gather_patch_rows<float>(const float*, long, int, int, int, float*, long, hipStream_t);
template int gather_patch_rows<bf16_t>(const float*, long, int, int, int, bf16_t*, long, hipStream_t);

// torch.optim.AdamW single-tensor update (decoupled weight decay), fp32 master weights:
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
             size_t n, float decay, float w1, float b2, float w2, float eps, float step_size, float bc2_sqrt,
             float grad_scale) {
    // torch.optim.AdamW, single-tensor path: every scalar below is a Python double in torch, rounded to fp32 when it
    // meets the fp32 tensor - the host computes them in double and passes the rounded values
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        float pi = p[i];
        pi = pi * decay;                                             // param.mul_(1 - lr * weight_decay)
        const float mi = m[i] + (gi - m[i]) * w1;                    // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + w2 * gi * gi;                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);                          // param.addcdiv_(exp_avg, denom, value=-step_size)
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

}  // namespace rvlm

using namespace rvlm;

extern "C" int rvlm_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                               double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                               float grad_scale, rvlm_stream_t stream) {
    RVLM_REQUIRE(params && grads && exp_avg && exp_avg_sq && step >= 1, "rvlm_adamw_step: bad arguments");
    // bias corrections in double, like torch's Python scalars (fp32 powf was a ~1e-5 relative step error early on)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
    size_t blocks = (n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, n, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)bc2_sqrt, grad_scale);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
