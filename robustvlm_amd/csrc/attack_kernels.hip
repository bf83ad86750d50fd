// L-inf perturb / project / clamp kernels of the PGD and APGD loops (HBM-bound, bit-exact).
//
// Replaces the ~12 elementwise kernels + 4 reductions + 4 host syncs per iteration of
// train/pgd_train.py:38-63 with ONE pass (28 B/element: read g, delta, v, x; write delta, v, x_adv)
// and the ~25 small kernels + nonzero() syncs of train/apgd_train.py:205-229,301-355 with three.
// Arithmetic follows SURVEY.md Appendix A exactly: one IEEE fp32 rounding per reference op, no FMA
// contraction (this file is built with -ffp-contract=off).  NaN / range asserts become bits of a
// device flag word that the host reads once after the loop.
#include "common.h"

namespace rvlm {

__device__ __forceinline__ float sgnf(float a) {  // torch.sign: sign(+-0)=0, sign(NaN)=0
    return (a > 0.0f) ? 1.0f : ((a < 0.0f) ? -1.0f : 0.0f);
}
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
check_range_kernel(const float* __restrict__ x, size_t n, int32_t* flags) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        float v = x[i];
        bad |= !(v < 1.000001f && v > -1e-6f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, RVLM_FLAG_INPUT_RANGE);
}

template <bool VEC>
__global__ void __launch_bounds__(256)
pgd_linf_update_kernel(const float* __restrict__ x, const float* __restrict__ g,
                       float* __restrict__ delta, float* __restrict__ vel, size_t n, float eps,
                       float step, float mom, int mode_max, float* __restrict__ x_adv_out,
                       int32_t* flags) {
    int f = 0;
    constexpr int V = VEC ? 4 : 1;
    const size_t nv = n / V;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (size_t)gridDim.x * blockDim.x) {
        float xv[V], gv[V], dv[V], vv[V], xa[V];
        if (VEC) {
            *(float4*)xv = ((const float4*)x)[i];
            *(float4*)gv = ((const float4*)g)[i];
            *(float4*)dv = ((const float4*)delta)[i];
            *(float4*)vv = ((const float4*)vel)[i];
        } else {
            xv[0] = x[i]; gv[0] = g[i]; dv[0] = delta[i]; vv[0] = vel[i];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float gi = gv[e];
            if (gi != gi) { gi = 0.0f; f |= RVLM_FLAG_NAN_GRAD; }      // pgd_train.py:40-42
            float s = sgnf(gi);                                        // utils.py:21
            float v = sgnf(mom * vv[e] + s);                           // :46-47
            float sv = step * v;
            float d = mode_max ? (dv[e] + sv) : (dv[e] - sv);          // :49-52
            d = fminf(fmaxf(d, -eps), eps);                            // :56 project_perturbation
            float a = clamp01(xv[e] + d);
            d = a - xv[e];                                             // :57-59
            if (d != d) f |= RVLM_FLAG_NAN_DELTA;                      // :60
            float s2 = xv[e] + d;                                      // :61-63 and :68
            if (!(s2 < 1.000001f && s2 > -1e-6f)) f |= RVLM_FLAG_ADV_RANGE;
            dv[e] = d; vv[e] = v; xa[e] = s2;
        }
        if (VEC) {
            ((float4*)delta)[i] = *(float4*)dv;
            ((float4*)vel)[i] = *(float4*)vv;
            if (x_adv_out) ((float4*)x_adv_out)[i] = *(float4*)xa;
        } else {
            delta[i] = dv[0]; vel[i] = vv[0];
            if (x_adv_out) x_adv_out[i] = xa[0];
        }
    }
    if (flags) {
        int wf = f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wf |= __shfl_xor(wf, o, 64);
        if (wf && (threadIdx.x & 63) == 0) atomicOr(flags, wf);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// L2 branch of pgd() (train/pgd_train.py:38-63 with vlm_eval/attacks/utils.py:12-14,22-26): the per-sample norms make
// it three dependent reductions, so ONE workgroup owns one sample and walks its pixels four times (the sample's four
// tensors are 2.4 MB: the later passes are L2-cache reads):
//   g = NaN -> 0;  g /= max(|g|_2, 1e-12)                         F.normalize(grad.view(bs, -1), p=2, dim=1)
//   v = mom * v + g;  v /= max(|v|_2, 1e-12)                      momentum, normalize again
//   d = d +- step * v;  d *= eps / (|d|_2 + 1e-7) if |d|_2 > eps  torch.renorm(d, p=2, dim=0, maxnorm=eps)
//   d = clamp(x + d, 0, 1) - x
// Sums of squares are fp32 in a fixed order (lane-strided partial sums, wave shuffles, 16 wave partials in LDS):
// deterministic, but not torch's reduction order - parity with the reference is to fp32 rounding of the norms, not
// bit-exact like the L-inf branch.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                                   // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}

__global__ void __launch_bounds__(1024)
pgd_l2_update_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ delta,
                     float* __restrict__ vel, size_t n_per, float eps, float step, float mom, int mode_max,
                     float* __restrict__ x_adv_out, int32_t* flags) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * n_per;
    const float* xs = x + base;
    const float* gs = g + base;
    float* ds = delta + base;
    float* vs = vel + base;
    int f = 0;
    float acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        float gi = gs[i];
        if (gi != gi) { gi = 0.0f; f |= RVLM_FLAG_NAN_GRAD; }
        acc = fmaf(gi, gi, acc);
    }
    const float gn = fmaxf(sqrtf(block_sum_1024(acc, red)), 1e-12f);
    acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        float gi = gs[i];
        if (gi != gi) gi = 0.0f;
        const float v = mom * vs[i] + gi / gn;
        vs[i] = v;                                      // un-normalised for now (same thread re-reads it below)
        acc = fmaf(v, v, acc);
    }
    const float vn = fmaxf(sqrtf(block_sum_1024(acc, red)), 1e-12f);
    acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        const float v = vs[i] / vn;
        vs[i] = v;
        const float sv = step * v;
        const float d = mode_max ? (ds[i] + sv) : (ds[i] - sv);
        ds[i] = d;
        acc = fmaf(d, d, acc);
    }
    const float dn = sqrtf(block_sum_1024(acc, red));
    const float sc = dn > eps ? eps / (dn + 1e-7f) : 1.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        const float xv = xs[i];
        float d = ds[i] * sc;
        d = clamp01(xv + d) - xv;
        if (d != d) f |= RVLM_FLAG_NAN_DELTA;
        const float s2 = xv + d;
        if (!(s2 < 1.000001f && s2 > -1e-6f)) f |= RVLM_FLAG_ADV_RANGE;
        ds[i] = d;
        if (x_adv_out) x_adv_out[base + i] = s2;
    }
    if (flags) {
        int wf = f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wf |= __shfl_xor(wf, o, 64);
        if (wf && (threadIdx.x & 63) == 0) atomicOr(flags, wf);
    }
}

// L2 branch of the APGD step (train/apgd_train.py:231-254), one workgroup per sample, three per-sample norms:
//   grad2 = x_adv - x_adv_old;  x_adv_old = x_adv
//   z = x_adv + (step * grad) / (|grad|_2 + 1e-12)
//   z = clamp(x + (z - x) / (|z - x|_2 + 1e-12) * min(eps, |z - x|_2), 0, 1)
//   u = x_adv + (z - x_adv) * a + grad2 * (1 - a)
//   x_adv = clamp(x + (u - x) / (|u - x|_2 + 1e-12) * min(eps, |u - x|_2), 0, 1)
// z is recomputed instead of stored (x_adv and x_adv_old are rewritten only in the third pass).  fp32 sums in this
// kernel's own fixed order: equal to the reference to fp32 rounding of the norms.
__global__ void __launch_bounds__(1024)
apgd_l2_step_kernel(const float* __restrict__ x, float* __restrict__ x_adv, float* __restrict__ x_adv_old,
                    const float* __restrict__ grad, const float* __restrict__ step, float a, float one_minus_a,
                    float eps, size_t n_per) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * n_per;
    const float* xs = x + base;
    const float* gs = grad + base;
    float* xa = x_adv + base;
    float* xo = x_adv_old + base;
    const float st = step[blockIdx.x];
    float acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) acc = fmaf(gs[i], gs[i], acc);
    const float gd = sqrtf(block_sum_1024(acc, red)) + 1e-12f;
    acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        const float z = xa[i] + (st * gs[i]) / gd;
        const float d = z - xs[i];
        acc = fmaf(d, d, acc);
    }
    const float n1 = sqrtf(block_sum_1024(acc, red));
    const float d1 = n1 + 1e-12f, m1 = fminf(eps, n1);
    acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        const float xv = xs[i], av = xa[i];
        float z = av + (st * gs[i]) / gd;
        z = clamp01(xv + (z - xv) / d1 * m1);
        const float g2 = av - xo[i];
        const float u = (av + (z - av) * a) + g2 * one_minus_a;
        xo[i] = av;
        xa[i] = u;                                        // finished in the last pass
        const float d = u - xv;
        acc = fmaf(d, d, acc);
    }
    const float n2 = sqrtf(block_sum_1024(acc, red));
    const float d2 = n2 + 1e-12f, m2 = fminf(eps, n2);
    for (size_t i = threadIdx.x; i < n_per; i += 1024) {
        const float xv = xs[i];
        xa[i] = clamp01(xv + (xa[i] - xv) / d2 * m2);
    }
}

// ---- the standalone helpers of vlm_eval/attacks/utils.py:8-26 as device kernels (robustvlm_amd/attack_utils.py) ----
__global__ void __launch_bounds__(256)
ew_clamp_kernel(const float* __restrict__ x, float lo, float hi, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        out[i] = (v != v) ? v : fminf(fmaxf(v, lo), hi);         // torch.clamp propagates NaN
    }
}
__global__ void __launch_bounds__(256)
ew_sign_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = sgnf(x[i]);                                      // sign(+-0) = 0, sign(NaN) = 0 like torch.sign
}
// mode 0: F.normalize(x.view(B, -1), p=2, dim=1): x / max(|x|_2, 1e-12);  mode 1: torch.renorm(x, 2, 0, maxnorm):
// rows with |x|_2 > maxnorm scaled by maxnorm / (|x|_2 + 1e-7)
__global__ void __launch_bounds__(1024)
row_l2_scale_kernel(const float* __restrict__ x, size_t n_per, int mode, float maxnorm, float* __restrict__ out) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * n_per;
    float acc = 0.0f;
    for (size_t i = threadIdx.x; i < n_per; i += 1024) acc = fmaf(x[base + i], x[base + i], acc);
    const float nrm = sqrtf(block_sum_1024(acc, red));
    if (mode == 0) {
        const float d = fmaxf(nrm, 1e-12f);
        for (size_t i = threadIdx.x; i < n_per; i += 1024) out[base + i] = x[base + i] / d;
    } else {
        const float sc = nrm > maxnorm ? maxnorm / (nrm + 1e-7f) : 1.0f;
        for (size_t i = threadIdx.x; i < n_per; i += 1024) out[base + i] = x[base + i] * sc;
    }
}

// one block row per sample chunk: blockIdx.y = sample
__global__ void __launch_bounds__(256)
apgd_linf_step_kernel(const float* __restrict__ x, float* __restrict__ x_adv,
                      float* __restrict__ x_adv_old, const float* __restrict__ grad,
                      const float* __restrict__ step, float a, float one_minus_a, float eps,
                      size_t n_per) {
    const int b = blockIdx.y;
    const float st = step[b];
    const size_t base = (size_t)b * n_per;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_per;
         j += (size_t)gridDim.x * blockDim.x) {
        size_t i = base + j;
        float xa = x_adv[i], xo = x_adv_old[i], xc = x[i];
        float grad2 = xa - xo;                                   // apgd_train.py:206
        float lo = xc - eps, hi = xc + eps;
        float z = xa + st * sgnf(grad[i]);                       // :213
        z = clamp01(fminf(fmaxf(z, lo), hi));                    // :214-221
        float t1 = (z - xa) * a;
        float t2 = grad2 * one_minus_a;
        float u = (xa + t1) + t2;                                // :225
        u = clamp01(fminf(fmaxf(u, lo), hi));                    // :222-229
        x_adv_old[i] = xa;                                       // :207
        x_adv[i] = u;
    }
}

__global__ void apgd_controller_kernel(int i, int B, int n_iter, int k, int do_check, double rho,
                                       const float* __restrict__ loss_i,
                                       const uint8_t* __restrict__ pred, float* loss_steps,
                                       float* loss_best, float* loss_best_last_check,
                                       float* reduced_last_check, float* step, uint8_t* acc,
                                       uint8_t* f_notpred, uint8_t* f_improved,
                                       uint8_t* f_reduced) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        float y1 = loss_i[b];
        uint8_t p = pred[b];
        acc[b] = acc[b] < p ? acc[b] : p;                        // apgd_train.py:302
        f_notpred[b] = p ? 0 : 1;                                // :304-305
        loss_steps[(size_t)i * B + b] = y1;                      // :322
        int imp = y1 > loss_best[b];                             // :323
        float lb = imp ? y1 : loss_best[b];
        loss_best[b] = lb;
        f_improved[b] = (uint8_t)imp;
        int red = 0;
        if (do_check) {                                          // :331
            float t = 0.0f;
            for (int c = 0; c < k; ++c) {                        // check_oscillation :117-122
                int r0 = i - c, r1 = i - c - 1;
                if (r0 < 0) r0 += n_iter;                        // negative index wraps
                if (r1 < 0) r1 += n_iter;
                float l0 = (r0 == i) ? y1 : loss_steps[(size_t)r0 * B + b];
                float l1 = loss_steps[(size_t)r1 * B + b];
                t += (l0 > l1) ? 1.0f : 0.0f;
            }
            float thr = (float)((double)k * rho);               // `k * k3` in double, then * ones_like(t) (fp32)
            float osc = (t <= thr) ? 1.0f : 0.0f;
            float noimp = (1.0f - reduced_last_check[b]) *
                          ((loss_best_last_check[b] >= lb) ? 1.0f : 0.0f);   // :337-338
            float r = fmaxf(osc, noimp);
            reduced_last_check[b] = r;
            loss_best_last_check[b] = lb;
            if (r > 0.0f) { step[b] = step[b] / 2.0f; red = 1; }            // :346-348
        }
        f_reduced[b] = (uint8_t)red;
    }
}

__global__ void __launch_bounds__(256)
apgd_select_kernel(float* __restrict__ x_adv, float* __restrict__ grad, float* __restrict__ x_best,
                   float* __restrict__ grad_best, float* __restrict__ x_best_adv,
                   const uint8_t* __restrict__ f_notpred, const uint8_t* __restrict__ f_improved,
                   const uint8_t* __restrict__ f_reduced, size_t n_per) {
    const int b = blockIdx.y;
    const bool np = f_notpred[b], im = f_improved[b], rd = f_reduced[b];
    if (!(np || im || rd)) return;
    const size_t base = (size_t)b * n_per;
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_per;
         j += (size_t)gridDim.x * blockDim.x) {
        size_t i = base + j;
        float xa = x_adv[i], gr = grad[i];
        if (np) x_best_adv[i] = xa;                              // apgd_train.py:305
        float xb, gb;
        if (im) { xb = xa; gb = gr; x_best[i] = xb; grad_best[i] = gb; }   // :324-325
        else if (rd) { xb = x_best[i]; gb = grad_best[i]; }
        if (rd) { x_adv[i] = xb; grad[i] = gb; }                 // :351-352
    }
}

// per-sample max|t| then x + eps * (t / (max + 1e-12))   (autopgd_base.py:180-183, 210-214)
__global__ void __launch_bounds__(256)
linf_random_start_kernel(const float* __restrict__ x, const float* __restrict__ t, float eps,
                         size_t n_per, float* __restrict__ x_adv) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const size_t base = (size_t)b * n_per;
    float m = 0.0f;
    for (size_t j = threadIdx.x; j < n_per; j += blockDim.x) m = fmaxf(m, fabsf(t[base + j]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float den = m + 1e-12f;
    for (size_t j = threadIdx.x; j < n_per; j += blockDim.x) {
        float q = t[base + j] / den;
        float e = eps * 1.0f;            // eps * ones_like(x)
        x_adv[base + j] = x[base + j] + e * q;
    }
}

// L2 random start of APGDAttack (autopgd_base.py:184-185, 215-218): x + eps * (t / (|t|_2 + 1e-12)), t ~ N(0, 1) drawn by the
// caller; one workgroup per sample, the norm a deterministic fp32 sum in this kernel's own order
__global__ void __launch_bounds__(1024)
l2_random_start_kernel(const float* __restrict__ x, const float* __restrict__ t, float eps, size_t n_per,
                       float* __restrict__ x_adv) {
    __shared__ float red[16];
    const size_t base = (size_t)blockIdx.x * n_per;
    float acc = 0.0f;
    for (size_t j = threadIdx.x; j < n_per; j += 1024) acc = fmaf(t[base + j], t[base + j], acc);
    const float den = sqrtf(block_sum_1024(acc, red)) + 1e-12f;
    for (size_t j = threadIdx.x; j < n_per; j += 1024) x_adv[base + j] = x[base + j] + eps * (t[base + j] / den);
}

static inline int ew_grid(size_t n, int per_thread = 4) {
    size_t blocks = (n / per_thread + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest
    return (int)blocks;
}

}  // namespace rvlm

using namespace rvlm;

// ---- Square Attack, L-inf (autoattack/square.py:256-263, 288-291) --------------------------------------------------
// One query: candidates for the n_active still-robust images idx[a], written compactly (the model runs on them next):
//   x_new[a] = clamp(min(max(x_best[idx[a]] + window, x[idx[a]] - eps), x[idx[a]] + eps), 0, 1)
// window = two_eps * sign[c] on rows [vh, vh+s) x columns [vw, vw+s) of every channel c, 0 elsewhere - ONE window per
// query for the whole batch, as in the reference.  Same fp32 operations in the same order (no contraction).
__global__ void __launch_bounds__(256)
square_linf_propose_kernel(const float* __restrict__ x, const float* __restrict__ x_best, const long long* __restrict__ idx,
                           int C, int H, int W, int vh, int vw, int s, float eps, float two_eps,
                           const float* __restrict__ sign, float* __restrict__ x_new) {
    const long n_img = (long)C * H * W;
    const long src = (long)idx[blockIdx.y] * n_img, dst = (long)blockIdx.y * n_img;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_img; e += (long)gridDim.x * blockDim.x) {
        const int col = (int)(e % W), row = (int)((e / W) % H), c = (int)(e / ((long)W * H));
        const bool in = row >= vh && row < vh + s && col >= vw && col < vw + s;
        const float w = in ? two_eps * sign[c] : 0.0f;
        const float xv = x[src + e];
        float v = x_best[src + e] + w;
        v = fminf(fmaxf(v, xv - eps), xv + eps);
        x_new[dst + e] = fminf(fmaxf(v, 0.0f), 1.0f);
    }
}
// accepted candidates replace the incumbent: x_best[idx[a]] = x_new[a] where take[a] != 0
__global__ void __launch_bounds__(256)
square_accept_kernel(float* __restrict__ x_best, const float* __restrict__ x_new, const long long* __restrict__ idx,
                     const float* __restrict__ take, long n_img) {
    if (take[blockIdx.y] == 0.0f) return;
    const long dst = (long)idx[blockIdx.y] * n_img, src = (long)blockIdx.y * n_img;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_img; e += (long)gridDim.x * blockDim.x)
        x_best[dst + e] = x_new[src + e];
}

extern "C" int rvlm_check_image_range(const float* x, size_t n, int32_t* flags,
                                      rvlm_stream_t stream) {
    RVLM_REQUIRE(x && flags, "rvlm_check_image_range: null pointer");
    hipLaunchKernelGGL(check_range_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, (hipStream_t)stream,
                       x, n, flags);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_pgd_linf_update(const float* x, const float* grad, float* delta,
                                    float* velocity, size_t n, float eps, float stepsize,
                                    float momentum, int mode_max, float* x_adv_out, int32_t* flags,
                                    rvlm_stream_t stream) {
    RVLM_REQUIRE(x && grad && delta && velocity, "rvlm_pgd_linf_update: null pointer");
    auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    bool vec = (n % 4 == 0) && al(x) && al(grad) && al(delta) && al(velocity) &&
               (!x_adv_out || al(x_adv_out));
    if (vec)
        hipLaunchKernelGGL(pgd_linf_update_kernel<true>, dim3(ew_grid(n, 4)), dim3(256), 0,
                           (hipStream_t)stream, x, grad, delta, velocity, n, eps, stepsize,
                           momentum, mode_max, x_adv_out, flags);
    else
        hipLaunchKernelGGL(pgd_linf_update_kernel<false>, dim3(ew_grid(n, 1)), dim3(256), 0,
                           (hipStream_t)stream, x, grad, delta, velocity, n, eps, stepsize,
                           momentum, mode_max, x_adv_out, flags);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_pgd_l2_update(const float* x, const float* grad, float* delta, float* velocity, size_t n_per_sample,
                                  int B, float eps, float stepsize, float momentum, int mode_max, float* x_adv_out,
                                  int32_t* flags, rvlm_stream_t stream) {
    RVLM_REQUIRE(x && grad && delta && velocity && n_per_sample > 0 && B > 0, "rvlm_pgd_l2_update: bad arguments");
    hipLaunchKernelGGL(pgd_l2_update_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, grad, delta, velocity,
                       n_per_sample, eps, stepsize, momentum, mode_max, x_adv_out, flags);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_apgd_linf_step(const float* x, float* x_adv, float* x_adv_old,
                                   const float* grad, const float* step, float a, float eps,
                                   size_t n_per_sample, int B, rvlm_stream_t stream) {
    RVLM_REQUIRE(x && x_adv && x_adv_old && grad && step && B > 0, "rvlm_apgd_linf_step: bad args");
    int gx = (int)((n_per_sample + 1023) / 1024);
    if (gx > 64) gx = 64;
    float oma = (float)(1.0 - (double)a);
    hipLaunchKernelGGL(apgd_linf_step_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x,
                       x_adv, x_adv_old, grad, step, a, oma, eps, n_per_sample);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_project_perturbation(const float* pert, size_t n_per_sample, int B, int norm_kind, float eps, float* out,
                                         rvlm_stream_t stream) {
    RVLM_REQUIRE(pert && out && B > 0 && n_per_sample > 0, "rvlm_project_perturbation: bad arguments");
    const size_t n = n_per_sample * (size_t)B;
    if (norm_kind == 0)
        hipLaunchKernelGGL(ew_clamp_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, (hipStream_t)stream, pert, -eps, eps, n, out);
    else if (norm_kind == 2)
        hipLaunchKernelGGL(row_l2_scale_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, pert, n_per_sample, 1, eps, out);
    else
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_project_perturbation: norm must be L-inf (0) or L2 (2)");
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
extern "C" int rvlm_normalize_grad(const float* grad, size_t n_per_sample, int B, int norm_kind, float* out,
                                   rvlm_stream_t stream) {
    RVLM_REQUIRE(grad && out && B > 0 && n_per_sample > 0, "rvlm_normalize_grad: bad arguments");
    const size_t n = n_per_sample * (size_t)B;
    if (norm_kind == 0)
        hipLaunchKernelGGL(ew_sign_kernel, dim3(ew_grid(n, 1)), dim3(256), 0, (hipStream_t)stream, grad, n, out);
    else if (norm_kind == 2)
        hipLaunchKernelGGL(row_l2_scale_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, grad, n_per_sample, 0, 0.0f, out);
    else
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_normalize_grad: norm must be L-inf (0) or L2 (2)");
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_apgd_l2_step(const float* x, float* x_adv, float* x_adv_old, const float* grad, const float* step,
                                 float a, float eps, size_t n_per_sample, int B, rvlm_stream_t stream) {
    RVLM_REQUIRE(x && x_adv && x_adv_old && grad && step && B > 0 && n_per_sample > 0, "rvlm_apgd_l2_step: bad args");
    const float oma = (float)(1.0 - (double)a);
    hipLaunchKernelGGL(apgd_l2_step_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, x_adv, x_adv_old, grad, step,
                       a, oma, eps, n_per_sample);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_apgd_controller_rho(int i, int B, int n_iter, int k, int do_check, double rho,
                                        const float* loss_i, const uint8_t* pred, float* loss_steps,
                                        float* loss_best, float* loss_best_last_check,
                                        float* reduced_last_check, float* step, uint8_t* acc,
                                        uint8_t* f_notpred, uint8_t* f_improved, uint8_t* f_reduced,
                                        rvlm_stream_t stream) {
    RVLM_REQUIRE(loss_i && pred && loss_steps && loss_best && loss_best_last_check &&
                     reduced_last_check && step && acc && f_notpred && f_improved && f_reduced,
                 "rvlm_apgd_controller: null pointer");
    RVLM_REQUIRE(i >= 0 && i < n_iter && k >= 1 && B > 0, "rvlm_apgd_controller: bad sizes");
    RVLM_REQUIRE(rho == rho, "rvlm_apgd_controller: rho is NaN");
    hipLaunchKernelGGL(apgd_controller_kernel, dim3(cdiv(B, 256)), dim3(256), 0,
                       (hipStream_t)stream, i, B, n_iter, k, do_check, rho, loss_i, pred, loss_steps,
                       loss_best, loss_best_last_check, reduced_last_check, step, acc, f_notpred,
                       f_improved, f_reduced);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_apgd_controller(int i, int B, int n_iter, int k, int do_check,
                                    const float* loss_i, const uint8_t* pred, float* loss_steps,
                                    float* loss_best, float* loss_best_last_check,
                                    float* reduced_last_check, float* step, uint8_t* acc,
                                    uint8_t* f_notpred, uint8_t* f_improved, uint8_t* f_reduced,
                                    rvlm_stream_t stream) {
    return rvlm_apgd_controller_rho(i, B, n_iter, k, do_check, 0.75, loss_i, pred, loss_steps, loss_best,
                                    loss_best_last_check, reduced_last_check, step, acc, f_notpred, f_improved,
                                    f_reduced, stream);
}

extern "C" int rvlm_apgd_select(float* x_adv, float* grad, float* x_best, float* grad_best,
                                float* x_best_adv, const uint8_t* f_notpred,
                                const uint8_t* f_improved, const uint8_t* f_reduced,
                                size_t n_per_sample, int B, rvlm_stream_t stream) {
    RVLM_REQUIRE(x_adv && grad && x_best && grad_best && x_best_adv && f_notpred && f_improved &&
                     f_reduced && B > 0, "rvlm_apgd_select: bad args");
    int gx = (int)((n_per_sample + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(apgd_select_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x_adv,
                       grad, x_best, grad_best, x_best_adv, f_notpred, f_improved, f_reduced,
                       n_per_sample);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_linf_random_start(const float* x, const float* t, float eps,
                                      size_t n_per_sample, int B, float* x_adv,
                                      rvlm_stream_t stream) {
    RVLM_REQUIRE(x && t && x_adv && B > 0, "rvlm_linf_random_start: bad args");
    hipLaunchKernelGGL(linf_random_start_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, t,
                       eps, n_per_sample, x_adv);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_l2_random_start(const float* x, const float* t, float eps, size_t n_per_sample, int B, float* x_adv,
                                    rvlm_stream_t stream) {
    RVLM_REQUIRE(x && t && x_adv && B > 0 && n_per_sample > 0, "rvlm_l2_random_start: bad args");
    hipLaunchKernelGGL(l2_random_start_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, t, eps, n_per_sample, x_adv);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_square_linf_propose(const float* x, const float* x_best, const int64_t* idx, int n_active, int C,
                                        int H, int W, int vh, int vw, int s, float eps, const float* sign,
                                        float* x_new, rvlm_stream_t stream) {
    RVLM_REQUIRE(x && x_best && idx && sign && x_new && n_active > 0 && C > 0 && H > 0 && W > 0,
                 "rvlm_square_linf_propose: bad args");
    RVLM_REQUIRE(s >= 1 && vh >= 0 && vw >= 0 && vh + s <= H && vw + s <= W, "rvlm_square_linf_propose: window outside the image");
    const long n_img = (long)C * H * W;
    const int gx = (int)std::min<long>((n_img + 255) / 256, 64);
    hipLaunchKernelGGL(square_linf_propose_kernel, dim3(gx, n_active), dim3(256), 0, (hipStream_t)stream, x, x_best,
                       (const long long*)idx, C, H, W, vh, vw, s, eps, 2.0f * eps, sign, x_new);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
extern "C" int rvlm_square_accept(float* x_best, const float* x_new, const int64_t* idx, const float* take, int n_active,
                                  size_t n_per_image, rvlm_stream_t stream) {
    RVLM_REQUIRE(x_best && x_new && idx && take && n_active > 0 && n_per_image > 0, "rvlm_square_accept: bad args");
    const int gx = (int)std::min<size_t>((n_per_image + 255) / 256, 64);
    hipLaunchKernelGGL(square_accept_kernel, dim3(gx, n_active), dim3(256), 0, (hipStream_t)stream, x_best, x_new,
                       (const long long*)idx, take, (long)n_per_image);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
