// FARE (l2) / TeCoA (ce) losses with their gradient w.r.t. the embedding, fused:
// replaces compute_loss / l2 / ce (train/adversarial_training_clip.py:495-528) + autograd.
#include "kernels.h"

namespace rvlm {

// per-sample  sum_d (e - e0)^2  (…clip.py:515-519) and d = 2 (e - e0) * gscale, one wave/sample
__global__ void __launch_bounds__(256)
l2_loss_kernel(const float* __restrict__ emb, const float* __restrict__ ref, int B, int D,
               float gscale, float* __restrict__ per_sample, float* __restrict__ d_emb) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float acc = 0.0f;
    for (int c = lane; c < D; c += 64) {
        const float d = emb[(long)row * D + c] - ref[(long)row * D + c];
        acc = fmaf(d, d, acc);
        if (d_emb) d_emb[(long)row * D + c] = 2.0f * d * gscale;
    }
    acc = wave_sum(acc);
    if (lane == 0 && per_sample) per_sample[row] = acc;
}

// logits row -> loss = logsumexp - logit[y] ; dlogits = (softmax - onehot) * gscale (in place);
// pred_eq = (argmax == y), ties -> first index
__global__ void __launch_bounds__(256)
ce_loss_kernel(float* __restrict__ logits, const int64_t* __restrict__ targets, int B, int C,
               float gscale, float* __restrict__ per_sample, uint8_t* __restrict__ pred_eq,
               int write_grad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float* l = logits + (long)row * C;
    const int y = (int)targets[row];
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = l[c];
        if (v > m) { m = v; mi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    float sum = 0.0f;
    for (int c = lane; c < C; c += 64) sum += expf(l[c] - m);
    sum = wave_sum(sum);
    const float lse = m + logf(sum);
    if (lane == 0) {
        if (per_sample) per_sample[row] = lse - l[y];
        if (pred_eq) pred_eq[row] = (mi == y) ? 1 : 0;
    }
    if (write_grad) {
        for (int c = lane; c < C; c += 64) {
            const float p = expf(l[c] - lse);
            l[c] = (p - (c == y ? 1.0f : 0.0f)) * gscale;
        }
    }
}

// DLR losses of AutoAttack on a logits row (autopgd_base.py:195-201 untargeted, :613-618 targeted), one wave per row:
// top-4 selection (4 rounds of wave arg-max over per-lane candidates), the loss with the reference's operation order,
// and d loss / d logits written in place (autograd of the same expression: the sort routes gradient to the elements
// holding ranks 1, 3 (and 4 / 2)).  pred_eq = (arg-max == y).
__global__ void __launch_bounds__(256)
dlr_loss_kernel(float* __restrict__ logits, const int64_t* __restrict__ targets, const int64_t* __restrict__ y_target,
                int B, int C, float gscale, float* __restrict__ per_sample, uint8_t* __restrict__ pred_eq,
                int write_grad) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float* l = logits + (long)row * C;
    const int y = (int)targets[row];
    const int yt = y_target ? (int)y_target[row] : -1;
    // per-lane top-4 of the elements lane, lane+64, ... (descending; ties keep the smaller index first)
    float v[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int ix[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int c = lane; c < C; c += 64) {
        float x = l[c];
        int xi = c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x > v[k]) {
                const float tv = v[k]; const int ti = ix[k];
                v[k] = x; ix[k] = xi; x = tv; xi = ti;
            }
        }
    }
    float z[4];
    int zi[4];
    int head = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float m = -INFINITY;
        int mi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (head == k) { m = v[k]; mi = ix[k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o, 64);
            const int oi = __shfl_xor(mi, o, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        z[r] = m; zi[r] = mi;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (head == k && ix[k] == mi) { head = k + 1; break; }
    }
    const float zy = l[y];
    float num, den, g_y, g_1, g_2 = 0.0f, g_3, g_4 = 0.0f, g_t = 0.0f;
    if (yt < 0) {
        const float ind = (zi[0] == y) ? 1.0f : 0.0f;
        num = -((zy - z[1] * ind) - z[0] * (1.0f - ind));
        den = (z[0] - z[2]) + 1e-12f;
        const float gn = 1.0f / den, gd = -num / (den * den);
        g_y = -gn; g_2 = ind * gn; g_1 = (1.0f - ind) * gn + gd; g_3 = -gd;
    } else {
        num = -(zy - l[yt]);
        den = (z[0] - 0.5f * (z[2] + z[3])) + 1e-12f;
        const float gn = 1.0f / den, gd = -num / (den * den);
        g_y = -gn; g_t = gn; g_1 = gd; g_3 = -0.5f * gd; g_4 = -0.5f * gd;
    }
    if (lane == 0) {
        if (per_sample) per_sample[row] = num / den;
        if (pred_eq) pred_eq[row] = (zi[0] == y) ? 1 : 0;
    }
    if (write_grad) {
        for (int c = lane; c < C; c += 64) {
            float g = 0.0f;
            if (c == y) g += g_y;
            if (c == yt) g += g_t;
            if (c == zi[0]) g += g_1;
            if (c == zi[1]) g += g_2;
            if (c == zi[2]) g += g_3;
            if (c == zi[3]) g += g_4;
            l[c] = g * gscale;
        }
    }
}

// deterministic single-block reduction of the per-sample losses -> scalar
__global__ void __launch_bounds__(256)
reduce_loss_kernel(const float* __restrict__ per_sample, int B, float scale, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < B; i += 256) s += per_sample[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

__global__ void __launch_bounds__(256)
argmax_eq_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets, int B, int C,
                 uint8_t* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* l = logits + (long)row * C;
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = l[c];
        if (v > m) { m = v; mi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) out[row] = ((int64_t)mi == targets[row]) ? 1 : 0;
}

// F.cosine_similarity(a, b, dim=1, eps=1e-8) per row: a.b / sqrt(max(|a|^2 |b|^2, eps^2)); one wave per row
__global__ void __launch_bounds__(256)
cosine_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, int B, int D, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float ab = 0.0f, aa = 0.0f, bb = 0.0f;
    for (int c = lane; c < D; c += 64) {
        const float x = a[(long)row * D + c], y = b[(long)row * D + c];
        ab = fmaf(x, y, ab); aa = fmaf(x, x, aa); bb = fmaf(y, y, bb);
    }
    ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
    if (lane == 0) out[row] = ab / sqrtf(fmaxf(aa * bb, 1e-16f));
}

}  // namespace rvlm

using namespace rvlm;

extern "C" int rvlm_loss_grad(int loss_kind, int reduction, const float* emb, const float* ref,
                              const int64_t* targets, const int64_t* y_target, int B, int D, int C, float logit_scale,
                              float* loss_per_sample, float* loss_scalar, float* d_emb,
                              uint8_t* pred_eq, float* scratch, rvlm_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    RVLM_REQUIRE(emb && ref && B > 0 && D > 0, "rvlm_loss_grad: bad arguments");
    // the reference's losses assert batch > 1 (…clip.py:513,526)
    RVLM_REQUIRE(B > 1, "rvlm_loss_grad: batch size must be > 1 (reference asserts out.shape[0] > 1)");
    RVLM_REQUIRE(reduction == RVLM_RED_MEAN || reduction == RVLM_RED_NONE,
                 "rvlm_loss_grad: unknown reduction");
    const float gscale = (reduction == RVLM_RED_MEAN) ? 1.0f / (float)B : 1.0f;
    if (loss_kind == RVLM_LOSS_L2) {
        RVLM_REQUIRE(loss_per_sample || !loss_scalar, "rvlm_loss_grad: loss_scalar needs loss_per_sample");
        hipLaunchKernelGGL(l2_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, emb, ref, B, D, gscale,
                           loss_per_sample, d_emb);
        RVLM_CHECK_LAUNCH();
    } else if (loss_kind == RVLM_LOSS_CE || loss_kind == RVLM_LOSS_DLR || loss_kind == RVLM_LOSS_DLR_TARGETED) {
        RVLM_REQUIRE(targets && scratch && C > 0, "rvlm_loss_grad: head losses need targets, scratch, C");
        RVLM_REQUIRE(loss_kind != RVLM_LOSS_DLR_TARGETED || y_target, "rvlm_loss_grad: dlr-targeted needs y_target");
        // checks.check_n_classes: the DLR loss needs 3 logits, its targeted form 4
        RVLM_REQUIRE(loss_kind == RVLM_LOSS_CE || C >= (loss_kind == RVLM_LOSS_DLR ? 3 : 4),
                     "rvlm_loss_grad: too few classes for the DLR loss");
        RVLM_REQUIRE(loss_per_sample || !loss_scalar, "rvlm_loss_grad: loss_scalar needs loss_per_sample");
        float* logits = scratch;                 // [B, C]
        float* Ts = scratch + (size_t)B * C;     // [D, C] = logit_scale * T   (…clip.py:501)
        int rc = scale_copy_f32(ref, Ts, (size_t)D * C, logit_scale, s);
        if (rc) return rc;
        GemmF32 g;
        g.A = emb; g.sam = D; g.sak = 1;
        g.B = Ts; g.sbn = 1; g.sbk = C;
        g.C = logits; g.scm = C; g.scn = 1;
        g.M = B; g.N = C; g.K = D;
        rc = gemm_f32(g, s);
        if (rc) return rc;
        if (loss_kind == RVLM_LOSS_CE)
            hipLaunchKernelGGL(ce_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, logits, targets, B, C,
                               gscale, loss_per_sample, pred_eq, d_emb ? 1 : 0);
        else
            hipLaunchKernelGGL(dlr_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, logits, targets,
                               loss_kind == RVLM_LOSS_DLR_TARGETED ? y_target : (const int64_t*)nullptr, B, C, gscale,
                               loss_per_sample, pred_eq, d_emb ? 1 : 0);
        RVLM_CHECK_LAUNCH();
        if (d_emb) {
            GemmF32 h;
            h.A = logits; h.sam = C; h.sak = 1;     // dlogits [B, C]
            h.B = Ts; h.sbn = C; h.sbk = 1;         // (n = d, k = c)
            h.C = d_emb; h.scm = D; h.scn = 1;
            h.M = B; h.N = D; h.K = C;
            rc = gemm_f32(h, s);
            if (rc) return rc;
        }
    } else {
        return fail(RVLM_ERR_ARG, "rvlm_loss_grad: loss not supported");  // ValueError in …clip.py:506
    }
    if (loss_scalar) {
        hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, s, loss_per_sample, B,
                           reduction == RVLM_RED_MEAN ? 1.0f / (float)B : 1.0f, loss_scalar);
        RVLM_CHECK_LAUNCH();
    }
    return RVLM_OK;
}

extern "C" int rvlm_argmax_eq(const float* logits, const int64_t* targets, int B, int C,
                              uint8_t* out, rvlm_stream_t stream) {
    RVLM_REQUIRE(logits && targets && out && B > 0 && C > 0, "rvlm_argmax_eq: bad arguments");
    hipLaunchKernelGGL(argmax_eq_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, logits,
                       targets, B, C, out);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// ---- plain cross entropy on given logits (ce(), train/adversarial_training_clip.py:523-528) ----------------------
extern "C" int rvlm_ce_logits(const float* logits, const int64_t* targets, int B, int C, int reduction,
                              float* loss_per_sample, float* loss_scalar, float* d_logits, uint8_t* pred_eq,
                              rvlm_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    RVLM_REQUIRE(logits && targets && d_logits && loss_per_sample && B > 0 && C > 0, "rvlm_ce_logits: bad arguments");
    // no B > 1 here: the batch-size assert belongs to the trainer's ce() (train/adversarial_training_clip.py:526, kept in
    // clip_model.ce); AutoPGD's nn.CrossEntropyLoss(reduction='none') (autoattack/autopgd_base.py:249) takes one sample
    RVLM_REQUIRE(reduction == RVLM_RED_MEAN || reduction == RVLM_RED_NONE, "rvlm_ce_logits: unknown reduction");
    const float gscale = (reduction == RVLM_RED_MEAN) ? 1.0f / (float)B : 1.0f;
    if (d_logits != logits) RVLM_HIP(hipMemcpyAsync(d_logits, logits, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(ce_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, d_logits, targets, B, C, gscale,
                       loss_per_sample, pred_eq, 1);
    RVLM_CHECK_LAUNCH();
    if (loss_scalar) {
        hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, s, loss_per_sample, B,
                           reduction == RVLM_RED_MEAN ? 1.0f / (float)B : 1.0f, loss_scalar);
        RVLM_CHECK_LAUNCH();
    }
    return RVLM_OK;
}

// ---- zero-shot head of ClassificationModel.forward (CLIP_eval/clip_robustbench.py:66-68): logits = (emb @ T) * scale,
// in that order; backward d_emb = (d_logits * scale) @ T^T --------------------------------------------------------------
extern "C" int rvlm_head_logits(const float* emb, const float* T, int B, int D, int C, float scale, float* logits,
                                rvlm_stream_t stream) {
    RVLM_REQUIRE(emb && T && logits && B > 0 && D > 0 && C > 0, "rvlm_head_logits: bad arguments");
    GemmF32 g;
    g.A = emb; g.sam = D; g.sak = 1;
    g.B = T; g.sbn = 1; g.sbk = C;
    g.C = logits; g.scm = C; g.scn = 1;
    g.M = B; g.N = C; g.K = D; g.alpha = scale;     // alpha multiplies the finished fp32 dot product: (emb @ T) * scale
    return gemm_f32(g, (hipStream_t)stream);
}
extern "C" int rvlm_head_logits_bwd(const float* d_logits, const float* T, int B, int D, int C, float scale,
                                    float* d_emb, rvlm_stream_t stream) {
    RVLM_REQUIRE(d_logits && T && d_emb && B > 0 && D > 0 && C > 0, "rvlm_head_logits_bwd: bad arguments");
    GemmF32 h;
    h.A = d_logits; h.sam = C; h.sak = 1;
    h.B = T; h.sbn = C; h.sbk = 1;          // (n = d, k = c)
    h.C = d_emb; h.scm = D; h.scn = 1;
    h.M = B; h.N = D; h.K = C; h.alpha = scale;
    return gemm_f32(h, (hipStream_t)stream);
}

// ---- logging metrics of the training step (train/adversarial_training_clip.py:368-387) -------------------------------
extern "C" int rvlm_cosine_rows(const float* a, const float* b, int B, int D, float* out_per_row, float* out_mean,
                                rvlm_stream_t stream) {
    RVLM_REQUIRE(a && b && out_per_row && B > 0 && D > 0, "rvlm_cosine_rows: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cosine_rows_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, a, b, B, D, out_per_row);
    RVLM_CHECK_LAUNCH();
    if (out_mean) {
        hipLaunchKernelGGL(reduce_loss_kernel, dim3(1), dim3(256), 0, s, out_per_row, B, 1.0f / (float)B, out_mean);
        RVLM_CHECK_LAUNCH();
    }
    return RVLM_OK;
}
extern "C" int rvlm_l2_normalize_rows(const float* e, int B, int D, float* out, float* inv_norm, rvlm_stream_t stream) {
    RVLM_REQUIRE(e && out && inv_norm && B > 0 && D > 0, "rvlm_l2_normalize_rows: bad arguments");
    return l2_normalize_fwd(e, out, inv_norm, B, D, (hipStream_t)stream);
}
