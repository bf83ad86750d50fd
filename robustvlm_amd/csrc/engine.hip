// The fixed-function "ViT forward + input-gradient" engine and the device-resident PGD / APGD loops.
//
// Not an autograd graph: one handle owns the converted weights and a workspace sized for max_batch
// (MI355X has 288 GB of HBM3E, so every activation the dgrad chain needs is simply kept resident:
// ~19 GB at ViT-L/14, B=128, bf16), and forward / backward are fixed launch sequences on the caller's
// stream with no host synchronisation.  Only the INPUT gradient is produced (no wgrad, no saved
// linear inputs), exactly what torch.autograd.grad(loss, perturbation) asks for in
// train/pgd_train.py:38 and train/apgd_train.py:185,295.
#include <vector>
#include <cstring>
#include <map>

#include "kernels.h"

namespace rvlm {
void attn_set_use_tr(int on);

struct Layer {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *b_in, *b_out, *b_fc, *b_proj;
    float *w_in, *w_out, *w_fc, *w_proj;                          // fp32 mode (reference layout)
    bf16_t *w_in_nk, *w_in_t, *w_out_nk, *w_out_t, *w_fc_nk, *w_fc_t, *w_proj_nk, *w_proj_t;  // bf16
};

struct ProfRec { int cls; hipEvent_t a, b; };
struct ProfAcc { double ms = 0, flops = 0, bytes = 0; int64_t n = 0; };

}  // namespace rvlm

using namespace rvlm;

struct rvlm_vit {
    rvlm_vit_config cfg;
    int S, G, W, L, H, D, P, img, Kp, Kpad, maxB, Mp, Mp0;
    bool bf16;
    bool x3 = false;     // fp32 storage, split-bf16 linears (RVLM_PREC_F32X3, csrc/x3_kernels.hip): w_*_nk / w_*_t hold [N, 3K] / [K, 3N]
    bf16_t* a3 = nullptr;        // x3: [Mp, 3 * 4W] bf16, the split copy [hi | hi | lo] of the linear in front of the GEMM
    const void* a3_of = nullptr; // x3: the fp32 buffer whose split a3 ALREADY holds (written by the fused activation + split pass)
    size_t esz;  // activation element size
    std::vector<Layer> layers;
    float *cls, *pos, *proj, *lnpre_w, *lnpre_b, *lnpost_w, *lnpost_b, *conv_f32;
    bf16_t *conv_nk, *conv_t;
    // activations
    void* A0;            // [Mp0, Kpad] T
    float* patch_out;    // [Mp0, W] f32
    std::vector<float*> xs;      // 2L+1 x [Mp, W] f32 residual stream
    float *st_mean, *st_rstd;    // [(2L+2) * Mp]
    void* ln_out;        // [Mp, W] T
    std::vector<void*> qkv;      // L x [Mp, 3W] T
    std::vector<void*> attn_o;   // L x [Mp, W] T
    std::vector<float*> lse;     // L x [B*H*Sp]
    std::vector<float*> lse2;    // fp32 storage: L x [B*H*Sp], the flash kernels' log-sum-exp rows, written by the softmax pass of a
                                 // saving forward for the handoff to a bf16 handle's backward (vit_backward_from)
    float* cur_lse2 = nullptr;   // ... of the block the forward is in
    // Forward FOR a bf16 handle (the handoff, vit_backward_from): `peer` is set for the duration of a saving forward whose input
    // gradient that handle will evaluate.  The attention then runs as the fp32 flash kernel (no probabilities kept: this handle's
    // own backward is refused for that pass, saved_mode 3), which writes the peer's bf16 qkv / attention output / log-sum-exp rows as
    // it goes; the split-bf16 activation pass writes the peer's act'(fc1).  What was exported is remembered per saved forward.
    rvlm_vit* peer = nullptr;
    unsigned long long uid = 0;          // unique per created handle (never reused, unlike an address)
    unsigned long long exported_to = 0;  // uid of the handle the saved forward's bf16 qkv / attention output / lse rows already sit in
    bool exported_dact = false;          // ... and its act'(fc1)
    bf16_t* cur_dact_out = nullptr;      // the peer's act'(fc1) buffer of the block the forward is in
    bf16_t *cur_qkv_bf = nullptr, *cur_o_bf = nullptr;
    bool cur_flash = false;              // this forward's attention runs as the flash kernel
    // Non-saving forwards on the flash kernel too (rvlm_vit_set_flash_inference).  OFF by default: a handle's saving and non-saving
    // forwards must agree BIT FOR BIT - FARE's loss at delta = 0 is |phi(x) - phi(x)|^2 = 0 with a zero gradient exactly (apgd_train
    // starts there, train/apgd_train.py:157-160), and 1e-7 of rounding difference between two attention paths turns that into a
    // random sign pattern.  ON for the fp32-storage handle of a handoff engine, whose saving forwards (run for the peer) are flash.
    bool flash_inference = false;
    // fp32-storage handle at a sequence length the fp32 flash BACKWARD takes (S = 257): every attention of this handle - saving or not,
    // forward and backward - runs on the flash kernels (attention_f32.hip); no probabilities are kept or allocated (13 GB at ViT-L/14,
    // B = 128) and saving / non-saving forwards agree bit for bit by construction.
    bool own_flash = false;
    // bf16 handles, attack path (backward_impl): the residual-GRADIENT stream is the bf16 buffer only - the LayerNorm backward reads it
    // to accumulate and writes it back, 10 instead of 16 B per element and pass (no fp32 copy).  What it costs: an accumulator of 49
    // terms rounded to 8 bits of mantissa 49 times - measured by the emulation (oracle/split_bf16_emulation.py, arm
    // f32fwd-bf16bwd-bf16res: first-step sign agreement of a bf16 backward 0.9982 -> 0.9961, CLIP-like 0.9977 -> 0.9945) and by the
    // full-size tests; the handoff's backward (the faithful first iteration) and the training step keep the fp32 stream.
    // RVLM_DRES_FP32=1 (read at creation) keeps it everywhere.
    bool lp_dres = false;
    std::vector<void*> h_pre;    // L x [Mp, 4W] T
    void* g_act;         // [Mp, 4W] T
    float *pooled, *emb_raw, *inv_norm;   // [maxB, W], [maxB, D], [maxB]
    // backward scratch
    float* dres;  void* dres_lp;  void* d_o;  void* dqkv;  void* dh;  void* d_ln;  void* d_patch;
    float* dA0;   float* dsum;   float *d_raw, *d_pooled;
    float* dscores;            // fp32 mode: dP / dS of one block [B, H, S, round_up(S, 4)] (the probabilities live in lse[l])
    float* splitk_scratch;     // fp32 slabs of the split-K few-row GEMMs (+ weight-gradient GEMMs when trainable)
    size_t splitk_bytes = 0;
    // class-token tail (bf16 mode): in the last block only the class token's row is live downstream of the attention
    // (the output is ln_post(x[:, 0]) @ proj), so out-proj / MLP / their LayerNorm run on B rows and the attention on
    // the class query only.  Row b of the B-row intermediates is image b; x rows stay in place (stride S*W).
    bool cls_tail = false;
    float* lse_cls = nullptr;
    float* red_scratch = nullptr;   // partial column sums of the training step
    size_t red_floats = 0;
    // training (cfg.trainable): inputs of every linear layer + embedding tokens + transpose scratch
    bool trainable = false;
    bool inference_only = false;
    bool provider = false;       // cfg.trainable == -2: an fp32-storage handle whose saving forwards are all run FOR a bf16 handle
                                 // (rvlm_vit_forward_for / rvlm_pgd_run_mixed_fwd) on the flash attention: the fp32 residual stream, the
                                 // LayerNorm statistics and the log-sum-exp rows are kept per block; qkv, attention output, fc1
                                 // pre-activation and probabilities have ONE slot (their bf16 forms are written into the consumer as
                                 // the forward goes), and there is no backward scratch: 51 -> 13 GiB at ViT-L/14, B = 128
    std::vector<void*> ln1_out, ln2_out, g_act_l;   // L x [Mp,W], [Mp,W], [Mp,4W] T
    float *tokens, *dtok;      // [Mp, W] f32
    void *tA, *tB;             // bf16 mode: transposed operands of the wgrad GEMMs the copy-free form does not take
                               // (conv1, widths that are no multiple of 256, fewer than 256 tokens); t_elems elements each
    long Mpt = 0;
    size_t t_elems = 0;        // bf16 elements of tA (= of tB)
    // attack state
    float* img_buf[5];         // [maxB*3*img*img]
    float *emb, *d_emb, *loss_ps, *loss_scalar, *loss_scratch;
    float *ap_loss_steps, *ap_loss_best, *ap_loss_best_lc, *ap_reduced_lc, *ap_step;
    uint8_t *ap_acc, *ap_pred, *ap_f0, *ap_f1, *ap_f2;
    double ap_rho = 0.75;      // APGDAttack's oscillation threshold (rvlm_vit_set_apgd_rho)
    size_t loss_scratch_floats;
    // bookkeeping
    std::vector<void*> allocs;
    size_t bytes = 0;
    int saved_B = 0;
    int saved_mode = 0;
    int next_param_stage = 0;   // rvlm_vit_backward_params_stages: the stage the saved forward's backward continues at
    bool saved_norm = false;
    // profiling
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<std::string> cls_names;
    std::vector<ProfAcc> accs;
    std::vector<double> cls_flops, cls_bytes;

    float* mean_at(int i) { return st_mean + (size_t)i * Mp; }
    float* rstd_at(int i) { return st_rstd + (size_t)i * Mp; }
};

namespace rvlm {

static int dev_alloc(rvlm_vit* h, void** p, size_t bytes, bool zero = true) {
    bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    if (zero) {
        e = hipMemset(*p, 0, bytes);
        if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    h->allocs.push_back(*p);
    h->bytes += bytes;
    return RVLM_OK;
}
#define ALLOC(ptr, bytes)                                              \
    do { int _rc = dev_alloc(h, (void**)&(ptr), (bytes)); if (_rc) return _rc; } while (0)

// ---- profiling ------------------------------------------------------------------------------
static int prof_class(rvlm_vit* h, const char* name) {
    for (size_t i = 0; i < h->cls_names.size(); ++i) if (h->cls_names[i] == name) return (int)i;
    h->cls_names.push_back(name);
    h->accs.emplace_back();
    return (int)h->cls_names.size() - 1;
}
struct ProfScope {
    rvlm_vit* h; hipStream_t s; int idx = -1;
    ProfScope(rvlm_vit* h_, hipStream_t s_, const char* name, double flops, double bytes) : h(h_), s(s_) {
        if (!h->prof) return;
        ProfRec r;
        r.cls = prof_class(h, name);
        (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, s);
        h->recs.push_back(r);
        idx = (int)h->recs.size() - 1;
        h->accs[r.cls].flops += flops;
        h->accs[r.cls].bytes += bytes;
        h->accs[r.cls].n += 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(h->recs[idx].b, s); }
};
#define PROF(name, flops, bytes) ProfScope _ps(h, s, name, (double)(flops), (double)(bytes))

// ---- weights --------------------------------------------------------------------------------
static int copy_f32(rvlm_vit* h, float** dst, const float* src, size_t n, hipStream_t s, bool alloc) {
    if (!src) return fail(RVLM_ERR_ARG, "rvlm_vit: null weight pointer");
    if (alloc) { int rc = dev_alloc(h, (void**)dst, n * 4, false); if (rc) return rc; }
    RVLM_HIP(hipMemcpyAsync(*dst, src, n * 4, hipMemcpyDeviceToDevice, s));
    return RVLM_OK;
}
// nk: [rows, cols] bf16 (ld = cols_pad) ; t: [cols, rows] bf16
static int conv_w(rvlm_vit* h, bf16_t** nk, bf16_t** t, const float* src, int rows, int cols, int cols_pad,
                  hipStream_t s, bool alloc) {
    if (!src) return fail(RVLM_ERR_ARG, "rvlm_vit: null weight pointer");
    if (alloc) {
        int rc = dev_alloc(h, (void**)nk, (size_t)rows * cols_pad * 2); if (rc) return rc;
        rc = dev_alloc(h, (void**)t, (size_t)cols_pad * rows * 2); if (rc) return rc;
    }
    if (convert_f32_to_bf16_pair(src, cols, *nk, cols_pad, *t, rows, rows, cols, s)) { RVLM_CHECK_LAUNCH(); return RVLM_OK; }
    int rc = convert_f32_to_bf16(src, cols, *nk, cols_pad, rows, cols, 0, s); if (rc) return rc;
    return convert_f32_to_bf16(src, cols, *t, rows, rows, cols, 1, s);
}

static int load_weights(rvlm_vit* h, const rvlm_vit_weights* w, hipStream_t s, bool alloc) {
    const int W = h->W, L = h->L;
    int rc;
#define CP(dst, src, n) if ((rc = copy_f32(h, &(dst), (src), (n), s, alloc))) return rc
    CP(h->cls, w->class_embedding, W);
    CP(h->pos, w->positional_embedding, (size_t)h->S * W);
    CP(h->proj, w->proj, (size_t)W * h->D);
    CP(h->lnpre_w, w->ln_pre_weight, W); CP(h->lnpre_b, w->ln_pre_bias, W);
    CP(h->lnpost_w, w->ln_post_weight, W); CP(h->lnpost_b, w->ln_post_bias, W);
    if (h->bf16) {
        if ((rc = conv_w(h, &h->conv_nk, &h->conv_t, w->conv1_weight, W, h->Kp, h->Kpad, s, alloc))) return rc;
    } else {
        CP(h->conv_f32, w->conv1_weight, (size_t)W * h->Kp);
    }
    if (!w->blocks_host) return fail(RVLM_ERR_ARG, "rvlm_vit: blocks_host is null");
    for (int l = 0; l < L; ++l) {
        const rvlm_vit_block_weights& b = w->blocks_host[l];
        Layer& y = h->layers[l];
        CP(y.ln1_w, b.ln_1_weight, W); CP(y.ln1_b, b.ln_1_bias, W);
        CP(y.ln2_w, b.ln_2_weight, W); CP(y.ln2_b, b.ln_2_bias, W);
        CP(y.b_in, b.attn_in_proj_bias, 3 * W); CP(y.b_out, b.attn_out_proj_bias, W);
        CP(y.b_fc, b.mlp_c_fc_bias, 4 * W); CP(y.b_proj, b.mlp_c_proj_bias, W);
        if (h->bf16) {
            if ((rc = conv_w(h, &y.w_in_nk, &y.w_in_t, b.attn_in_proj_weight, 3 * W, W, W, s, alloc))) return rc;
            if ((rc = conv_w(h, &y.w_out_nk, &y.w_out_t, b.attn_out_proj_weight, W, W, W, s, alloc))) return rc;
            if ((rc = conv_w(h, &y.w_fc_nk, &y.w_fc_t, b.mlp_c_fc_weight, 4 * W, W, W, s, alloc))) return rc;
            if ((rc = conv_w(h, &y.w_proj_nk, &y.w_proj_t, b.mlp_c_proj_weight, W, 4 * W, 4 * W, s, alloc))) return rc;
        } else {
            CP(y.w_in, b.attn_in_proj_weight, (size_t)3 * W * W);
            CP(y.w_out, b.attn_out_proj_weight, (size_t)W * W);
            CP(y.w_fc, b.mlp_c_fc_weight, (size_t)4 * W * W);
            CP(y.w_proj, b.mlp_c_proj_weight, (size_t)4 * W * W);
            if (h->x3) {      // [W_hi | W_lo | W_hi] and the same of W^T, from the device copies just made
                struct { bf16_t **nk, **t; const float* src; int rows, cols; } ws[4] = {
                    {&y.w_in_nk, &y.w_in_t, b.attn_in_proj_weight, 3 * W, W}, {&y.w_out_nk, &y.w_out_t, b.attn_out_proj_weight, W, W},
                    {&y.w_fc_nk, &y.w_fc_t, b.mlp_c_fc_weight, 4 * W, W}, {&y.w_proj_nk, &y.w_proj_t, b.mlp_c_proj_weight, W, 4 * W}};
                for (auto& q : ws) {
                    if (alloc) {
                        if ((rc = dev_alloc(h, (void**)q.nk, (size_t)q.rows * q.cols * 6, false))) return rc;
                        if ((rc = dev_alloc(h, (void**)q.t, (size_t)q.rows * q.cols * 6, false))) return rc;
                    }
                    if ((rc = x3_prepare_weight(q.src, q.rows, q.cols, *q.nk, *q.t, s))) return rc;
                }
            }
        }
    }
#undef CP
    return RVLM_OK;
}

// every GEMM descriptor carries its handle's split-K slab scratch (no process-wide state: handles coexist)
static inline void with_scratch(const rvlm_vit* h, GemmBf16& g) { g.splitk = h->splitk_scratch; g.splitk_bytes = h->splitk_bytes; }

// ---- linear layers (dispatch on precision) ---------------------------------------------------
// forward: out[M,N] = epi(A[M,K] @ Wt[N,K]^T + bias)
template <typename T>
static int linear_fwd(rvlm_vit* h, hipStream_t s, const void* A, long lda, int M, int N, int K,
                      const float* w_f32, const bf16_t* w_nk, const float* bias, int epi, void* out, long ldo,
                      void* out_pre, const float* residual, int a_rows = 0);
// x3 mode: the encoder linears whose shapes the persistent bf16 GEMM takes (everything else - the patch embedding with its
// K = 3 P^2, tiny test models - stays on the fp32 tiles)
static inline bool x3_takes(const rvlm_vit* h, int M, int N, int K) {
    return h->x3 && h->a3 && M >= 256 && N % 256 == 0 && K % 128 == 0 && (long)round_up(M, 256) * 3 * K * 2 < (1L << 31);
}
template <>
int linear_fwd<bf16_t>(rvlm_vit* h, hipStream_t s, const void* A, long lda, int M, int N, int K,
                       const float*, const bf16_t* w_nk, const float* bias, int epi, void* out, long ldo,
                       void* out_pre, const float* residual, int a_rows) {
    GemmBf16 g;
    g.A = (const bf16_t*)A; g.lda = lda; g.Bw = w_nk; g.ldb = K; g.M = M; g.N = N; g.K = K;
    g.a_rows = a_rows > 0 ? a_rows : (int)round_up(M, 128); g.epi = epi; g.bias = bias; g.out = out; g.ldo = ldo;
    g.out_pre = (bf16_t*)out_pre; g.residual = residual; g.act = h->cfg.act;
    with_scratch(h, g);
    return gemm_bf16_nt(g, s);
}
template <>
int linear_fwd<float>(rvlm_vit* h, hipStream_t s, const void* A, long lda, int M, int N, int K,
                      const float* w_f32, const bf16_t* w_nk, const float* bias, int epi, void* out, long ldo,
                      void* out_pre, const float* residual, int) {
    if (x3_takes(h, M, N, K) && w_nk) {
        // split-bf16: A3 = [A_hi | A_hi | A_lo] against W3 = [W_hi | W_lo | W_hi] - one bf16 GEMM of contraction length 3 K with an
        // fp32 epilogue; the activation (fc1) runs as its own pass over the fp32 pre-activation
        int rc = RVLM_OK;
        if (h->a3_of != A && (rc = x3_split_rows((const float*)A, lda, h->a3, M, (int)round_up(M, 256), K, s))) return rc;
        h->a3_of = nullptr;
        const bool act = epi == EPI_BF16_ACT;
        GemmBf16 g;
        g.A = h->a3; g.lda = 3L * K; g.Bw = w_nk; g.ldb = 3L * K; g.M = M; g.N = N; g.K = 3 * K;
        g.a_rows = (int)round_up(M, 256); g.bias = bias; g.residual = residual;
        g.epi = residual ? EPI_F32_RESID : EPI_F32;
        g.out = (act && out_pre) ? out_pre : out; g.ldo = ldo; g.act = h->cfg.act;
        with_scratch(h, g);
        if ((rc = gemm_bf16_nt(g, s))) return rc;
        if (act) {
            // act(h) goes straight into the split copy the next linear (fc2: A = `out`, K = N) reads; `out` itself is not written
            // (its only other reader is the weight gradient, which this precision does not have)
            const bool fuse = x3_takes(h, M, 256, N);
            if ((rc = x3_act((const float*)g.out, ldo, (float*)out, ldo, fuse ? h->a3 : nullptr, M, (int)round_up(M, 256), N, h->cfg.act, 0, s,
                             h->cur_dact_out, ldo)))
                return rc;
            if (h->cur_dact_out) h->exported_dact = true;
            if (fuse) h->a3_of = out;
        }
        return RVLM_OK;
    }
    GemmF32 g;
    g.A = (const float*)A; g.sam = lda; g.sak = 1;
    g.B = w_f32; g.sbn = K; g.sbk = 1;
    g.C = (float*)out; g.scm = ldo; g.scn = 1;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.residual = residual;
    if (epi == EPI_BF16_ACT) { g.act = h->cfg.act; g.C_pre = (float*)out_pre; }
    int rc = gemm_f32(g, s);
    if (rc == RVLM_OK && epi == EPI_BF16_ACT && h->cur_dact_out && out_pre) {      // a forward for a peer on the fp32 tiles: act'(h) for it
        rc = x3_export_bf16((const float*)out_pre, ldo, h->cur_dact_out, ldo, M, N, h->cfg.act, 1, s);
        h->exported_dact = true;
    }
    return rc;
}
// dgrad: out[M,K] = epi(dY[M,N] @ W[N,K])   (w_t = W^T stored [K,N] for the bf16 path)
template <typename T>
static int linear_dgrad(rvlm_vit* h, hipStream_t s, const void* dY, long lddy, int M, int N, int K,
                        const float* w_f32, long ldw, const bf16_t* w_t, int epi, void* out, long ldo,
                        const void* h_pre, int a_rows = 0);
template <>
int linear_dgrad<bf16_t>(rvlm_vit* h, hipStream_t s, const void* dY, long lddy, int M, int N, int K,
                         const float*, long, const bf16_t* w_t, int epi, void* out, long ldo,
                         const void* h_pre, int a_rows) {
    GemmBf16 g;
    g.A = (const bf16_t*)dY; g.lda = lddy; g.Bw = w_t; g.ldb = N; g.M = M; g.N = K; g.K = N;
    g.a_rows = a_rows > 0 ? a_rows : (int)round_up(M, 128); g.epi = epi; g.out = out; g.ldo = ldo;
    g.h_pre = (const bf16_t*)h_pre; g.act = h->cfg.act;
    with_scratch(h, g);
    return gemm_bf16_nt(g, s);
}
template <>
int linear_dgrad<float>(rvlm_vit* h, hipStream_t s, const void* dY, long lddy, int M, int N, int K,
                        const float* w_f32, long ldw, const bf16_t* w_t, int epi, void* out, long ldo,
                        const void* h_pre, int) {
    if (x3_takes(h, M, K, N) && w_t) {      // out[M, K] = dY[M, N] W[N, K]: contraction over N, B operand = [W^T_hi | W^T_lo | W^T_hi]
        int rc = RVLM_OK;
        if (h->a3_of != dY && (rc = x3_split_rows((const float*)dY, lddy, h->a3, M, (int)round_up(M, 256), N, s))) return rc;
        h->a3_of = nullptr;
        GemmBf16 g;
        g.A = h->a3; g.lda = 3L * N; g.Bw = w_t; g.ldb = 3L * N; g.M = M; g.N = K; g.K = 3 * N;
        g.a_rows = (int)round_up(M, 256); g.epi = EPI_F32; g.out = out; g.ldo = ldo; g.act = h->cfg.act;
        with_scratch(h, g);
        if ((rc = gemm_bf16_nt(g, s))) return rc;
        if (epi == EPI_BF16_DACT) {      // out * act'(h): straight into the split copy the next dgrad (fc1: dY = `out`) reads
            const bool fuse = x3_takes(h, M, 256, K);
            if ((rc = x3_act((const float*)h_pre, ldo, (float*)out, ldo, fuse ? h->a3 : nullptr, M, (int)round_up(M, 256), K, h->cfg.act, 1, s)))
                return rc;
            if (fuse) h->a3_of = out;
        }
        return RVLM_OK;
    }
    GemmF32 g;
    g.A = (const float*)dY; g.sam = lddy; g.sak = 1;
    g.B = w_f32; g.sbn = 1; g.sbk = ldw;          // (n = k_out, k = n_in) at n_in*ldw + k_out
    g.C = (float*)out; g.scm = ldo; g.scn = 1;
    g.M = M; g.N = K; g.K = N;
    if (epi == EPI_BF16_DACT) { g.dact_h = (const float*)h_pre; g.dact_kind = h->cfg.act; }
    return gemm_f32(g, s);
}

// x3 mode: the LayerNorm in front of a linear writes that linear's split operand itself (a3_of = the fp32 buffer the linear will be
// handed, which is then never written: its only other reader is the weight gradient, which this precision does not have)
static bool ln_into_split(rvlm_vit* h, hipStream_t s, const float* x, const float* gamma, const float* beta, const void* ln_out,
                          float* mean, float* rstd, int M, int N_next) {
    if (!x3_takes(h, M, N_next, h->W)) return false;
    if (!x3_layernorm_fwd(x, h->W, gamma, beta, h->a3, mean, rstd, M, (int)round_up(M, 256), h->W, s)) return false;
    h->a3_of = ln_out;
    return true;
}

// ---- attention (dispatch on precision) -------------------------------------------------------
template <typename T>
static int attention_fwd(rvlm_vit* h, hipStream_t s, const void* qkv, void* o, float* lse, int B);
template <>
int attention_fwd<bf16_t>(rvlm_vit* h, hipStream_t s, const void* qkv, void* o, float* lse, int B) {
    return attn_fwd_bf16((const bf16_t*)qkv, 3 * h->W, (bf16_t*)o, h->W, lse, B, h->H, h->S, s);
}
// fp32 mode: the score matrices are materialised as [B, H, S, Sld] with Sld = round_up(S, 4) and zero pad columns (never
// written), so that every product that contracts over or runs along the key dimension of S = 257 takes 16-byte operand loads
// on the fp32 matrix-pipe tiles (GemmF32::pad4); round 5: the unaligned scalar-load form ran the attention core at 26 TFLOP/s.
// P = softmax(0.125 Q K^T) of one block into `P` ([B, H, S, Sld]): the block's slot of rvlm_vit::lse, which in fp32 mode holds
// the probabilities themselves - the backward reads them back instead of recomputing scores + softmax (round 5: 13 GB for
// ViT-L/14 at B = 128 of 288, one batched GEMM + one softmax pass less per block and backward)
static int attn_scores_f32(rvlm_vit* h, hipStream_t s, const float* qkv, float* P, int B) {
    const int S = h->S, W = h->W, H = h->H, Sld = (int)round_up(S, 4);
    GemmF32 g;  // scores = 0.125 * Q K^T
    g.A = qkv; g.sam = 3 * W; g.sak = 1; g.sab1 = (long)S * 3 * W; g.sab2 = 64;
    g.B = qkv + W; g.sbn = 3 * W; g.sbk = 1; g.sbb1 = (long)S * 3 * W; g.sbb2 = 64;
    g.C = P; g.scm = Sld; g.scn = 1; g.scb1 = (long)H * S * Sld; g.scb2 = (long)S * Sld;
    g.M = S; g.N = S; g.K = 64; g.nb1 = B; g.nb2 = H; g.alpha = 0.125f;
    int rc = gemm_f32(g, s); if (rc) return rc;
    return softmax_rows_fwd(P, (long)B * H * S, S, Sld, s, h->cur_lse2, (int)round_up(S, 32));
}
template <>
int attention_fwd<float>(rvlm_vit* h, hipStream_t s, const void* qkv_, void* o, float* P, int B) {
    const int S = h->S, W = h->W, H = h->H, Sld = (int)round_up(S, 4);
    const float* qkv = (const float*)qkv_;
    int rc = RVLM_OK;
    if (h->cur_flash) {      // no backward of this handle's own will read P: one fused kernel, no score matrices
        if (attn_fwd_f32_flash(qkv, (float*)o, h->cur_lse2, (int)round_up(S, 32), h->cur_qkv_bf, h->cur_o_bf, B, H, S, s, &rc)) return rc;
        if (h->peer || h->own_flash)      // (no fall-through: there is no probability buffer behind P on these paths)
            return fail(RVLM_ERR_UNSUPPORTED, "attention forward: sequence length not covered by the fp32 flash kernel");
    }
    rc = attn_scores_f32(h, s, qkv, P, B); if (rc) return rc;
    GemmF32 g;  // O = P V
    g.A = P; g.sam = Sld; g.sak = 1; g.sab1 = (long)H * S * Sld; g.sab2 = (long)S * Sld; g.pad4 = 1;
    g.B = qkv + 2 * W; g.sbn = 1; g.sbk = 3 * W; g.sbb1 = (long)S * 3 * W; g.sbb2 = 64;
    g.C = (float*)o; g.scm = W; g.scn = 1; g.scb1 = (long)S * W; g.scb2 = 64;
    g.M = S; g.N = 64; g.K = S; g.nb1 = B; g.nb2 = H;
    return gemm_f32(g, s);
}
template <typename T>
static int attention_bwd(rvlm_vit* h, hipStream_t s, const void* qkv, const void* o, const void* d_o,
                         const float* lse, void* dqkv, int B);
template <>
int attention_bwd<bf16_t>(rvlm_vit* h, hipStream_t s, const void* qkv, const void* o, const void* d_o,
                          const float* lse, void* dqkv, int B) {
    return attn_bwd_bf16((const bf16_t*)qkv, 3 * h->W, (const bf16_t*)o, h->W, (const bf16_t*)d_o, h->W,
                         lse, h->dsum, (bf16_t*)dqkv, 3 * h->W, B, h->H, h->S, s);
}
template <>
int attention_bwd<float>(rvlm_vit* h, hipStream_t s, const void* qkv_, const void* o_, const void* d_o_,
                         const float* P, void* dqkv_, int B) {
    const int S = h->S, W = h->W, H = h->H, Sld = (int)round_up(S, 4);
    const float* qkv = (const float*)qkv_;
    const float* d_o = (const float*)d_o_;
    float* dqkv = (float*)dqkv_;
    int rc = RVLM_OK;                                             // (P: kept by the forward)
    if (h->own_flash) {      // probabilities recomputed from the forward's log-sum-exp rows (cur_lse2: the block's)
        if (attn_bwd_f32_flash(qkv, (const float*)o_, d_o, h->cur_lse2, (int)round_up(S, 32), h->dsum, dqkv, B, H, S, s, &rc)) return rc;
        return fail(RVLM_ERR_STATE, "attention backward: flash kernels refused a sequence length they cover");
    }
    const long bs1 = (long)H * S * Sld, bs2 = (long)S * Sld, qs1 = (long)S * 3 * W, os1 = (long)S * W;
    GemmF32 g;  // dP = dO V^T
    g.A = d_o; g.sam = W; g.sak = 1; g.sab1 = os1; g.sab2 = 64;
    g.B = qkv + 2 * W; g.sbn = 3 * W; g.sbk = 1; g.sbb1 = qs1; g.sbb2 = 64;
    g.C = h->dscores; g.scm = Sld; g.scn = 1; g.scb1 = bs1; g.scb2 = bs2;
    g.M = S; g.N = S; g.K = 64; g.nb1 = B; g.nb2 = H;
    if ((rc = gemm_f32(g, s))) return rc;
    if ((rc = softmax_rows_bwd(P, h->dscores, (long)B * H * S, S, Sld, 0.125f, s))) return rc;
    GemmF32 q;  // dQ = dS K
    q.A = h->dscores; q.sam = Sld; q.sak = 1; q.sab1 = bs1; q.sab2 = bs2; q.pad4 = 1;
    q.B = qkv + W; q.sbn = 1; q.sbk = 3 * W; q.sbb1 = qs1; q.sbb2 = 64;
    q.C = dqkv; q.scm = 3 * W; q.scn = 1; q.scb1 = qs1; q.scb2 = 64;
    q.M = S; q.N = 64; q.K = S; q.nb1 = B; q.nb2 = H;
    if ((rc = gemm_f32(q, s))) return rc;
    GemmF32 k;  // dK = dS^T Q
    k.A = h->dscores; k.sam = 1; k.sak = Sld; k.sab1 = bs1; k.sab2 = bs2; k.pad4 = 1;
    k.B = qkv; k.sbn = 1; k.sbk = 3 * W; k.sbb1 = qs1; k.sbb2 = 64;
    k.C = dqkv + W; k.scm = 3 * W; k.scn = 1; k.scb1 = qs1; k.scb2 = 64;
    k.M = S; k.N = 64; k.K = S; k.nb1 = B; k.nb2 = H;
    if ((rc = gemm_f32(k, s))) return rc;
    GemmF32 v;  // dV = P^T dO
    v.A = P; v.sam = 1; v.sak = Sld; v.sab1 = bs1; v.sab2 = bs2; v.pad4 = 1;
    v.B = d_o; v.sbn = 1; v.sbk = W; v.sbb1 = os1; v.sbb2 = 64;
    v.C = dqkv + 2 * W; v.scm = 3 * W; v.scn = 1; v.scb1 = qs1; v.scb2 = 64;
    v.M = S; v.N = 64; v.K = S; v.nb1 = B; v.nb2 = H;
    return gemm_f32(v, s);
}

// ---- QKV projection of the last block under the class-token tail: K, V for every token, Q for the class rows only
// (the other 256 query rows of that block are dead).  Forward: qkv[:, W:3W] by the big GEMM (N = 2W), then the class
// rows' Q by a B-row GEMM.  Backward: d_ln = dqkv[:, W:3W] @ W_in[W:3W] for every row (K = 2W), then the class rows are
// overwritten with their full product (K = 3W).
static int qkv_tail_fwd(rvlm_vit* h, hipStream_t s, const void* ln1o, const Layer& y, void* qkv, int B) {
    const int S = h->S, W = h->W, M = B * S;
    GemmBf16 g;
    g.A = (const bf16_t*)ln1o; g.lda = W; g.Bw = y.w_in_nk + (size_t)W * W; g.ldb = W;
    g.M = M; g.N = 2 * W; g.K = W; g.a_rows = (int)round_up(M, 128); g.epi = EPI_BF16;
    g.bias = y.b_in + W; g.out = (bf16_t*)qkv + W; g.ldo = 3 * W; g.act = h->cfg.act;
    with_scratch(h, g);
    int rc = gemm_bf16_nt(g, s);
    if (rc) return rc;
    GemmBf16 q;
    q.A = (const bf16_t*)ln1o; q.lda = (long)S * W; q.Bw = y.w_in_nk; q.ldb = W;
    q.M = B; q.N = W; q.K = W; q.a_rows = B; q.epi = EPI_BF16;
    q.bias = y.b_in; q.out = qkv; q.ldo = (long)S * 3 * W; q.act = h->cfg.act;
    with_scratch(h, q);
    return gemm_bf16_nt(q, s);
}
static int qkv_tail_bwd(rvlm_vit* h, hipStream_t s, const void* dqkv, const Layer& y, void* d_ln, int B) {
    const int S = h->S, W = h->W, M = B * S;
    GemmBf16 g;
    g.A = (const bf16_t*)dqkv + W; g.lda = 3 * W; g.Bw = y.w_in_t + W; g.ldb = 3 * W;
    g.M = M; g.N = W; g.K = 2 * W; g.a_rows = (int)round_up(M, 128); g.epi = EPI_BF16;
    g.out = d_ln; g.ldo = W; g.act = h->cfg.act;
    with_scratch(h, g);
    int rc = gemm_bf16_nt(g, s);
    if (rc) return rc;
    GemmBf16 q;
    q.A = (const bf16_t*)dqkv; q.lda = (long)S * 3 * W; q.Bw = y.w_in_t; q.ldb = 3 * W;
    q.M = B; q.N = W; q.K = 3 * W; q.a_rows = B; q.epi = EPI_BF16;
    q.out = d_ln; q.ldo = (long)S * W; q.act = h->cfg.act;
    with_scratch(h, q);
    return gemm_bf16_nt(q, s);
}

// ---- forward -----------------------------------------------------------------------------------
template <typename T>
static int forward_impl(rvlm_vit* h, const float* x, const float* delta, int B, int normalize, int save,
                        float* out_emb, hipStream_t s) {
    const int S = h->S, W = h->W, L = h->L, D = h->D, M = B * S, M0 = B * h->G * h->G;
    const double attn_flops = 4.0 * B * h->H * (double)S * S * 64;
    int rc;
    // EVERY forward overwrites state a pending backward reads (the shared LayerNorm statistics of all layers, slot 0 of
    // xs / qkv / attn_o / lse, pooled, emb_raw, inv_norm): a non-saving pass therefore invalidates the saved forward
    // instead of letting a later backward run on a mixture of two passes.
    h->saved_B = 0;
    // a forward FOR a peer (handoff) is a saving forward of an fp32-storage handle; it overwrites the peer's bf16 tensors
    // (sequence lengths the flash kernel does not take - S = 577 at 336 px - run the batched attention path as before: the handoff then
    // exports the bf16 tensors itself and takes the log-sum-exp rows the softmax pass wrote, vit_backward_from)
    const bool flash_ok = !h->bf16 && attn_fwd_f32_flash_covers(h->S);
    rvlm_vit* const peer = (flash_ok && save == 1) ? h->peer : nullptr;
    if (h->provider && save != 0 && !peer)
        return fail(RVLM_ERR_STATE, "forward: this handle was created as a forward provider (trainable = -2): its saving forwards are run "
                                    "for a bf16 handle (rvlm_vit_forward_for, rvlm_pgd_run_mixed_fwd)");
    if (peer) peer->saved_B = 0;
    h->cur_flash = flash_ok && (h->own_flash || peer != nullptr || (save == 0 && h->flash_inference));
    h->exported_to = 0; h->exported_dact = false;
    h->cur_qkv_bf = h->cur_o_bf = nullptr; h->cur_dact_out = nullptr;
    {
        PROF("patch_im2col", 0, (double)B * 3 * h->img * h->img * (delta ? 8 : 4) + (double)M0 * h->Kpad * sizeof(T));
        if ((rc = im2col_normalize<T>(x, delta, B, h->img, h->P, h->cfg.mean, h->cfg.std, (T*)h->A0, h->Kpad, h->Kpad, s))) return rc;
    }
    {
        PROF("gemm_patch_fwd", 2.0 * M0 * W * h->Kp, 0);
        if (h->bf16) {
            GemmBf16 g;
            g.A = (const bf16_t*)h->A0; g.lda = h->Kpad; g.Bw = h->conv_nk; g.ldb = h->Kpad;
            g.M = M0; g.N = W; g.K = h->Kpad; g.a_rows = h->Mp0; g.epi = EPI_F32; g.out = h->patch_out; g.ldo = W;
            with_scratch(h, g);
            rc = gemm_bf16_nt(g, s);
        } else {
            rc = linear_fwd<float>(h, s, h->A0, h->Kpad, M0, W, h->Kp, h->conv_f32, nullptr, nullptr, EPI_F32,
                                   h->patch_out, W, nullptr, nullptr);
        }
        if (rc) return rc;
    }
    auto XS = [&](int i) { return h->xs[save ? i : 0]; };
    {
        PROF("embed_lnpre_fwd", 0, (double)M * W * 8);
        if ((rc = embed_lnpre_fwd<float>(h->patch_out, W, h->cls, h->pos, h->lnpre_w, h->lnpre_b, XS(0), W,
                                         h->mean_at(0), h->rstd_at(0), B, S, W, s,
                                         save == 2 ? h->tokens : nullptr))) return rc;
    }
    for (int l = 0; l < L; ++l) {
        Layer& y = h->layers[l];
        const int sl = save ? l : 0;
        float* x_in = XS(2 * l); float* x_mid = XS(2 * l + 1); float* x_out = XS(2 * l + 2);
        void* ln1o = save == 2 ? h->ln1_out[l] : h->ln_out;     // linear inputs are kept only for wgrad
        void* ln2o = save == 2 ? h->ln2_out[l] : h->ln_out;
        void* gact = save == 2 ? h->g_act_l[l] : h->g_act;
        {
            PROF("layernorm_fwd", 0, (double)M * W * (4 + sizeof(T)));
            if (ln_into_split(h, s, x_in, y.ln1_w, y.ln1_b, ln1o, h->mean_at(1 + 2 * l), h->rstd_at(1 + 2 * l), M, 3 * W)) {}
            else
            if ((rc = layernorm_fwd<T>(x_in, W, y.ln1_w, y.ln1_b, (T*)ln1o, W, h->mean_at(1 + 2 * l),
                                       h->rstd_at(1 + 2 * l), M, W, s))) return rc;
        }
        if (h->cls_tail && l == L - 1) {
            PROF("gemm_qkv_fwd", 2.0 * M * W * 2 * W + 2.0 * B * W * W, 0);
            if ((rc = qkv_tail_fwd(h, s, ln1o, y, h->qkv[sl], B))) return rc;
        } else {
            PROF("gemm_qkv_fwd", 2.0 * M * W * 3 * W, 0);
            if ((rc = linear_fwd<T>(h, s, ln1o, W, M, 3 * W, W, y.w_in, y.w_in_nk, y.b_in, EPI_BF16,
                                    h->qkv[sl], 3 * W, nullptr, nullptr))) return rc;
        }
        if (h->cls_tail && l == L - 1) {
            const long SW = (long)S * W;
            {
                PROF("tail_attn_fwd", 4.0 * B * h->H * (double)S * 64, (double)M * 2 * W * sizeof(T));
                if ((rc = attn_cls_fwd_bf16((const bf16_t*)h->qkv[sl], 3 * W, (bf16_t*)h->attn_o[sl], W, h->lse_cls, B, h->H,
                                            S, s))) return rc;
            }
            {
                PROF("tail_mlp_fwd", 2.0 * B * W * 9 * W, 0);
                if ((rc = linear_fwd<T>(h, s, h->attn_o[sl], W, B, W, W, y.w_out, y.w_out_nk, y.b_out, EPI_F32_RESID,
                                        x_mid, SW, nullptr, x_in))) return rc;
                if ((rc = layernorm_fwd<T>(x_mid, SW, y.ln2_w, y.ln2_b, (T*)ln2o, W, h->mean_at(2 + 2 * l),
                                           h->rstd_at(2 + 2 * l), B, W, s))) return rc;
                if ((rc = linear_fwd<T>(h, s, ln2o, W, B, 4 * W, W, y.w_fc, y.w_fc_nk, y.b_fc, EPI_BF16_ACT,
                                        gact, 4 * W, save ? h->h_pre[sl] : nullptr, nullptr))) return rc;
                if ((rc = linear_fwd<T>(h, s, gact, 4 * W, B, W, 4 * W, y.w_proj, y.w_proj_nk, y.b_proj,
                                        EPI_F32_RESID, x_out, SW, nullptr, x_mid))) return rc;
            }
            continue;
        }
        {
            PROF("attn_fwd", attn_flops, 0);
            h->cur_lse2 = (save && !h->lse2.empty()) ? h->lse2[sl] : nullptr;
            if (peer) { h->cur_qkv_bf = (bf16_t*)peer->qkv[l]; h->cur_o_bf = (bf16_t*)peer->attn_o[l]; }   // (the lse rows stay in h->lse2)
            if ((rc = attention_fwd<T>(h, s, h->qkv[sl], h->attn_o[sl], h->lse[sl], B))) return rc;
        }
        {
            PROF("gemm_out_fwd", 2.0 * M * W * W, 0);
            if ((rc = linear_fwd<T>(h, s, h->attn_o[sl], W, M, W, W, y.w_out, y.w_out_nk, y.b_out, EPI_F32_RESID,
                                    x_mid, W, nullptr, x_in))) return rc;
        }
        {
            PROF("layernorm_fwd", 0, (double)M * W * (4 + sizeof(T)));
            if (ln_into_split(h, s, x_mid, y.ln2_w, y.ln2_b, ln2o, h->mean_at(2 + 2 * l), h->rstd_at(2 + 2 * l), M, 4 * W)) {}
            else
            if ((rc = layernorm_fwd<T>(x_mid, W, y.ln2_w, y.ln2_b, (T*)ln2o, W, h->mean_at(2 + 2 * l),
                                       h->rstd_at(2 + 2 * l), M, W, s))) return rc;
        }
        {
            PROF("gemm_fc1_fwd", 2.0 * M * W * 4 * W, 0);
            h->cur_dact_out = peer ? (bf16_t*)peer->h_pre[l] : nullptr;
            rc = linear_fwd<T>(h, s, ln2o, W, M, 4 * W, W, y.w_fc, y.w_fc_nk, y.b_fc, EPI_BF16_ACT,
                               gact, 4 * W, save ? h->h_pre[sl] : nullptr, nullptr);
            h->cur_dact_out = nullptr;
            if (rc) return rc;
        }
        {
            PROF("gemm_fc2_fwd", 2.0 * M * W * 4 * W, 0);
            if ((rc = linear_fwd<T>(h, s, gact, 4 * W, M, W, 4 * W, y.w_proj, y.w_proj_nk, y.b_proj,
                                    EPI_F32_RESID, x_out, W, nullptr, x_mid))) return rc;
        }
    }
    {
        PROF("head_fwd", 2.0 * B * W * D, 0);
        float* xf = XS(2 * L);
        if ((rc = layernorm_fwd<float>(xf, (long)S * W, h->lnpost_w, h->lnpost_b, h->pooled, W,
                                       h->mean_at(2 * L + 1), h->rstd_at(2 * L + 1), B, W, s))) return rc;
        GemmF32 g;
        g.A = h->pooled; g.sam = W; g.sak = 1;
        g.B = h->proj; g.sbn = 1; g.sbk = D;
        g.C = normalize ? h->emb_raw : out_emb; g.scm = D; g.scn = 1;
        g.M = B; g.N = D; g.K = W;
        if ((rc = gemm_f32(g, s))) return rc;
        if (normalize) {
            if ((rc = l2_normalize_fwd(h->emb_raw, out_emb, h->inv_norm, B, D, s))) return rc;
        }
    }
    if (save) { h->saved_B = B; h->saved_norm = normalize != 0; h->saved_mode = save; h->next_param_stage = 0; }
    if (peer) {      // (saved_mode 3: for the peer's backward only - a provider's single slots / no probabilities kept for a batched backward)
        if (h->provider || !h->own_flash) h->saved_mode = 3;
        h->exported_to = peer->uid;
    }
    h->cur_flash = false; h->cur_qkv_bf = h->cur_o_bf = nullptr; h->cur_lse2 = nullptr;
    return RVLM_OK;
}

static int ew_blocks(size_t n);
__global__ void __launch_bounds__(256)
widen_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = *(const bf16x8*)(src + 8 * i);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
        *(float4*)(dst + 8 * i) = *(const float4*)&o[0];
        *(float4*)(dst + 8 * i + 4) = *(const float4*)&o[4];
    }
}

// ---- backward (input gradient only) --------------------------------------------------------------
template <typename T>
static int backward_impl(rvlm_vit* h, const float* d_emb, int B, float* grad_x, hipStream_t s, bool fp32_dres = false) {
    const int S = h->S, W = h->W, L = h->L, D = h->D, M = B * S, M0 = B * h->G * h->G;
    const double attn_flops = 8.0 * B * h->H * (double)S * S * 64;
    constexpr bool LP = !std::is_same<T, float>::value;
    // the fp32 residual-gradient stream, or none (LayerNorm backward then accumulates in the bf16 buffer, rvlm_vit::lp_dres)
    float* const dres32 = (LP && h->lp_dres && !fp32_dres) ? nullptr : h->dres;
    int rc;
    {
        PROF("head_bwd", 2.0 * B * W * D, 0);
        const float* d_raw = d_emb;
        if (h->saved_norm) {
            if ((rc = l2_normalize_bwd(d_emb, h->emb_raw, h->inv_norm, h->d_raw, B, D, s))) return rc;
            d_raw = h->d_raw;
        }
        GemmF32 g;  // d_pooled = d_raw @ proj^T
        g.A = d_raw; g.sam = D; g.sak = 1;
        g.B = h->proj; g.sbn = D; g.sbk = 1;
        g.C = h->d_pooled; g.scm = W; g.scn = 1;
        g.M = B; g.N = W; g.K = D;
        if ((rc = gemm_f32(g, s))) return rc;
        if (!h->cls_tail) {   // with the class-token tail the last block's ln_1 backward overwrites the other rows
            if (dres32) RVLM_HIP(hipMemsetAsync(dres32, 0, (size_t)h->Mp * W * 4, s));
            if (LP) RVLM_HIP(hipMemsetAsync(h->dres_lp, 0, (size_t)h->Mp * W * sizeof(T), s));
        }
        if ((rc = layernorm_bwd<float, T>(h->d_pooled, W, h->xs[2 * L], (long)S * W, h->lnpost_w,
                                          h->mean_at(2 * L + 1), h->rstd_at(2 * L + 1), dres32, (long)S * W,
                                          LP ? (T*)h->dres_lp : nullptr, (long)S * W, 0, B, W, s))) return rc;
    }
    const void* dres_A = LP ? (const void*)h->dres_lp : (const void*)h->dres;
    for (int l = L - 1; l >= 0; --l) {
        Layer& y = h->layers[l];
        if (h->cls_tail && l == L - 1) {
            const long SW = (long)S * W;
            {
                PROF("tail_mlp_bwd", 2.0 * B * W * 9 * W, 0);
                if ((rc = linear_dgrad<T>(h, s, dres_A, SW, B, W, 4 * W, y.w_proj, 4 * W, y.w_proj_t, EPI_BF16_DACT,
                                          h->dh, 4 * W, h->h_pre[l], B))) return rc;
                if ((rc = linear_dgrad<T>(h, s, h->dh, 4 * W, B, 4 * W, W, y.w_fc, W, y.w_fc_t, EPI_BF16, h->d_ln, W,
                                          nullptr))) return rc;
                if ((rc = layernorm_bwd<T, T>((const T*)h->d_ln, W, h->xs[2 * l + 1], SW, y.ln2_w, h->mean_at(2 + 2 * l),
                                              h->rstd_at(2 + 2 * l), dres32, SW, (T*)h->dres_lp, SW, 1, B, W, s)))
                    return rc;
                if ((rc = linear_dgrad<T>(h, s, dres_A, SW, B, W, W, y.w_out, W, y.w_out_t, EPI_BF16, h->d_o, W,
                                          nullptr, B))) return rc;
            }
            {
                PROF("tail_attn_bwd", 12.0 * B * h->H * (double)S * 64, (double)M * 5 * W * sizeof(T));
                if ((rc = attn_cls_bwd_bf16((const bf16_t*)h->qkv[l], 3 * W, (const bf16_t*)h->attn_o[l], W,
                                            (const bf16_t*)h->d_o, W, h->lse_cls, (bf16_t*)h->dqkv, 3 * W, B, h->H, S, s)))
                    return rc;
            }
        } else {
        {
            PROF("gemm_fc2_bwd", 2.0 * M * W * 4 * W, 0);
            if ((rc = linear_dgrad<T>(h, s, dres_A, W, M, W, 4 * W, y.w_proj, 4 * W, y.w_proj_t, EPI_BF16_DACT,
                                      h->dh, 4 * W, h->h_pre[l]))) return rc;
        }
        {
            PROF("gemm_fc1_bwd", 2.0 * M * W * 4 * W, 0);
            if ((rc = linear_dgrad<T>(h, s, h->dh, 4 * W, M, 4 * W, W, y.w_fc, W, y.w_fc_t, EPI_BF16, h->d_ln, W,
                                      nullptr))) return rc;
        }
        {
            PROF("layernorm_bwd", 0, (double)M * W * ((dres32 ? 12 : 6) + 2 * sizeof(T)));
            if ((rc = layernorm_bwd<T, T>((const T*)h->d_ln, W, h->xs[2 * l + 1], W, y.ln2_w, h->mean_at(2 + 2 * l),
                                          h->rstd_at(2 + 2 * l), dres32, W, LP ? (T*)h->dres_lp : nullptr, W, 1,
                                          M, W, s))) return rc;
        }
        {
            PROF("gemm_out_bwd", 2.0 * M * W * W, 0);
            if ((rc = linear_dgrad<T>(h, s, dres_A, W, M, W, W, y.w_out, W, y.w_out_t, EPI_BF16, h->d_o, W,
                                      nullptr))) return rc;
        }
        {
            PROF("attn_bwd", attn_flops, 0);
            h->cur_lse2 = h->lse2.empty() ? nullptr : h->lse2[l];
            if ((rc = attention_bwd<T>(h, s, h->qkv[l], h->attn_o[l], h->d_o, h->lse[l], h->dqkv, B))) return rc;
        }
        }
        if (h->cls_tail && l == L - 1) {
            PROF("gemm_qkv_bwd", 2.0 * M * W * 2 * W + 2.0 * B * W * 3 * W, 0);
            if ((rc = qkv_tail_bwd(h, s, h->dqkv, y, h->d_ln, B))) return rc;
        } else {
            PROF("gemm_qkv_bwd", 2.0 * M * W * 3 * W, 0);
            if ((rc = linear_dgrad<T>(h, s, h->dqkv, 3 * W, M, 3 * W, W, y.w_in, W, y.w_in_t, EPI_BF16, h->d_ln, W,
                                      nullptr))) return rc;
        }
        {
            PROF("layernorm_bwd", 0, (double)M * W * ((dres32 ? 12 : 6) + 2 * sizeof(T)));
            if ((rc = layernorm_bwd<T, T>((const T*)h->d_ln, W, h->xs[2 * l], W, y.ln1_w, h->mean_at(1 + 2 * l),
                                          h->rstd_at(1 + 2 * l), dres32, W, LP ? (T*)h->dres_lp : nullptr, W,
                                          (h->cls_tail && l == L - 1) ? -S : 1, M, W, s))) return rc;
        }
    }
    {
        PROF("embed_lnpre_bwd", 0, (double)M * W * 8);
        if (!dres32) {      // (the embedding backward reads the residual gradient as fp32: one widening pass per backward, 40 us)
            hipLaunchKernelGGL(widen_bf16_kernel, dim3(ew_blocks((size_t)M * W / 8)), dim3(256), 0, s, (const bf16_t*)h->dres_lp, h->dres, (size_t)M * W / 8);
            RVLM_CHECK_LAUNCH();
        }
        if ((rc = embed_lnpre_bwd<T>(h->dres, W, h->patch_out, W, h->cls, h->pos, h->lnpre_w, h->mean_at(0),
                                     h->rstd_at(0), (T*)h->d_patch, W, B, S, W, s))) return rc;
    }
    {
        PROF("gemm_patch_bwd", 2.0 * M0 * W * h->Kp, 0);
        if (h->bf16) {
            GemmBf16 g;
            g.A = (const bf16_t*)h->d_patch; g.lda = W; g.Bw = h->conv_t; g.ldb = W;
            g.M = M0; g.N = h->Kpad; g.K = W; g.a_rows = h->Mp0; g.epi = EPI_F32; g.out = h->dA0; g.ldo = h->Kpad;
            with_scratch(h, g);
            rc = gemm_bf16_nt(g, s);
        } else {
            GemmF32 g;
            g.A = (const float*)h->d_patch; g.sam = W; g.sak = 1;
            g.B = h->conv_f32; g.sbn = 1; g.sbk = h->Kp;
            g.C = h->dA0; g.scm = h->Kpad; g.scn = 1;
            g.M = M0; g.N = h->Kp; g.K = W;
            rc = gemm_f32(g, s);
        }
        if (rc) return rc;
    }
    {
        PROF("patch_col2im", 0, (double)B * 3 * h->img * h->img * 8);
        if ((rc = col2im_grad<float>(h->dA0, h->Kpad, B, h->img, h->P, h->cfg.std, grad_x, s))) return rc;
    }
    return RVLM_OK;
}

// ---- weight gradients ------------------------------------------------------------------------------
// dW[N,K] (+)= dY[M,N]^T @ X[M,K]
// dbias != null: dbias[N] (+)= column sums of dY as well (fused into the operand transpose where that exists)
template <typename T>
static int wgrad(rvlm_vit* h, hipStream_t s, const void* dY, long lddy, const void* X, long ldx, int M, int N, int K,
                 float* dW, long lddw, int accumulate, float* dbias = nullptr);
template <>
int wgrad<bf16_t>(rvlm_vit* h, hipStream_t s, const void* dY, long lddy, const void* X, long ldx, int M, int N,
                  int K, float* dW, long lddw, int accumulate, float* dbias) {
    // contraction over the token dimension.  Encoder shapes: the persistent kernel's contraction-major form reads dY and X as the
    // backward left them (token-major), split-K over token chunks in one batched launch - no transposed copies.
    int rc, splits = 0, Kc = 0;
#ifdef RVLM_EXPERIMENTAL_GEMM     // A/B arm (make EXPERIMENTAL=1, RVLM_WGRAD_TRANSPOSED=1): the token-chunk transposes + NT GEMM of round 3
    static int transposed = -1;
    if (transposed < 0) { const char* e = getenv("RVLM_WGRAD_TRANSPOSED"); transposed = e ? atoi(e) : 0; }
    if (transposed && wgrad_split_plan(M, N, K, h->splitk_bytes, &splits, &Kc) && (long)splits * Kc <= h->Mpt) {
        if ((rc = transpose_split((const bf16_t*)dY, lddy, M, N, (bf16_t*)h->tA, Kc, splits, dbias, accumulate, h->red_scratch,
                                  h->red_floats, s))) return rc;
        if ((rc = transpose_split((const bf16_t*)X, ldx, M, K, (bf16_t*)h->tB, Kc, splits, nullptr, 0, nullptr, 0, s))) return rc;
        return gemm_bf16_wgrad_split((const bf16_t*)h->tA, (const bf16_t*)h->tB, splits, Kc, N, K, dW, lddw, accumulate,
                                     h->splitk_scratch, h->splitk_bytes, s);
    }
#endif
    if (wgrad_split_plan(M, N, K, h->splitk_bytes, &splits, &Kc) && lddy % 8 == 0 && ldx % 8 == 0 &&
        (((size_t)dY | (size_t)X) & 15) == 0) {
        if (dbias && (rc = colsum<bf16_t>((const bf16_t*)dY, lddy, M, N, dbias, accumulate, h->red_scratch, h->red_floats, s)))
            return rc;
        return gemm_bf16_wgrad_tn((const bf16_t*)dY, lddy, (const bf16_t*)X, ldx, M, splits, Kc, N, K, dW, lddw, accumulate,
                                  h->splitk_scratch, h->splitk_bytes, s);
    }
    // other shapes (conv1: K = 3*P*P): whole-K transposes zero-padded to a multiple of 64, NT GEMM as usual
    if (dbias && (rc = colsum<bf16_t>((const bf16_t*)dY, lddy, M, N, dbias, accumulate, h->red_scratch, h->red_floats, s)))
        return rc;
    const int Mk = (int)round_up(M, 64);
    // leading dimension of the transposed copies: all tokens when the operand's rows fit that way (conv1; every linear of a
    // width that is no multiple of 256), else the few tokens of this call (the < 256-token fallback of the encoder linears)
    const size_t need_rows = (size_t)round_up(std::max(N, K), 128);
    long ldt = h->Mpt;
    if (need_rows * (size_t)ldt > h->t_elems) ldt = Mk;
    if (need_rows * (size_t)ldt > h->t_elems || Mk > ldt)
        return fail(RVLM_ERR_STATE, "wgrad: transposed-operand scratch too small for this shape");
    if ((rc = transpose_pad<bf16_t>((const bf16_t*)dY, lddy, M, N, (bf16_t*)h->tA, ldt, Mk, s))) return rc;
    if ((rc = transpose_pad<bf16_t>((const bf16_t*)X, ldx, M, K, (bf16_t*)h->tB, ldt, Mk, s))) return rc;
    GemmBf16 g;
    g.A = (const bf16_t*)h->tA; g.lda = ldt; g.Bw = (const bf16_t*)h->tB; g.ldb = ldt;
    g.M = N; g.N = K; g.K = Mk; g.a_rows = (int)std::min<size_t>(h->t_elems / (size_t)ldt, (size_t)4 * h->W);
    g.epi = accumulate ? EPI_F32_RESID : EPI_F32; g.residual = accumulate ? dW : nullptr;
    g.out = dW; g.ldo = lddw;
    with_scratch(h, g);
    return gemm_bf16_nt(g, s);
}
template <>
int wgrad<float>(rvlm_vit* h, hipStream_t s, const void* dY, long lddy, const void* X, long ldx, int M, int N, int K,
                 float* dW, long lddw, int accumulate, float* dbias) {
    int rc;
    if (dbias && (rc = colsum<float>((const float*)dY, lddy, M, N, dbias, accumulate, h->red_scratch, h->red_floats, s)))
        return rc;
    GemmF32 g;
    g.A = (const float*)dY; g.sam = 1; g.sak = lddy;      // (m_out = n, k = token)
    g.B = (const float*)X; g.sbn = 1; g.sbk = ldx;        // (n_out = k, k = token)
    g.C = dW; g.scm = lddw; g.scn = 1;
    g.M = N; g.N = K; g.K = M;
    g.residual = accumulate ? dW : nullptr;
    return gemm_f32(g, s);
}

// Stages of the parameter backward, in execution order: 0 = head (proj, ln_post), 1 + j = transformer block L-1-j,
// L + 1 = embeddings (ln_pre, positional / class embedding, conv1).  [stage_begin, stage_end) lets the host cut the
// pass into gradient buckets whose all-reduce overlaps the remaining stages (robustvlm_amd/trainer.py).
template <typename T>
static int backward_params_impl(rvlm_vit* h, const float* d_emb, int B, const rvlm_vit_weights* gw, int acc,
                                int stage_begin, int stage_end, hipStream_t s) {
    const int S = h->S, W = h->W, L = h->L, D = h->D, M = B * S, M0 = B * h->G * h->G;
    constexpr bool LP = !std::is_same<T, float>::value;
    auto G = [](const float* p) { return const_cast<float*>(p); };
    int rc;
    const void* dres_A = LP ? (const void*)h->dres_lp : (const void*)h->dres;
    if (stage_begin <= 0 && stage_end > 0) {
    // ---- head ----
    const float* d_raw = d_emb;
    if (h->saved_norm) {
        if ((rc = l2_normalize_bwd(d_emb, h->emb_raw, h->inv_norm, h->d_raw, B, D, s))) return rc;
        d_raw = h->d_raw;
    }
    {   // dproj[W,D] (+)= pooled^T d_raw
        GemmF32 g;
        g.A = h->pooled; g.sam = 1; g.sak = W;
        g.B = d_raw; g.sbn = 1; g.sbk = D;
        g.C = G(gw->proj); g.scm = D; g.scn = 1;
        g.M = W; g.N = D; g.K = B;
        g.residual = acc ? gw->proj : nullptr;
        if ((rc = gemm_f32(g, s))) return rc;
    }
    {   // d_pooled = d_raw @ proj^T
        GemmF32 g;
        g.A = d_raw; g.sam = D; g.sak = 1;
        g.B = h->proj; g.sbn = D; g.sbk = 1;
        g.C = h->d_pooled; g.scm = W; g.scn = 1;
        g.M = B; g.N = W; g.K = D;
        if ((rc = gemm_f32(g, s))) return rc;
    }
    if ((rc = ln_param_grad<float>(h->d_pooled, W, h->xs[2 * L], (long)S * W, h->mean_at(2 * L + 1),
                                   h->rstd_at(2 * L + 1), B, W, G(gw->ln_post_weight), G(gw->ln_post_bias), acc,
                                   h->red_scratch, h->red_floats, s)))
        return rc;
    if (!h->cls_tail) {
        RVLM_HIP(hipMemsetAsync(h->dres, 0, (size_t)h->Mp * W * 4, s));
        if (LP) RVLM_HIP(hipMemsetAsync(h->dres_lp, 0, (size_t)h->Mp * W * sizeof(T), s));
    }
    if ((rc = layernorm_bwd<float, T>(h->d_pooled, W, h->xs[2 * L], (long)S * W, h->lnpost_w, h->mean_at(2 * L + 1),
                                      h->rstd_at(2 * L + 1), h->dres, (long)S * W, LP ? (T*)h->dres_lp : nullptr,
                                      (long)S * W, 0, B, W, s))) return rc;
    }
    for (int l = L - 1; l >= 0; --l) {
        if (L - l < stage_begin || L - l >= stage_end) continue;
        Layer& y = h->layers[l];
        const rvlm_vit_block_weights& gb = gw->blocks_host[l];
        // last block with the class-token tail: only the B class-token rows are live from here up to the attention
        const bool tail = h->cls_tail && l == L - 1;
        const int Mr = tail ? B : M, ar = tail ? B : 0;
        const long ldr = tail ? (long)S * W : (long)W;
        // fc2 (c_proj): dY = d(residual), X = act(fc1)
        if ((rc = wgrad<T>(h, s, dres_A, ldr, h->g_act_l[l], 4 * W, Mr, W, 4 * W, G(gb.mlp_c_proj_weight), 4 * W, acc,
                           G(gb.mlp_c_proj_bias)))) return rc;
        if ((rc = linear_dgrad<T>(h, s, dres_A, ldr, Mr, W, 4 * W, y.w_proj, 4 * W, y.w_proj_t, EPI_BF16_DACT, h->dh,
                                  4 * W, h->h_pre[l], ar))) return rc;
        // fc1 (c_fc)
        if ((rc = wgrad<T>(h, s, h->dh, 4 * W, h->ln2_out[l], W, Mr, 4 * W, W, G(gb.mlp_c_fc_weight), W, acc,
                           G(gb.mlp_c_fc_bias)))) return rc;
        if ((rc = linear_dgrad<T>(h, s, h->dh, 4 * W, Mr, 4 * W, W, y.w_fc, W, y.w_fc_t, EPI_BF16, h->d_ln, W, nullptr)))
            return rc;
        if ((rc = ln_param_grad<T>((const T*)h->d_ln, W, h->xs[2 * l + 1], ldr, h->mean_at(2 + 2 * l),
                                   h->rstd_at(2 + 2 * l), Mr, W, G(gb.ln_2_weight), G(gb.ln_2_bias), acc,
                                   h->red_scratch, h->red_floats, s))) return rc;
        if ((rc = layernorm_bwd<T, T>((const T*)h->d_ln, W, h->xs[2 * l + 1], ldr, y.ln2_w, h->mean_at(2 + 2 * l),
                                      h->rstd_at(2 + 2 * l), h->dres, ldr, LP ? (T*)h->dres_lp : nullptr, ldr, 1, Mr, W, s)))
            return rc;
        // attention out-proj
        if ((rc = wgrad<T>(h, s, dres_A, ldr, h->attn_o[l], W, Mr, W, W, G(gb.attn_out_proj_weight), W, acc,
                           G(gb.attn_out_proj_bias)))) return rc;
        if ((rc = linear_dgrad<T>(h, s, dres_A, ldr, Mr, W, W, y.w_out, W, y.w_out_t, EPI_BF16, h->d_o, W, nullptr, ar)))
            return rc;
        if (tail) {
            if ((rc = attn_cls_bwd_bf16((const bf16_t*)h->qkv[l], 3 * W, (const bf16_t*)h->attn_o[l], W,
                                        (const bf16_t*)h->d_o, W, h->lse_cls, (bf16_t*)h->dqkv, 3 * W, B, h->H, S, s)))
                return rc;
        } else {
            h->cur_lse2 = h->lse2.empty() ? nullptr : h->lse2[l];
            if ((rc = attention_bwd<T>(h, s, h->qkv[l], h->attn_o[l], h->d_o, h->lse[l], h->dqkv, B))) return rc;
        }
        // qkv in-proj
        if ((rc = wgrad<T>(h, s, h->dqkv, 3 * W, h->ln1_out[l], W, M, 3 * W, W, G(gb.attn_in_proj_weight), W, acc,
                           G(gb.attn_in_proj_bias)))) return rc;
        if ((rc = linear_dgrad<T>(h, s, h->dqkv, 3 * W, M, 3 * W, W, y.w_in, W, y.w_in_t, EPI_BF16, h->d_ln, W, nullptr)))
            return rc;
        if ((rc = ln_param_grad<T>((const T*)h->d_ln, W, h->xs[2 * l], W, h->mean_at(1 + 2 * l), h->rstd_at(1 + 2 * l),
                                   M, W, G(gb.ln_1_weight), G(gb.ln_1_bias), acc, h->red_scratch, h->red_floats, s)))
            return rc;
        if ((rc = layernorm_bwd<T, T>((const T*)h->d_ln, W, h->xs[2 * l], W, y.ln1_w, h->mean_at(1 + 2 * l),
                                      h->rstd_at(1 + 2 * l), h->dres, W, LP ? (T*)h->dres_lp : nullptr, W,
                                      tail ? -S : 1, M, W, s)))
            return rc;
    }
    if (stage_end <= L + 1) return RVLM_OK;
    // ---- embeddings: ln_pre, positional / class embedding, conv1 ----
    if ((rc = ln_param_grad<float>(h->dres, W, h->tokens, W, h->mean_at(0), h->rstd_at(0), M, W, G(gw->ln_pre_weight),
                                   G(gw->ln_pre_bias), acc, h->red_scratch, h->red_floats, s))) return rc;
    if ((rc = layernorm_bwd<float, float>(h->dres, W, h->tokens, W, h->lnpre_w, h->mean_at(0), h->rstd_at(0), h->dtok,
                                          W, nullptr, W, 0, M, W, s))) return rc;
    if ((rc = pos_cls_grad(h->dtok, W, B, S, W, G(gw->positional_embedding), G(gw->class_embedding), acc, s))) return rc;
    if ((rc = gather_patch_rows<T>(h->dtok, W, B, S, W, (T*)h->d_patch, W, s))) return rc;
    // conv1.weight [W, 3*P*P] (+)= d_patch^T A0   (A0 has Kpad >= Kp columns; only Kp are weights)
    if ((rc = wgrad<T>(h, s, h->d_patch, W, h->A0, h->Kpad, M0, W, h->Kp, G(gw->conv1_weight), h->Kp, acc))) return rc;
    return RVLM_OK;
}

static int vit_forward(rvlm_vit* h, const float* x, const float* delta, int B, int normalize, int save,
                       float* out_emb, hipStream_t s) {
    return h->bf16 ? forward_impl<bf16_t>(h, x, delta, B, normalize, save, out_emb, s)
                   : forward_impl<float>(h, x, delta, B, normalize, save, out_emb, s);
}
static int vit_backward(rvlm_vit* h, const float* d_emb, int B, float* grad_x, hipStream_t s) {
    if (h->saved_mode == 3) return fail(RVLM_ERR_STATE, "backward: the saved forward was run for another handle's backward (no probabilities kept)");
    return h->bf16 ? backward_impl<bf16_t>(h, d_emb, B, grad_x, s) : backward_impl<float>(h, d_emb, B, grad_x, s);
}

// HANDOFF (round 6): the input gradient of the forward SAVED ON `hx` (an fp32-storage handle: x3 or fp32 precision) evaluated by the
// bf16 handle `hb`'s backward kernels.  FARE's first cotangent 2 (phi(x + d0) - phi(x)) needs a faithful FORWARD - the difference of two
// nearly equal embeddings is where bf16 rounding noise decides a fifth of the first step's signs - but not a faithful backward: the
// cotangent's way back through the network only meets relative rounding errors (oracle/split_bf16_emulation.py: split-bf16 forward +
// bf16 backward 0.998 first-step sign agreement with the fp32 oracle; split-bf16 both ways 0.9996; bf16 both ways 0.82).  What the
// bf16 backward reads as fp32 (residual stream, LayerNorm statistics, patch embeddings, the unnormalised output) it reads from hx's
// buffers in place (pointers swapped for the duration of the enqueue); what it reads in bf16 (qkv, attention output, act'(fc1)) is
// exported here from the fp32 tensors hx kept; the log-sum-exp rows were written by hx's softmax pass in the flash kernels' convention.
// The class-token tail is switched off for this pass (hx ran the last block on every row).
static int vit_backward_from(rvlm_vit* hb, rvlm_vit* hx, const float* d_emb, int B, float* grad_x, hipStream_t s) {
    if (!hb->bf16 || hx->bf16) return fail(RVLM_ERR_STATE, "handoff: needs a bf16 handle and an fp32-storage (fp32 / x3) handle");
    if (hx->saved_B != B || B <= 0) return fail(RVLM_ERR_STATE, "handoff: no saved forward for this batch size on the fp32-storage handle");
    if (hx->lse2.empty() || hb->inference_only) return fail(RVLM_ERR_STATE, "handoff: inference-only handle");
    if (hb->Mp != hx->Mp || hb->Mp0 != hx->Mp0 || hb->W != hx->W || hb->L != hx->L || hb->S != hx->S || hb->H != hx->H ||
        hb->D != hx->D || hb->P != hx->P || hb->cfg.act != hx->cfg.act)
        return fail(RVLM_ERR_STATE, "handoff: the two handles must hold the same architecture and max_batch");
    rvlm_vit* h = hb;                      // (PROF accounts on the bf16 handle)
    const int W = hb->W, L = hb->L, M = B * hb->S;
    int rc;
    {
        PROF("handoff_export", 0, (double)M * W * (3 * 6 + 6 + 4 * 6) * L);
        const bool have_qo = hx->exported_to == hb->uid, have_dact = have_qo && hx->exported_dact;     // written by the forward itself
        if (hx->provider && !(have_qo && have_dact))
            return fail(RVLM_ERR_STATE, "handoff: the forward-provider handle's saved pass was not run for this bf16 handle");
        for (int l = 0; l < L; ++l) {
            if (!have_qo && (rc = x3_export_bf16((const float*)hx->qkv[l], 3 * W, (bf16_t*)hb->qkv[l], 3 * W, M, 3 * W, 0, 0, s))) return rc;
            if (!have_qo && (rc = x3_export_bf16((const float*)hx->attn_o[l], W, (bf16_t*)hb->attn_o[l], W, M, W, 0, 0, s))) return rc;
            if (!have_dact && (rc = x3_export_bf16((const float*)hx->h_pre[l], 4 * W, (bf16_t*)hb->h_pre[l], 4 * W, M, 4 * W, hb->cfg.act, 1, s))) return rc;
        }
    }
    struct Swap {      // hb reads hx's fp32 state in place; restored on every exit path
        rvlm_vit *a, *b;
        Swap(rvlm_vit* a_, rvlm_vit* b_) : a(a_), b(b_) { swap(); }
        ~Swap() { swap(); }
        void swap() {
            std::swap(a->xs, b->xs); std::swap(a->st_mean, b->st_mean); std::swap(a->st_rstd, b->st_rstd);
            std::swap(a->patch_out, b->patch_out); std::swap(a->emb_raw, b->emb_raw); std::swap(a->inv_norm, b->inv_norm);
            std::swap(a->lse, b->lse2);
        }
    } swapped(hb, hx);
    const bool tail = hb->cls_tail;
    const bool sn = hb->saved_norm;
    hb->cls_tail = false; hb->saved_norm = hx->saved_norm;
    rc = backward_impl<bf16_t>(hb, d_emb, B, grad_x, s, /*fp32_dres=*/true);
    hb->cls_tail = tail; hb->saved_norm = sn;
    hb->saved_B = 0;        // (hb's own saved forward, if any, lost its bf16 tensors to the exports)
    return rc;
}

__global__ void __launch_bounds__(256)
add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = a[i] + (b ? b[i] : 0.0f);
}
__global__ void __launch_bounds__(256)
clamp01_kernel(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(a[i], 0.0f), 1.0f);
}
__global__ void init_apgd_state_kernel(int B, const float* loss0, const uint8_t* pred0, float step0,
                                       float* loss_best, float* loss_best_lc, float* reduced_lc, float* step,
                                       uint8_t* acc) {
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        loss_best[b] = loss0[b]; loss_best_lc[b] = loss0[b]; reduced_lc[b] = 1.0f; step[b] = step0;
        acc[b] = pred0[b];
    }
}
static int ew_blocks(size_t n) { size_t b = (n + 1023) / 1024; if (b > 2048) b = 2048; if (b < 1) b = 1; return (int)b; }

}  // namespace rvlm

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int rvlm_vit_create(const rvlm_vit_config* cfg, const rvlm_vit_weights* weights,
                               rvlm_stream_t stream, rvlm_vit** out) {
    RVLM_REQUIRE(cfg && weights && out, "rvlm_vit_create: null argument");
    RVLM_REQUIRE(cfg->patch > 0 && cfg->image_size % cfg->patch == 0, "rvlm_vit_create: image_size % patch");
    RVLM_REQUIRE(cfg->heads > 0 && cfg->width == cfg->heads * 64, "rvlm_vit_create: head_dim must be 64");
    RVLM_REQUIRE(cfg->layers > 0 && cfg->out_dim > 0 && cfg->max_batch > 0, "rvlm_vit_create: sizes");
    RVLM_REQUIRE(cfg->precision == RVLM_PREC_F32 || cfg->precision == RVLM_PREC_BF16 || cfg->precision == RVLM_PREC_F32X3,
                 "rvlm_vit_create: precision");
    RVLM_REQUIRE(cfg->precision != RVLM_PREC_F32X3 || cfg->trainable <= 0, "rvlm_vit_create: the split-bf16 mode is an attack / inference "
                 "precision (no weight gradients)");
    RVLM_REQUIRE(cfg->act == RVLM_ACT_QUICK_GELU || cfg->act == RVLM_ACT_GELU, "rvlm_vit_create: act");
    RVLM_REQUIRE(cfg->width <= 2048, "rvlm_vit_create: width > 2048 unsupported");
    hipStream_t s = (hipStream_t)stream;
    rvlm_vit* h = new rvlm_vit();
    static unsigned long long next_uid = 0;
    h->uid = __atomic_add_fetch(&next_uid, 1, __ATOMIC_RELAXED);
    h->cfg = *cfg;
    h->bf16 = cfg->precision == RVLM_PREC_BF16;
    h->x3 = cfg->precision == RVLM_PREC_F32X3;
    h->esz = h->bf16 ? 2 : 4;
    h->img = cfg->image_size; h->P = cfg->patch; h->G = h->img / h->P; h->S = h->G * h->G + 1;
    h->W = cfg->width; h->L = cfg->layers; h->H = cfg->heads; h->D = cfg->out_dim;
    h->Kp = 3 * h->P * h->P; h->Kpad = (int)round_up(h->Kp, 64);
    h->maxB = cfg->max_batch;
    h->Mp = (int)round_up((long)h->maxB * h->S, 256);
    h->Mp0 = (int)round_up((long)h->maxB * h->G * h->G, 256);
    h->layers.resize(h->L);
    const int W = h->W, L = h->L, S = h->S, B = h->maxB, D = h->D;
    const size_t Mp = h->Mp, Mp0 = h->Mp0, e = h->esz;
    int rc = load_weights(h, weights, s, true);
    if (rc) { rvlm_vit_destroy(h); return rc; }
#define ALLOC_OR_DIE(ptr, bytes) do { rc = dev_alloc(h, (void**)&(ptr), (bytes)); if (rc) { rvlm_vit_destroy(h); return rc; } } while (0)
    ALLOC_OR_DIE(h->A0, Mp0 * h->Kpad * e);
    ALLOC_OR_DIE(h->patch_out, Mp0 * W * 4);
    if (h->x3 && W % 256 == 0) {      // split copy of a linear's input: up to [Mp, 3 * 4W] bf16 (0.8 GB for ViT-L/14 at B = 128)
        rc = dev_alloc(h, (void**)&h->a3, Mp * 12 * W * 2, false);
        if (rc) { rvlm_vit_destroy(h); return rc; }
    }
    // (a provider needs the flash forward for its sequence length: otherwise the handoff exports from per-block fp32 tensors)
    {
        const char* e = getenv("RVLM_DRES_FP32");      // 1: the fp32 residual-gradient stream in every backward (rounds 1-5)
        h->lp_dres = h->bf16 && !(e && atoi(e) != 0);
    }
    {
        const char* e = getenv("RVLM_F32_FLASH");      // 0: the batched attention path over kept probabilities (A/B and parity arm)
        h->own_flash = !h->bf16 && attn_bwd_f32_flash_covers(S) && !(e && atoi(e) == 0);
    }
    const bool provider = cfg->trainable == -2 && !h->bf16 && attn_fwd_f32_flash_covers(S);
    h->provider = provider;
    const bool inference_only = cfg->trainable < 0 && !(cfg->trainable == -2 && !h->bf16);   // no backward of any kind: one slot per buffer kind
    h->inference_only = inference_only;
    h->xs.resize(2 * L + 1);
    for (size_t i = 0; i < h->xs.size(); ++i) {
        if (inference_only && i > 0) { h->xs[i] = h->xs[0]; continue; }
        ALLOC_OR_DIE(h->xs[i], Mp * W * 4);
    }
    ALLOC_OR_DIE(h->st_mean, (size_t)(2 * L + 2) * Mp * 4);
    ALLOC_OR_DIE(h->st_rstd, (size_t)(2 * L + 2) * Mp * 4);
    ALLOC_OR_DIE(h->ln_out, Mp * W * e);
    h->qkv.resize(L); h->attn_o.resize(L); h->lse.resize(L); h->h_pre.resize(L);
    const size_t Sp = round_up(S, 32);
    for (int l = 0; l < L; ++l) {
        if ((inference_only || provider) && l > 0) {
            h->qkv[l] = h->qkv[0]; h->attn_o[l] = h->attn_o[0]; h->lse[l] = h->lse[0]; h->h_pre[l] = h->h_pre[0];
            if (provider) { h->lse2.resize(L); ALLOC_OR_DIE(h->lse2[l], (size_t)B * h->H * Sp * 4); }
            continue;
        }
        ALLOC_OR_DIE(h->qkv[l], Mp * 3 * W * e);
        ALLOC_OR_DIE(h->attn_o[l], Mp * W * e);
        // bf16: log-sum-exp rows of the flash kernels; fp32: the block's probabilities [B, H, S, round_up(S, 4)], zero pad columns - a
        // stub where the fp32 flash kernels run both directions (own_flash: nothing is kept)
        ALLOC_OR_DIE(h->lse[l], h->bf16 ? (size_t)B * h->H * Sp * 4 : h->own_flash ? 256 : (size_t)B * h->H * S * round_up(S, 4) * 4);
        ALLOC_OR_DIE(h->h_pre[l], Mp * 4 * W * e);
        if (!h->bf16 && !inference_only) {      // (2.4 MB per block at ViT-L/14, B = 128)
            h->lse2.resize(L);
            ALLOC_OR_DIE(h->lse2[l], (size_t)B * h->H * Sp * 4);
        }
    }
    ALLOC_OR_DIE(h->g_act, Mp * 4 * W * e);
    ALLOC_OR_DIE(h->pooled, (size_t)B * W * 4);
    ALLOC_OR_DIE(h->emb_raw, (size_t)B * D * 4);
    ALLOC_OR_DIE(h->inv_norm, (size_t)B * 4);
    const size_t bs = provider ? 0 : 1;      // (a provider runs no backward: 256-byte stubs)
    ALLOC_OR_DIE(h->dres, bs * Mp * W * 4);
    ALLOC_OR_DIE(h->dres_lp, bs * Mp * W * e);
    ALLOC_OR_DIE(h->d_o, bs * Mp * W * e);
    ALLOC_OR_DIE(h->dqkv, bs * Mp * 3 * W * e);
    ALLOC_OR_DIE(h->dh, bs * Mp * 4 * W * e);
    ALLOC_OR_DIE(h->d_ln, bs * Mp * W * e);
    ALLOC_OR_DIE(h->d_patch, bs * Mp0 * W * e);
    ALLOC_OR_DIE(h->dA0, bs * Mp0 * h->Kpad * 4);
    ALLOC_OR_DIE(h->dsum, (size_t)B * h->H * Sp * 4);
    ALLOC_OR_DIE(h->d_raw, (size_t)B * D * 4);
    ALLOC_OR_DIE(h->d_pooled, (size_t)B * W * 4);
    if (!h->bf16) {
        // [B, H, S, round_up(S, 4)], zero-initialised: the pad columns are never written
        ALLOC_OR_DIE(h->dscores, (h->own_flash ? 0 : bs) * (size_t)B * h->H * S * round_up(S, 4) * 4);
    } else { h->dscores = nullptr; }
    h->trainable = cfg->trainable > 0;
    h->tokens = h->dtok = nullptr; h->tA = h->tB = nullptr;
    {
        const char* e = getenv("RVLM_CLS_TAIL");    // 0: run the last block on every row like the other blocks
        h->cls_tail = h->bf16 && !(e && atoi(e) == 0);
        if (h->cls_tail) ALLOC_OR_DIE(h->lse_cls, (size_t)B * h->H * 4);
    }
    if (h->trainable) {
        h->ln1_out.resize(L); h->ln2_out.resize(L); h->g_act_l.resize(L);
        for (int l = 0; l < L; ++l) {
            ALLOC_OR_DIE(h->ln1_out[l], Mp * W * e);
            ALLOC_OR_DIE(h->ln2_out[l], Mp * W * e);
            ALLOC_OR_DIE(h->g_act_l[l], Mp * 4 * W * e);
        }
        ALLOC_OR_DIE(h->tokens, Mp * W * 4);
        ALLOC_OR_DIE(h->dtok, Mp * W * 4);
        if (h->bf16) {
            // What the copy-free contraction-major weight gradient leaves to these buffers: conv1 (max(W, Kpad) rows x all tokens),
            // and every linear when fewer than 256 tokens are in flight (4W rows x < 320 columns, short leading dimension).
            // Widths that are no multiple of 256 - and the EXPERIMENTAL build's transposing A/B arm - transpose every linear:
            // 4W rows x all tokens (0.57 GB for ViT-L/14 at B = 128).
            h->Mpt = (long)Mp + 16 * 128;
            bool every_linear = W % 256 != 0;
#ifdef RVLM_EXPERIMENTAL_GEMM
            every_linear = true;
#endif
            const size_t rows = (size_t)(every_linear ? std::max(4 * W, h->Kpad) : std::max(W, h->Kpad));
            h->t_elems = std::max(rows * (size_t)h->Mpt, (size_t)std::max(4 * W, h->Kpad) * 320);
            ALLOC_OR_DIE(h->tA, h->t_elems * 2);
            ALLOC_OR_DIE(h->tB, h->t_elems * 2);
        }
    }
    {
        // split-K slabs: 8 x <=256 rows x 4W columns for the few-row GEMMs; trainable handles also run
        // the weight-gradient GEMMs split-K (few output tiles, K = all tokens): up to 3 x [3W, W] slabs
        size_t sk = (size_t)8 * 256 * 4 * W * sizeof(float);
        // (sized from the plan itself - ADVICE r5: 16 W^2 floats cover W = 1024 exactly but not W = 768, whose QKV plan wants 9
        // slabs of [3W, W]; those shapes then fell to the transposing path, whose scratch the copy-free sizing no longer covers)
        if (cfg->trainable > 0)
            sk = std::max({sk, wgrad_slab_bytes(3 * W, W), wgrad_slab_bytes(W, W), wgrad_slab_bytes(4 * W, W), wgrad_slab_bytes(W, 4 * W)});
        ALLOC_OR_DIE(h->splitk_scratch, sk);
        h->splitk_bytes = sk;
        if (cfg->trainable > 0) {
            // partial column sums: [2][RED_NCH][4W] (LayerNorm affine) or one row per 64-token tile of a transpose
            const size_t rf = std::max((size_t)2 * 128 * 4 * W, (size_t)((Mp + 16 * 128) / 64) * 4 * W);
            ALLOC_OR_DIE(h->red_scratch, rf * sizeof(float));
            h->red_floats = rf;
        }
    }
    const size_t npix = (size_t)B * 3 * h->img * h->img;
    for (int i = 0; i < 5; ++i) ALLOC_OR_DIE(h->img_buf[i], npix * 4);
    ALLOC_OR_DIE(h->emb, (size_t)B * D * 4);
    ALLOC_OR_DIE(h->d_emb, (size_t)B * D * 4);
    ALLOC_OR_DIE(h->loss_ps, (size_t)B * 4);
    ALLOC_OR_DIE(h->loss_scalar, 4096 * 4);
    h->loss_scratch_floats = (size_t)B * 1024 + (size_t)D * 1024;   // CE head with up to 1024 classes
    ALLOC_OR_DIE(h->loss_scratch, h->loss_scratch_floats * 4);
    ALLOC_OR_DIE(h->ap_loss_steps, (size_t)1024 * B * 4);
    ALLOC_OR_DIE(h->ap_loss_best, (size_t)B * 4);
    ALLOC_OR_DIE(h->ap_loss_best_lc, (size_t)B * 4);
    ALLOC_OR_DIE(h->ap_reduced_lc, (size_t)B * 4);
    ALLOC_OR_DIE(h->ap_step, (size_t)B * 4);
    ALLOC_OR_DIE(h->ap_acc, B); ALLOC_OR_DIE(h->ap_pred, B);
    ALLOC_OR_DIE(h->ap_f0, B); ALLOC_OR_DIE(h->ap_f1, B); ALLOC_OR_DIE(h->ap_f2, B);
#undef ALLOC_OR_DIE
    hipError_t he = hipStreamSynchronize(s);
    if (he != hipSuccess) { rvlm_vit_destroy(h); return fail(RVLM_ERR_HIP, std::string("create sync: ") + hipGetErrorString(he)); }
    *out = h;
    return RVLM_OK;
}

extern "C" int rvlm_vit_destroy(rvlm_vit* h) {
    if (!h) return RVLM_OK;
    (void)hipDeviceSynchronize();
    for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
    return RVLM_OK;
}

extern "C" int rvlm_vit_load_weights(rvlm_vit* h, const rvlm_vit_weights* weights, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && weights, "rvlm_vit_load_weights: null argument");
    return load_weights(h, weights, (hipStream_t)stream, false);
}

extern "C" size_t rvlm_vit_workspace_bytes(const rvlm_vit* h) { return h ? h->bytes : 0; }

extern "C" int rvlm_vit_forward(rvlm_vit* h, const float* x, const float* delta, int B, int output_normalize,
                                int save_for_backward, float* out_emb, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && x && out_emb, "rvlm_vit_forward: null argument");
    RVLM_REQUIRE(B > 0 && B <= h->maxB, "rvlm_vit_forward: batch exceeds max_batch");
    RVLM_REQUIRE(save_for_backward != 2 || h->trainable, "rvlm_vit_forward: save_for_backward == 2 needs a trainable handle");
    RVLM_REQUIRE(save_for_backward == 0 || !h->inference_only, "rvlm_vit_forward: inference-only handle cannot save activations");
    return vit_forward(h, x, delta, B, output_normalize, save_for_backward, out_emb, (hipStream_t)stream);
}

extern "C" int rvlm_vit_backward_input(rvlm_vit* h, const float* d_emb, int B, float* grad_x,
                                       rvlm_stream_t stream) {
    RVLM_REQUIRE(h && d_emb && grad_x, "rvlm_vit_backward_input: null argument");
    if (h->saved_B != B || B <= 0)
        return fail(RVLM_ERR_STATE, "rvlm_vit_backward_input: no saved forward for this batch size");
    return vit_backward(h, d_emb, B, grad_x, (hipStream_t)stream);
}

extern "C" int rvlm_vit_set_flash_inference(rvlm_vit* h, int on) {
    RVLM_REQUIRE(h, "rvlm_vit_set_flash_inference: null handle");
    h->flash_inference = on != 0 && !h->bf16;
    return RVLM_OK;
}

extern "C" int rvlm_vit_forward_for(rvlm_vit* h, rvlm_vit* consumer, const float* x, const float* delta, int B, int output_normalize,
                                    float* out_emb, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && consumer && x && out_emb, "rvlm_vit_forward_for: null argument");
    RVLM_REQUIRE(B > 0 && B <= h->maxB, "rvlm_vit_forward_for: batch exceeds max_batch");
    RVLM_REQUIRE(!h->bf16 && consumer->bf16 && !h->inference_only && !consumer->inference_only && h->Mp == consumer->Mp &&
                 h->W == consumer->W && h->L == consumer->L && h->S == consumer->S && h->H == consumer->H,
                 "rvlm_vit_forward_for: needs an fp32-storage handle and a bf16 handle of the same architecture and max_batch");
    h->peer = consumer;
    const int rc = vit_forward(h, x, delta, B, output_normalize, 1, out_emb, (hipStream_t)stream);
    h->peer = nullptr;
    return rc;
}

extern "C" int rvlm_vit_backward_input_from(rvlm_vit* h, rvlm_vit* h_saved, const float* d_emb, int B, float* grad_x,
                                            rvlm_stream_t stream) {
    RVLM_REQUIRE(h && h_saved && d_emb && grad_x, "rvlm_vit_backward_input_from: null argument");
    return vit_backward_from(h, h_saved, d_emb, B, grad_x, (hipStream_t)stream);
}

extern "C" int rvlm_vit_backward_params(rvlm_vit* h, const float* d_emb, int B, const rvlm_vit_weights* grads,
                                        int accumulate, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && d_emb && grads && grads->blocks_host, "rvlm_vit_backward_params: null argument");
    if (!h->trainable) return fail(RVLM_ERR_STATE, "rvlm_vit_backward_params: handle was not created trainable");
    if (h->saved_B != B || B <= 0 || h->saved_mode != 2)
        return fail(RVLM_ERR_STATE, "rvlm_vit_backward_params: needs a forward with save_for_backward == 2 for this batch");
    hipStream_t s = (hipStream_t)stream;
    h->next_param_stage = h->L + 2;
    return h->bf16 ? backward_params_impl<bf16_t>(h, d_emb, B, grads, accumulate, 0, h->L + 2, s)
                   : backward_params_impl<float>(h, d_emb, B, grads, accumulate, 0, h->L + 2, s);
}

extern "C" int rvlm_vit_backward_params_stages(rvlm_vit* h, const float* d_emb, int B, const rvlm_vit_weights* grads,
                                               int accumulate, int stage_begin, int stage_end, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && d_emb && grads && grads->blocks_host, "rvlm_vit_backward_params_stages: null argument");
    if (!h->trainable) return fail(RVLM_ERR_STATE, "rvlm_vit_backward_params_stages: handle was not created trainable");
    if (h->saved_B != B || B <= 0 || h->saved_mode != 2)
        return fail(RVLM_ERR_STATE, "rvlm_vit_backward_params_stages: needs a forward with save_for_backward == 2 for this batch");
    RVLM_REQUIRE(stage_begin >= 0 && stage_begin < stage_end && stage_end <= h->L + 2,
                 "rvlm_vit_backward_params_stages: need 0 <= stage_begin < stage_end <= layers + 2");
    // stages of one backward hand the residual gradient to each other through the handle: stage k reads what stage
    // k-1 left.  A range that does not continue where the saved forward's backward stands (a skipped or repeated
    // stage) would read stale gradients and write wrong ones silently - refuse it.  Stage 0 restarts the backward.
    if (stage_begin != 0 && stage_begin != h->next_param_stage)
        return fail(RVLM_ERR_STATE, "rvlm_vit_backward_params_stages: stages of one backward must run in order from 0 "
                                    "(stage_begin does not continue the previous call's stage_end)");
    hipStream_t s = (hipStream_t)stream;
    const int rc = h->bf16 ? backward_params_impl<bf16_t>(h, d_emb, B, grads, accumulate, stage_begin, stage_end, s)
                           : backward_params_impl<float>(h, d_emb, B, grads, accumulate, stage_begin, stage_end, s);
    h->next_param_stage = rc == RVLM_OK ? stage_end : -1;
    return rc;
}

static int loss_step(rvlm_vit* h, const rvlm_loss_spec* ls, int B, int reduction, float* loss_scalar,
                     uint8_t* pred_eq, hipStream_t s) {
    if (ls->loss_kind != RVLM_LOSS_L2) {
        RVLM_REQUIRE(ls->n_classes > 0 && ls->n_classes <= 1024, "loss: n_classes must be in 1..1024");
        RVLM_REQUIRE(ls->targets, "loss: head losses need targets");
    }
    PROF("loss", 0, 0);
    return rvlm_loss_grad(ls->loss_kind, reduction, h->emb, ls->ref, ls->targets, ls->y_target, B, h->D, ls->n_classes,
                          ls->logit_scale, h->loss_ps, loss_scalar, h->d_emb, pred_eq, h->loss_scratch, s);
}

// one call = forward (activations kept) + FARE / TeCoA loss + input gradient, i.e. what one
// iteration of pgd_train.py:33-38 asks of the model (out = forward(x + delta); loss = loss_fn(out, targets);
// grad = autograd.grad(loss, delta)).  Every output pointer is optional.
extern "C" int rvlm_vit_fwd_inputgrad(rvlm_vit* h, const float* x, const float* delta, int B, const rvlm_loss_spec* loss,
                                      float* out_emb, float* out_loss_per_sample, float* out_loss_scalar,
                                      float* out_grad_x, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && x && loss && loss->ref, "rvlm_vit_fwd_inputgrad: null argument");
    RVLM_REQUIRE(B > 1 && B <= h->maxB, "rvlm_vit_fwd_inputgrad: need 1 < B <= max_batch");
    RVLM_REQUIRE(!h->inference_only, "rvlm_vit_fwd_inputgrad: inference-only handle");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = vit_forward(h, x, delta, B, loss->output_normalize, 1, h->emb, s))) return rc;
    if ((rc = loss_step(h, loss, B, loss->reduction, out_loss_scalar ? h->loss_scalar : nullptr, nullptr, s))) return rc;
    const size_t eb = (size_t)B * h->D * 4;
    if (out_emb) RVLM_HIP(hipMemcpyAsync(out_emb, h->emb, eb, hipMemcpyDeviceToDevice, s));
    if (out_loss_per_sample) RVLM_HIP(hipMemcpyAsync(out_loss_per_sample, h->loss_ps, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    if (out_loss_scalar) RVLM_HIP(hipMemcpyAsync(out_loss_scalar, h->loss_scalar, 4, hipMemcpyDeviceToDevice, s));
    if (out_grad_x && (rc = vit_backward(h, h->d_emb, B, out_grad_x, s))) return rc;
    return RVLM_OK;
}

// One perturbation loop; iterations [0, n_first) evaluate model and gradient on `hf` (a second handle of the same model, e.g.
// its fp32 mode), the others on `h`.  The attack state (delta, velocity, gradient) lives in h's buffers throughout; a
// handle only contributes forward + loss + input gradient (what pgd_train.py:33-38 asks of the model).
// handoff: the first iterations run forward + loss on `hf` and the input gradient on h's (bf16) kernels from hf's saved forward
// (vit_backward_from above).
static int pgd_run_impl(rvlm_vit* h, rvlm_vit* hf, int n_first, const float* x, const float* delta0, int B,
                        const rvlm_loss_spec* loss, int norm_kind, float eps, int iterations, float stepsize,
                        float momentum, int mode_max, float* x_adv_out, float* loss_trace, int32_t* flags, hipStream_t s,
                        bool handoff = false) {
    const size_t n = (size_t)B * 3 * h->img * h->img;
    float *delta = h->img_buf[0], *vel = h->img_buf[1], *grad = h->img_buf[2];
    int rc;
    if (flags && (rc = rvlm_check_image_range(x, n, flags, s))) return rc;
    if (delta0) RVLM_HIP(hipMemcpyAsync(delta, delta0, n * 4, hipMemcpyDeviceToDevice, s));
    else RVLM_HIP(hipMemsetAsync(delta, 0, n * 4, s));
    RVLM_HIP(hipMemsetAsync(vel, 0, n * 4, s));
    if (iterations == 0) {
        hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, (const float*)delta, x_adv_out, n);
        RVLM_CHECK_LAUNCH();
    }
    for (int it = 0; it < iterations; ++it) {
        rvlm_vit* m = (hf && it < n_first) ? hf : h;
        if (handoff && m != h) m->peer = h;      // (the forward then writes h's bf16 tensors itself and keeps no probabilities)
        rc = vit_forward(m, x, delta, B, loss->output_normalize, 1, m->emb, s);
        m->peer = nullptr;
        if (rc) return rc;
        float* lsc = loss_trace ? loss_trace + it : (it < 4096 ? h->loss_scalar + it : nullptr);
        if ((rc = loss_step(m, loss, B, loss->reduction, lsc, nullptr, s))) return rc;
        if ((rc = (handoff && m != h) ? vit_backward_from(h, m, m->d_emb, B, grad, s) : vit_backward(m, m->d_emb, B, grad, s))) return rc;
        {
            PROF("linf_update", 0, (double)n * 28);
            float* xo = it == iterations - 1 ? x_adv_out : nullptr;
            if (norm_kind == 2) rc = rvlm_pgd_l2_update(x, grad, delta, vel, n / B, B, eps, stepsize, momentum, mode_max, xo, flags, s);
            else rc = rvlm_pgd_linf_update(x, grad, delta, vel, n, eps, stepsize, momentum, mode_max, xo, flags, s);
            if (rc) return rc;
        }
    }
    return RVLM_OK;
}

extern "C" int rvlm_pgd_run_norm(rvlm_vit* h, const float* x, const float* delta0, int B,
                                 const rvlm_loss_spec* loss, int norm_kind, float eps, int iterations, float stepsize,
                                 float momentum, int mode_max, float* x_adv_out, float* loss_trace,
                                 int32_t* flags, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && x && loss && x_adv_out && loss->ref, "rvlm_pgd_run: null argument");
    if (norm_kind != 0 && norm_kind != 2) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_pgd_run: norm must be L-inf (0) or L2 (2)");
    RVLM_REQUIRE(B > 1 && B <= h->maxB, "rvlm_pgd_run: need 1 < B <= max_batch");
    RVLM_REQUIRE(iterations >= 0 && iterations <= 4096, "rvlm_pgd_run: iterations");
    return pgd_run_impl(h, nullptr, 0, x, delta0, B, loss, norm_kind, eps, iterations, stepsize, momentum, mode_max,
                        x_adv_out, loss_trace, flags, (hipStream_t)stream);
}

// Mixed-precision loop (precision "bf16+fp32-first" of the Python mirror): the first n_first iterations on h_first - a
// handle of the SAME model in the reference's own precision (fp32: train/pgd_train.py:30-38 runs no autocast) - and the
// rest on h.  FARE's first cotangent 2 (phi(x + d0) - phi(x)) is a difference of two nearly equal embeddings (~1e-2 of
// their norm at the random start), one part bf16 rounding noise in three; one fp32
// iteration - with loss->ref = the fp32 embedding of x - gives the reference's first step, after which the difference has
// grown tenfold and bf16 agrees to 0.98-0.996.
extern "C" int rvlm_pgd_run_mixed(rvlm_vit* h, rvlm_vit* h_first, int n_first, const float* x, const float* delta0, int B,
                                  const rvlm_loss_spec* loss, int norm_kind, float eps, int iterations, float stepsize,
                                  float momentum, int mode_max, float* x_adv_out, float* loss_trace,
                                  int32_t* flags, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && h_first && x && loss && x_adv_out && loss->ref, "rvlm_pgd_run_mixed: null argument");
    if (norm_kind != 0 && norm_kind != 2) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_pgd_run_mixed: norm must be L-inf (0) or L2 (2)");
    RVLM_REQUIRE(B > 1 && B <= h->maxB && B <= h_first->maxB, "rvlm_pgd_run_mixed: need 1 < B <= max_batch of both handles");
    RVLM_REQUIRE(iterations >= 0 && iterations <= 4096 && n_first >= 0, "rvlm_pgd_run_mixed: iterations");
    RVLM_REQUIRE(h->img == h_first->img && h->P == h_first->P && h->W == h_first->W && h->L == h_first->L &&
                 h->H == h_first->H && h->D == h_first->D && h->cfg.act == h_first->cfg.act,
                 "rvlm_pgd_run_mixed: the two handles must hold the same architecture");
    RVLM_REQUIRE(!h->inference_only && !h_first->inference_only, "rvlm_pgd_run_mixed: inference-only handle");
    return pgd_run_impl(h, h_first, n_first, x, delta0, B, loss, norm_kind, eps, iterations, stepsize, momentum, mode_max,
                        x_adv_out, loss_trace, flags, (hipStream_t)stream);
}

// ... with the handoff (round 6): the first n_first iterations' FORWARD + loss on h_first (fp32 storage: x3 or fp32 precision), their
// input gradient on h's bf16 backward from h_first's saved forward - FARE's first step needs a faithful embedding difference, not a
// faithful backward.  Both handles: same architecture AND the same max_batch.
extern "C" int rvlm_pgd_run_mixed_fwd(rvlm_vit* h, rvlm_vit* h_first, int n_first, const float* x, const float* delta0, int B,
                                      const rvlm_loss_spec* loss, int norm_kind, float eps, int iterations, float stepsize,
                                      float momentum, int mode_max, float* x_adv_out, float* loss_trace,
                                      int32_t* flags, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && h_first && x && loss && x_adv_out && loss->ref, "rvlm_pgd_run_mixed_fwd: null argument");
    if (norm_kind != 0 && norm_kind != 2) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_pgd_run_mixed_fwd: norm must be L-inf (0) or L2 (2)");
    RVLM_REQUIRE(B > 1 && B <= h->maxB && B <= h_first->maxB, "rvlm_pgd_run_mixed_fwd: need 1 < B <= max_batch of both handles");
    RVLM_REQUIRE(iterations >= 0 && iterations <= 4096 && n_first >= 0, "rvlm_pgd_run_mixed_fwd: iterations");
    RVLM_REQUIRE(h->bf16 && !h_first->bf16, "rvlm_pgd_run_mixed_fwd: h must be a bf16 handle, h_first an fp32-storage (fp32 / x3) one");
    RVLM_REQUIRE(h->Mp == h_first->Mp && h->Mp0 == h_first->Mp0, "rvlm_pgd_run_mixed_fwd: the two handles need the same max_batch");
    RVLM_REQUIRE(!h->inference_only && !h_first->inference_only, "rvlm_pgd_run_mixed_fwd: inference-only handle");
    return pgd_run_impl(h, h_first, n_first, x, delta0, B, loss, norm_kind, eps, iterations, stepsize, momentum, mode_max,
                        x_adv_out, loss_trace, flags, (hipStream_t)stream, true);
}

extern "C" int rvlm_pgd_run(rvlm_vit* h, const float* x, const float* delta0, int B,
                            const rvlm_loss_spec* loss, float eps, int iterations, float stepsize,
                            float momentum, int mode_max, float* x_adv_out, float* loss_trace,
                            int32_t* flags, rvlm_stream_t stream) {
    return rvlm_pgd_run_norm(h, x, delta0, B, loss, 0, eps, iterations, stepsize, momentum, mode_max, x_adv_out,
                             loss_trace, flags, stream);
}

extern "C" int rvlm_apgd_run_norm(rvlm_vit* h, const float* x, const float* x_init, int B,
                                  const rvlm_loss_spec* loss, int norm_kind, float eps, int n_iter, float step0,
                                  int train_variant, int logits_from_head, float* x_best_adv, float* x_best_out,
                                  float* loss_best_out, uint8_t* acc_out, rvlm_stream_t stream) {
    RVLM_REQUIRE(h && x && loss && loss->ref && loss->targets, "rvlm_apgd_run: null argument");
    if (norm_kind != 0 && norm_kind != 2) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_apgd_run: norm must be L-inf (0) or L2 (2)");
    RVLM_REQUIRE(x_best_adv, "rvlm_apgd_run: x_best_adv output required");
    RVLM_REQUIRE(B > 1 && B <= h->maxB, "rvlm_apgd_run: need 1 < B <= max_batch");
    RVLM_REQUIRE(n_iter >= 1 && n_iter <= 1024, "rvlm_apgd_run: n_iter must be in 1..1024");
    RVLM_REQUIRE(!logits_from_head || loss->loss_kind != RVLM_LOSS_L2, "rvlm_apgd_run: head logits need a head loss (ce / dlr)");
    RVLM_REQUIRE(loss->loss_kind == RVLM_LOSS_L2 || loss->loss_kind == RVLM_LOSS_CE || logits_from_head,
                 "rvlm_apgd_run: the DLR losses run on the classification head (logits_from_head = 1)");
    hipStream_t s = (hipStream_t)stream;
    const size_t npix = (size_t)3 * h->img * h->img, n = npix * B;
    float *x_adv = h->img_buf[0], *x_adv_old = h->img_buf[1], *x_best = h->img_buf[2], *grad = h->img_buf[3],
          *grad_best = h->img_buf[4];
    int rc;
    // schedule (apgd_train.py:153-156)
    int k = std::max((int)(0.22 * n_iter), 1);
    const int n_iter_min = std::max((int)(0.06 * n_iter), 1), size_decr = std::max((int)(0.03 * n_iter), 1);
    int counter3 = 0;
    hipLaunchKernelGGL(clamp01_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x_init ? x_init : x, x_adv, n);
    RVLM_CHECK_LAUNCH();
    RVLM_HIP(hipMemcpyAsync(x_best, x_adv, n * 4, hipMemcpyDeviceToDevice, s));
    RVLM_HIP(hipMemcpyAsync(x_best_adv, x_adv, n * 4, hipMemcpyDeviceToDevice, s));
    RVLM_HIP(hipMemcpyAsync(x_adv_old, x_adv, n * 4, hipMemcpyDeviceToDevice, s));
    RVLM_HIP(hipMemsetAsync(h->ap_loss_steps, 0, (size_t)n_iter * B * 4, s));

    auto eval = [&](bool need_grad) -> int {
        int r;
        if ((r = vit_forward(h, x_adv, nullptr, B, loss->output_normalize, need_grad ? 1 : 0, h->emb, s))) return r;
        if ((r = loss_step(h, loss, B, RVLM_RED_NONE, nullptr, logits_from_head ? h->ap_pred : nullptr, s))) return r;
        if (!logits_from_head) {   // apgd_train.py:192,301: argmax over the model output (the embedding)
            if ((r = rvlm_argmax_eq(h->emb, loss->targets, B, h->D, h->ap_pred, s))) return r;
        }
        if (need_grad) { if ((r = vit_backward(h, h->d_emb, B, grad, s))) return r; }
        return RVLM_OK;
    };
    if ((rc = eval(true))) return rc;
    RVLM_HIP(hipMemcpyAsync(grad_best, grad, n * 4, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(init_apgd_state_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, B, h->loss_ps, h->ap_pred, step0,
                       h->ap_loss_best, h->ap_loss_best_lc, h->ap_reduced_lc, h->ap_step, h->ap_acc);
    RVLM_CHECK_LAUNCH();
    for (int i = 0; i < n_iter; ++i) {
        const float a = i > 0 ? 0.75f : 1.0f;
        {
            PROF("linf_update", 0, (double)n * 24);
            rc = norm_kind == 2 ? rvlm_apgd_l2_step(x, x_adv, x_adv_old, grad, h->ap_step, a, eps, npix, B, s)
                                : rvlm_apgd_linf_step(x, x_adv, x_adv_old, grad, h->ap_step, a, eps, npix, B, s);
            if (rc) return rc;
        }
        const bool need_grad = !(train_variant && i == n_iter - 1);   // apgd_train.py:293-295
        if ((rc = eval(need_grad))) return rc;
        counter3 += 1;
        const int do_check = counter3 == k;
        // apgd_train hard-codes the threshold (train/apgd_train.py:117,334: k3 = 0.75); `rho` is APGDAttack's parameter only, so
        // a value left on the handle by an APGDAttack(rho != 0.75) run does not reach a later apgd_train on the same handle
        if ((rc = rvlm_apgd_controller_rho(i, B, n_iter, k, do_check, train_variant ? 0.75 : h->ap_rho, h->loss_ps, h->ap_pred, h->ap_loss_steps,
                                       h->ap_loss_best, h->ap_loss_best_lc, h->ap_reduced_lc, h->ap_step,
                                       h->ap_acc, h->ap_f0, h->ap_f1, h->ap_f2, s))) return rc;
        {
            PROF("linf_update", 0, (double)n * 8);
            if ((rc = rvlm_apgd_select(x_adv, grad, x_best, grad_best, x_best_adv, h->ap_f0, h->ap_f1, h->ap_f2,
                                       npix, B, s))) return rc;
        }
        if (do_check) { counter3 = 0; k = std::max(k - size_decr, n_iter_min); }
    }
    if (x_best_out) RVLM_HIP(hipMemcpyAsync(x_best_out, x_best, n * 4, hipMemcpyDeviceToDevice, s));
    if (loss_best_out) RVLM_HIP(hipMemcpyAsync(loss_best_out, h->ap_loss_best, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    if (acc_out) RVLM_HIP(hipMemcpyAsync(acc_out, h->ap_acc, (size_t)B, hipMemcpyDeviceToDevice, s));
    return RVLM_OK;
}

extern "C" int rvlm_vit_set_apgd_rho(rvlm_vit* h, double rho) {
    RVLM_REQUIRE(h && rho == rho, "rvlm_vit_set_apgd_rho: null handle or NaN");
    h->ap_rho = rho;
    return RVLM_OK;
}

extern "C" int rvlm_apgd_run(rvlm_vit* h, const float* x, const float* x_init, int B,
                             const rvlm_loss_spec* loss, float eps, int n_iter, float step0,
                             int train_variant, int logits_from_head, float* x_best_adv, float* x_best_out,
                             float* loss_best_out, uint8_t* acc_out, rvlm_stream_t stream) {
    return rvlm_apgd_run_norm(h, x, x_init, B, loss, 0, eps, n_iter, step0, train_variant, logits_from_head, x_best_adv,
                              x_best_out, loss_best_out, acc_out, stream);
}

extern "C" int rvlm_vit_set_profiling(rvlm_vit* h, int enabled) {
    RVLM_REQUIRE(h, "rvlm_vit_set_profiling: null handle");
    h->prof = enabled != 0;
    return RVLM_OK;
}
extern "C" int rvlm_vit_reset_profile(rvlm_vit* h) {
    RVLM_REQUIRE(h, "rvlm_vit_reset_profile: null handle");
    (void)hipDeviceSynchronize();
    for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    h->recs.clear();
    for (auto& a : h->accs) a = ProfAcc();
    return RVLM_OK;
}
extern "C" int rvlm_vit_get_profile(rvlm_vit* h, rvlm_profile_entry* out, int* n) {
    RVLM_REQUIRE(h && out && n, "rvlm_vit_get_profile: null argument");
    RVLM_HIP(hipDeviceSynchronize());
    for (auto& a : h->accs) a.ms = 0;
    for (auto& r : h->recs) {
        float ms = 0.0f;
        hipError_t e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e == hipSuccess) h->accs[r.cls].ms += ms;
    }
    int cnt = 0;
    for (size_t i = 0; i < h->cls_names.size() && cnt < *n; ++i) {
        if (h->accs[i].n == 0) continue;
        rvlm_profile_entry& e = out[cnt++];
        memset(&e, 0, sizeof(e));
        strncpy(e.name, h->cls_names[i].c_str(), sizeof(e.name) - 1);
        e.total_ms = h->accs[i].ms; e.flops = h->accs[i].flops; e.bytes = h->accs[i].bytes; e.launches = h->accs[i].n;
    }
    *n = cnt;
    return RVLM_OK;
}
