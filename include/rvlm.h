/* rvlm.h - C ABI of librvlm.so, the MI355X (gfx950) adversarial inner loop for RobustVLM.
 *
 * Drop-in boundary.  Everything the reference's Python hot path does on
 * the device goes through these entry points; the Python shim in robustvlm_amd/ binds them with
 * ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host.
 *   - every function takes the hipStream_t to enqueue on (as void*), never synchronises the device
 *     (except *_create / *_destroy / *_get_profile), starts no threads and allocates nothing after
 *     rvlm_vit_create.
 *   - return value: 0 = RVLM_OK, otherwise an rvlm_status; rvlm_last_error() gives the message of
 *     the last failure on the calling thread.  No exceptions cross the ABI.
 *   - images are NCHW fp32 in [0,1] (un-normalised: the CLIP mean/std Normalize lives inside the
 *     encoder, train/adversarial_training_clip.py:105-108,254), targets are int64.
 */
#ifndef RVLM_H
#define RVLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVLM_VERSION 109

typedef void* rvlm_stream_t; /* hipStream_t */
typedef struct rvlm_vit rvlm_vit;

typedef enum {
    RVLM_OK = 0,
    RVLM_ERR_ARG = 1,         /* bad argument (null pointer, shape out of range, unknown enum) */
    RVLM_ERR_HIP = 2,         /* a HIP runtime call failed */
    RVLM_ERR_STATE = 3,       /* call order violated (e.g. backward without a saved forward) */
    RVLM_ERR_UNSUPPORTED = 4  /* configuration the kernels do not cover */
} rvlm_status;

/* GEMM / attention operand precision.  RVLM_PREC_F32X3 (ABI 108): fp32 storage and fp32 LayerNorm / softmax / attention products
 * like RVLM_PREC_F32, the encoder's linears as split-bf16 products (a_hi w_hi + a_hi w_lo + a_lo w_hi, fp32 accumulate) on the
 * bf16 matrix pipe: ~16 mantissa bits at 3 bf16 MFMAs per product (robustvlm_amd/csrc/x3_kernels.hip); attack / inference only */
enum { RVLM_PREC_F32 = 0, RVLM_PREC_BF16 = 1, RVLM_PREC_F32X3 = 2 };
enum { RVLM_ACT_QUICK_GELU = 0, RVLM_ACT_GELU = 1 };  /* open_clip QuickGELU / exact erf GELU */
enum { RVLM_LOSS_L2 = 0, RVLM_LOSS_CE = 1,             /* FARE l2 / TeCoA ce (…clip.py:495-528) */
       RVLM_LOSS_DLR = 2, RVLM_LOSS_DLR_TARGETED = 3 }; /* AutoAttack DLR losses (autopgd_base.py:195-201, 613-618) */
enum { RVLM_RED_MEAN = 0, RVLM_RED_NONE = 1 };        /* reduction='mean' | 'none' (grad of sum) */

/* bits of the device-side flag word written by the attack kernels (replaces the reference's
 * per-iteration host asserts, train/pgd_train.py:24,40,60-63) */
enum {
    RVLM_FLAG_INPUT_RANGE = 1, /* data_clean outside [0,1] (+-1e-6)          pgd_train.py:24    */
    RVLM_FLAG_NAN_GRAD = 2,    /* NaN in the gradient (zeroed, not an error) pgd_train.py:40-42 */
    RVLM_FLAG_NAN_DELTA = 4,   /* NaN in the perturbation                    pgd_train.py:60    */
    RVLM_FLAG_ADV_RANGE = 8    /* x+delta left [0,1] (+-1e-6)                pgd_train.py:61-63 */
};

/* ---------------------------------------------------------------------------------------------
 * Encoder: open_clip VisionTransformer forward + input-gradient.
 * Replaces `ClipVisionModel.forward` (train/adversarial_training_clip.py:253-257) and the autograd
 * backward `torch.autograd.grad(loss, perturbation)` (train/pgd_train.py:38, apgd_train.py:185,295).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t image_size; /* 224 */
    int32_t patch;      /* 14 (ViT-L/14), 32 (ViT-B/32) */
    int32_t width;      /* 1024 / 768; heads * 64 */
    int32_t layers;     /* 24 / 12 */
    int32_t heads;      /* 16 / 12 (head_dim must be 64) */
    int32_t out_dim;    /* 768 / 512 */
    int32_t act;        /* RVLM_ACT_* */
    int32_t precision;  /* RVLM_PREC_* */
    int32_t max_batch;  /* workspace is sized for this many images */
    float mean[3];      /* Normalize constants, …clip.py:116 */
    float std[3];
    int32_t trainable;  /* > 0: also keep what the weight-gradient backward needs (rvlm_vit_backward_params);
                           -1: inference only (frozen model_orig copy): no per-layer activation storage;
                           -2 (ABI 109, fp32-storage precisions): forward PROVIDER of a bf16 handle - its saving forwards are all
                           rvlm_vit_forward_for / rvlm_pgd_run_mixed_fwd: per-block residual stream and statistics only, no
                           backward scratch, no probabilities (51 -> 13 GiB at ViT-L/14, B = 128); on a bf16 handle -2 means -1 */
} rvlm_vit_config;

/* fp32 device pointers in `visual.state_dict()` layout (…clip.py:239,470; Appendix B key list). */
typedef struct {
    const float* ln_1_weight;          /* [W]      */
    const float* ln_1_bias;            /* [W]      */
    const float* attn_in_proj_weight;  /* [3W, W]  */
    const float* attn_in_proj_bias;    /* [3W]     */
    const float* attn_out_proj_weight; /* [W, W]   */
    const float* attn_out_proj_bias;   /* [W]      */
    const float* ln_2_weight;          /* [W]      */
    const float* ln_2_bias;            /* [W]      */
    const float* mlp_c_fc_weight;      /* [4W, W]  */
    const float* mlp_c_fc_bias;        /* [4W]     */
    const float* mlp_c_proj_weight;    /* [W, 4W]  */
    const float* mlp_c_proj_bias;      /* [W]      */
} rvlm_vit_block_weights;

typedef struct {
    const float* class_embedding;      /* [W]            */
    const float* positional_embedding; /* [tokens, W]    */
    const float* proj;                 /* [W, out_dim]   */
    const float* conv1_weight;         /* [W, 3, P, P]   */
    const float* ln_pre_weight;        /* [W] */
    const float* ln_pre_bias;
    const float* ln_post_weight;
    const float* ln_post_bias;
    const rvlm_vit_block_weights* blocks_host; /* HOST array of `layers` structs of device ptrs */
} rvlm_vit_weights;

/* Allocates the workspace, converts/transposes the weights into the engine's layout. */
int rvlm_vit_create(const rvlm_vit_config* cfg, const rvlm_vit_weights* weights,
                    rvlm_stream_t stream, rvlm_vit** out);
int rvlm_vit_destroy(rvlm_vit* h);
/* Re-import weights (after an optimizer step of the outer trainer). */
int rvlm_vit_load_weights(rvlm_vit* h, const rvlm_vit_weights* weights, rvlm_stream_t stream);
/* Bytes of device memory the handle owns. */
size_t rvlm_vit_workspace_bytes(const rvlm_vit* h);

/* emb[B,out_dim] = visual(Normalize(x + delta)) (delta may be NULL), L2-normalised iff
 * output_normalize.  save_for_backward: 0 = inference, 1 = keep what rvlm_vit_backward_input needs,
 * 2 = additionally keep the linear-layer inputs for rvlm_vit_backward_params (trainable handles). */
int rvlm_vit_forward(rvlm_vit* h, const float* x, const float* delta, int B, int output_normalize,
                     int save_for_backward, float* out_emb, rvlm_stream_t stream);
/* grad_x[B,3,H,W] = d<d_emb, emb>/d(x+delta) for the last saved forward (dgrad only, no wgrad). */
int rvlm_vit_backward_input(rvlm_vit* h, const float* d_emb, int B, float* grad_x,
                            rvlm_stream_t stream);
/* The same for the forward saved on ANOTHER handle of the same model and max_batch (ABI 109): h_saved holds fp32 storage
 * (RVLM_PREC_F32 / RVLM_PREC_F32X3), h is a bf16 handle whose backward kernels evaluate the gradient - qkv, attention output and
 * act'(fc1) are exported to bf16, the fp32 residual stream / LayerNorm statistics are read in place.  For the first FARE iteration
 * (torch.autograd.grad(loss, perturbation), train/pgd_train.py:38): the forward difference needs the precision, the backward does
 * not.  Invalidates h's own saved forward. */
int rvlm_vit_backward_input_from(rvlm_vit* h, rvlm_vit* h_saved, const float* d_emb, int B, float* grad_x,
                                 rvlm_stream_t stream);
/* A saving forward on the fp32-storage handle h FOR `consumer`'s backward (ABI 109): like rvlm_vit_forward(save_for_backward = 1), but
 * the attention runs as the fp32 flash kernel (no probabilities kept - rvlm_vit_backward_input on h is refused for this pass) and the
 * bf16 tensors rvlm_vit_backward_input_from(consumer, h, ...) reads are written by the forward itself instead of by export passes. */
int rvlm_vit_forward_for(rvlm_vit* h, rvlm_vit* consumer, const float* x, const float* delta, int B, int output_normalize,
                         float* out_emb, rvlm_stream_t stream);
/* Non-saving forwards of an fp32-storage handle on the fp32 flash attention as well (default off: a handle's saving and non-saving
 * forwards agree bit for bit, which FARE's zero gradient at delta = 0 relies on; switch it on for the handle whose saving forwards
 * are all rvlm_vit_forward_for, so that its clean embedding and its first-iteration embedding share one arithmetic).  ABI 109. */
int rvlm_vit_set_flash_inference(rvlm_vit* h, int on);

/* Weight gradients of the outer training step (loss_total.backward(), …clip.py:361): for the last
 * forward run with save_for_backward == 2 on a `trainable` handle, writes (accumulate == 0) or adds
 * (accumulate != 0) d<d_emb, emb>/d(param) for EVERY parameter into the fp32 tensors `grads` points to
 * (same struct / shapes as the weights).  No input gradient is produced. */
int rvlm_vit_backward_params(rvlm_vit* h, const float* d_emb, int B, const rvlm_vit_weights* grads,
                             int accumulate, rvlm_stream_t stream);
/* The same pass cut into stages (execution order): 0 = head (proj, ln_post), 1 + j = transformer block layers-1-j,
 * layers + 1 = embeddings (ln_pre, positional / class embedding, conv1).  Runs stages [stage_begin, stage_end); the
 * stages of one backward must be run in order.  The host uses it to all-reduce a finished gradient bucket (RCCL,
 * side stream) while the remaining stages compute - the reference's DataParallel reduces inside backward as well
 * (…clip.py:184-191). */
int rvlm_vit_backward_params_stages(rvlm_vit* h, const float* d_emb, int B, const rvlm_vit_weights* grads,
                                    int accumulate, int stage_begin, int stage_end, rvlm_stream_t stream);
/* torch.optim.AdamW single-tensor step on flat fp32 buffers (optimizer.step(), …clip.py:196-197,362):
 * grads are multiplied by grad_scale first (1/world_size after a sum all-reduce). `step` counts from 1.  The
 * hyper-parameters are doubles (Python floats in torch): bias corrections and step size are formed in double and
 * rounded to fp32 where torch rounds them. */
int rvlm_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                    float grad_scale, rvlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Losses (replace compute_loss / l2 / ce, …clip.py:495-528) - loss value and d loss / d emb.
 *   L2:  ref = embedding_orig [B,D];  per_sample = sum_d (emb-ref)^2
 *   CE:  ref = text head T [D,C] (column-normalised); logits = emb @ (logit_scale*T)
 * reduction MEAN: *loss_scalar = mean_b, d_emb carries the 1/B; NONE: d_emb = grad of the sum.
 *   DLR / DLR_TARGETED (APGD-DLR, APGD-T): same logits; per_sample =
 *        -(z_y - max_{c != y} z_c) / (z_(1) - z_(3) + 1e-12)            resp.
 *        -(z_y - z_t) / (z_(1) - (z_(3) + z_(4)) / 2 + 1e-12),  t = y_target[b],  z_(k) = k-th largest logit
 * reduction MEAN: *loss_scalar = mean_b, d_emb carries the 1/B; NONE: d_emb = grad of the sum.
 * pred_eq (optional, u8[B]) = argmax_c(logits) == targets (head losses only).  scratch: the head losses need
 * B*C + D*C floats (may be NULL for L2).  y_target: DLR_TARGETED only (else NULL).
 * ------------------------------------------------------------------------------------------- */
int rvlm_loss_grad(int loss_kind, int reduction, const float* emb, const float* ref,
                   const int64_t* targets, const int64_t* y_target, int B, int D, int C, float logit_scale,
                   float* loss_per_sample, float* loss_scalar, float* d_emb, uint8_t* pred_eq,
                   float* scratch, rvlm_stream_t stream);
/* ce() on given logits (…clip.py:523-528 = F.cross_entropy): per-sample loss, optional scalar (mean or sum per
 * `reduction`), d_logits [B,C] = d loss / d logits (MEAN carries the 1/B; may alias logits), optional pred_eq. */
int rvlm_ce_logits(const float* logits, const int64_t* targets, int B, int C, int reduction,
                   float* loss_per_sample, float* loss_scalar, float* d_logits, uint8_t* pred_eq,
                   rvlm_stream_t stream);
/* Zero-shot head of ClassificationModel.forward (CLIP_eval/clip_robustbench.py:66-68): logits[B,C] = (emb[B,D] @
 * T[D,C]) * scale, and its backward d_emb = (d_logits * scale) @ T^T. */
int rvlm_head_logits(const float* emb, const float* T, int B, int D, int C, float scale, float* logits,
                     rvlm_stream_t stream);
int rvlm_head_logits_bwd(const float* d_logits, const float* T, int B, int D, int C, float scale, float* d_emb,
                         rvlm_stream_t stream);
/* Logging metrics of the training step (…clip.py:368-387): out_per_row[b] = F.cosine_similarity(a[b], b[b]) (eps 1e-8),
 * optional *out_mean = their mean; F.normalize(e, dim=1) (eps 1e-12) with the reciprocal norms. */
int rvlm_cosine_rows(const float* a, const float* b, int B, int D, float* out_per_row, float* out_mean,
                     rvlm_stream_t stream);
int rvlm_l2_normalize_rows(const float* e, int B, int D, float* out, float* inv_norm, rvlm_stream_t stream);
/* out[b] = (argmax_j logits[b,j] == targets[b]); ties -> first index (…clip.py:490,
 * apgd_train.py:192,301). */
int rvlm_argmax_eq(const float* logits, const int64_t* targets, int B, int C, uint8_t* out,
                   rvlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * L-inf attack arithmetic (bit-exact with the reference).
 * ------------------------------------------------------------------------------------------- */
/* flags |= RVLM_FLAG_INPUT_RANGE if any x outside (-1e-6, 1+1e-6)   (pgd_train.py:24) */
int rvlm_check_image_range(const float* x, size_t n, int32_t* flags, rvlm_stream_t stream);
/* The standalone helpers of vlm_eval/attacks/utils.py as they are (robustvlm_amd/attack_utils.py):
 *   project_perturbation (:8-16): norm_kind 0 -> clamp(pert, -eps, eps); 2 -> torch.renorm(pert, p=2, dim=0, maxnorm=eps)
 *   normalize_grad (:19-26):      norm_kind 0 -> sign(grad);             2 -> F.normalize(grad.view(B, -1), p=2, dim=1)
 * over B samples of n_per_sample elements; out may alias the input. */
int rvlm_project_perturbation(const float* pert, size_t n_per_sample, int B, int norm_kind, float eps, float* out,
                              rvlm_stream_t stream);
int rvlm_normalize_grad(const float* grad, size_t n_per_sample, int B, int norm_kind, float* out, rvlm_stream_t stream);
/* One PGD step (pgd_train.py:38-63 + vlm_eval/attacks/utils.py:10,21), in place on delta/velocity:
 *   g=NaN->0; v=sign(mom*v+sign(g)); delta=clamp(delta +/- step*v, +-eps);
 *   delta=clamp(x+delta,0,1)-x.   x_adv_out (optional) = x + delta. */
int rvlm_pgd_linf_update(const float* x, const float* grad, float* delta, float* velocity,
                         size_t n, float eps, float stepsize, float momentum, int mode_max,
                         float* x_adv_out, int32_t* flags, rvlm_stream_t stream);
/* The L2 branch of the same block (norm in [2, 'l2', ...]; vlm_eval/attacks/utils.py:12-14,22-26), per sample b over
 * its n_per_sample pixels:  g = NaN->0; g /= max(|g|_2, 1e-12); v = mom*v + g; v /= max(|v|_2, 1e-12);
 * delta +/-= step*v; delta *= eps/(|delta|_2 + 1e-7) where |delta|_2 > eps (torch.renorm); delta = clamp(x+delta,0,1)-x.
 * The norms are deterministic fp32 sums in an order of their own: equal to the reference to fp32 rounding. */
int rvlm_pgd_l2_update(const float* x, const float* grad, float* delta, float* velocity, size_t n_per_sample, int B,
                       float eps, float stepsize, float momentum, int mode_max, float* x_adv_out, int32_t* flags,
                       rvlm_stream_t stream);
/* One APGD step (apgd_train.py:205-229 == autopgd_base.py:328-341); step is per-sample [B]. */
int rvlm_apgd_linf_step(const float* x, float* x_adv, float* x_adv_old, const float* grad,
                        const float* step, float a, float eps, size_t n_per_sample, int B,
                        rvlm_stream_t stream);
/* The L2 branch of the same step (apgd_train.py:231-254): gradient step of length `step` along grad / |grad|_2, projection
 * onto the eps ball (L2) around x and the image range, momentum mix (a, 1 - a), projection again; per-sample norms are
 * deterministic fp32 sums in an order of their own (equal to the reference to fp32 rounding). */
int rvlm_apgd_l2_step(const float* x, float* x_adv, float* x_adv_old, const float* grad, const float* step, float a,
                      float eps, size_t n_per_sample, int B, rvlm_stream_t stream);
/* Per-sample APGD bookkeeping for iteration i (apgd_train.py:301-305,320-355): k and do_check
 * follow the data-independent checkpoint schedule kept by the host. */
int rvlm_apgd_controller(int i, int B, int n_iter, int k, int do_check, const float* loss_i,
                         const uint8_t* pred, float* loss_steps, float* loss_best,
                         float* loss_best_last_check, float* reduced_last_check, float* step,
                         uint8_t* acc, uint8_t* f_notpred, uint8_t* f_improved,
                         uint8_t* f_reduced, rvlm_stream_t stream);
/* The same with the oscillation threshold as a parameter: `rho` of APGDAttack (autoattack/autopgd_base.py:111,137;
 * check_oscillation's k3, :170-175,415-416); rvlm_apgd_controller is rho = 0.75, the value apgd_train hard-codes
 * (train/apgd_train.py:117,334). */
int rvlm_apgd_controller_rho(int i, int B, int n_iter, int k, int do_check, double rho, const float* loss_i,
                             const uint8_t* pred, float* loss_steps, float* loss_best,
                             float* loss_best_last_check, float* reduced_last_check, float* step,
                             uint8_t* acc, uint8_t* f_notpred, uint8_t* f_improved,
                             uint8_t* f_reduced, rvlm_stream_t stream);
/* The index assignments of the same lines as one pass over the image tensors. */
int rvlm_apgd_select(float* x_adv, float* grad, float* x_best, float* grad_best,
                     float* x_best_adv, const uint8_t* f_notpred, const uint8_t* f_improved,
                     const uint8_t* f_reduced, size_t n_per_sample, int B, rvlm_stream_t stream);
/* APGDAttack random start (autopgd_base.py:210-214,180-183): x + eps * t / (max_b|t| + 1e-12). */
int rvlm_linf_random_start(const float* x, const float* t, float eps, size_t n_per_sample, int B,
                           float* x_adv, rvlm_stream_t stream);
/* Its L2 form (autopgd_base.py:184-185,215-218): x + eps * t / (|t|_2 + 1e-12), t ~ N(0, 1) drawn by the caller (the reference
 * draws it on the CPU generator); the norm is a deterministic fp32 sum in the kernel's own order. */
int rvlm_l2_random_start(const float* x, const float* t, float eps, size_t n_per_sample, int B, float* x_adv,
                         rvlm_stream_t stream);
/* Square Attack, L-inf: one query of SquareAttack.attack_single_run (autoattack/square.py:256-263).  Candidates of the
 * n_active still-robust images idx[a] (int64 indices into x / x_best [*, C, H, W]), written compactly to x_new
 * [n_active, C, H, W]:  clamp(min(max(x_best + window, x - eps), x + eps), 0, 1), window = 2*eps*sign[c] on rows
 * [vh, vh+s) x columns [vw, vw+s) and 0 elsewhere (one window per query for the whole batch).  sign: C floats. */
int rvlm_square_linf_propose(const float* x, const float* x_best, const int64_t* idx, int n_active, int C, int H,
                             int W, int vh, int vw, int s, float eps, const float* sign, float* x_new,
                             rvlm_stream_t stream);
/* square.py:288-291: x_best[idx[a]] = x_new[a] for the candidates with take[a] != 0 (loss improved or misclassified). */
int rvlm_square_accept(float* x_best, const float* x_new, const int64_t* idx, const float* take, int n_active,
                       size_t n_per_image, rvlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole loops, device resident (no host sync inside): replace
 *   pgd()        train/pgd_train.py:5-68          -> rvlm_pgd_run
 *   apgd_train() train/apgd_train.py:125-373      -> rvlm_apgd_run (train_variant = 1)
 *   APGDAttack.attack_single_run autopgd_base.py:205-451 -> rvlm_apgd_run (train_variant = 0)
 * with the FARE / TeCoA losses bound (ComputeLossWrapper, …clip.py:260-274).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t loss_kind;        /* RVLM_LOSS_* */
    int32_t reduction;        /* pgd: MEAN (trainer) ; apgd: NONE */
    int32_t output_normalize; /* F.normalize the embedding */
    int32_t n_classes;        /* C for CE (columns of ref) */
    float logit_scale;        /* 100. */
    const float* ref;         /* L2: embedding_orig [B,D];  CE: T [D,C] */
    const int64_t* targets;   /* [B] (CE, and the argmax test of apgd); may be NULL for pgd+L2 */
    const int64_t* y_target;  /* [B] target classes, RVLM_LOSS_DLR_TARGETED only (else NULL) */
} rvlm_loss_spec;

/* x_adv_out = pgd(...) ; delta0 may be NULL (zeros).  flags: see RVLM_FLAG_*.
 * loss_trace (optional, [iterations]) receives the scalar loss of every iteration. */
int rvlm_pgd_run(rvlm_vit* h, const float* x, const float* delta0, int B,
                 const rvlm_loss_spec* loss, float eps, int iterations, float stepsize,
                 float momentum, int mode_max, float* x_adv_out, float* loss_trace,
                 int32_t* flags, rvlm_stream_t stream);

/* pgd() with the norm as a parameter: norm_kind 0 = L-inf (rvlm_pgd_run), 2 = L2 (rvlm_pgd_l2_update per iteration). */
int rvlm_pgd_run_norm(rvlm_vit* h, const float* x, const float* delta0, int B, const rvlm_loss_spec* loss, int norm_kind,
                      float eps, int iterations, float stepsize, float momentum, int mode_max, float* x_adv_out,
                      float* loss_trace, int32_t* flags, rvlm_stream_t stream);

/* pgd() in two precisions: iterations [0, n_first) evaluate forward + loss + input gradient on h_first, a second handle of
 * the SAME model (its fp32 mode: the reference's own precision, train/pgd_train.py:30-38), the rest on h; the attack state
 * lives in h.  loss->ref should be the h_first-precision embedding (FARE's first cotangent is a difference of nearly equal
 * embeddings).  Both handles need max_batch >= B. */
int rvlm_pgd_run_mixed(rvlm_vit* h, rvlm_vit* h_first, int n_first, const float* x, const float* delta0, int B,
                       const rvlm_loss_spec* loss, int norm_kind, float eps, int iterations, float stepsize, float momentum,
                       int mode_max, float* x_adv_out, float* loss_trace, int32_t* flags, rvlm_stream_t stream);

/* rvlm_pgd_run_mixed with the handoff (ABI 109): iterations [0, n_first) evaluate forward + loss on h_first (fp32 storage) and the
 * input gradient on h's bf16 kernels from h_first's saved forward (rvlm_vit_backward_input_from).  Same max_batch on both. */
int rvlm_pgd_run_mixed_fwd(rvlm_vit* h, rvlm_vit* h_first, int n_first, const float* x, const float* delta0, int B,
                           const rvlm_loss_spec* loss, int norm_kind, float eps, int iterations, float stepsize, float momentum,
                           int mode_max, float* x_adv_out, float* loss_trace, int32_t* flags, rvlm_stream_t stream);

/* APGD L-inf.  x_init: NULL -> start from clamp(x,0,1) (apgd_train) or the caller-provided random
 * start (APGDAttack).  logits_from_head: 0 -> the `argmax(model output)==y` test runs on the
 * embedding (apgd_train quirk); 1 -> on emb @ (logit_scale*T) logits
 * (ClassificationModel, CLIP_eval/clip_robustbench.py:50-69).
 * Outputs (each optional): x_best_adv, x_best, loss_best [B], acc u8[B]. */
int rvlm_apgd_run(rvlm_vit* h, const float* x, const float* x_init, int B,
                  const rvlm_loss_spec* loss, float eps, int n_iter, float alpha,
                  int train_variant, int logits_from_head, float* x_best_adv, float* x_best,
                  float* loss_best, uint8_t* acc, rvlm_stream_t stream);

/* `rho` of APGDAttack for the following rvlm_apgd_run* calls WITH train_variant == 0 on this handle (default 0.75).
 * train/apgd_train.py has no such parameter: train_variant != 0 runs always use 0.75, whatever was set here.
 * Returns RVLM_ERR_ARG for NaN. */
int rvlm_vit_set_apgd_rho(rvlm_vit* h, double rho);

/* SURVEY.md section 8(b) `rvlm_vit_fwd_inputgrad`: ONE iteration's model work in one call - forward of x (+ delta,
 * may be NULL) with the activations kept, the bound FARE / TeCoA loss, and grad_x = d loss / d (x + delta)
 * (pgd_train.py:33-38: out = forward(...); loss = loss_fn(out, targets); autograd.grad(loss, perturbation)).
 * Outputs (each optional): out_emb [B,out_dim], out_loss_per_sample [B], out_loss_scalar [1] (per loss->reduction),
 * out_grad_x [B,3,H,W].  Equivalent to rvlm_vit_forward(save=1) + rvlm_loss_grad + rvlm_vit_backward_input. */
int rvlm_vit_fwd_inputgrad(rvlm_vit* h, const float* x, const float* delta, int B, const rvlm_loss_spec* loss,
                           float* out_emb, float* out_loss_per_sample, float* out_loss_scalar, float* out_grad_x,
                           rvlm_stream_t stream);

/* rvlm_apgd_run with the norm as a parameter: norm_kind 0 = L-inf, 2 = L2 (apgd_train(norm='l2'); train_variant only). */
int rvlm_apgd_run_norm(rvlm_vit* h, const float* x, const float* x_init, int B, const rvlm_loss_spec* loss, int norm_kind,
                       float eps, int n_iter, float alpha, int train_variant, int logits_from_head, float* x_best_adv,
                       float* x_best, float* loss_best, uint8_t* acc, rvlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Input front end: replaces Compose([Resize(size, bicubic), CenterCrop(size),
 * ToTensor()]) over a decoded RGB image (train/adversarial_training_clip.py:105-116; torchvision 0.15.2 over Pillow).
 * img_hwc: device uint8 [H, W, 3]; out_chw: device float32 [3, size, size] in [0,1] (NOT normalised).  Bit-exact with
 * Pillow's 8-bit bicubic resampler.  The taps of an input shape are computed on the host and cached in the object:
 * consecutive images of equal shape cost one kernel launch.  max_input_dim bounds the longer input edge.
 * ------------------------------------------------------------------------------------------- */
typedef struct rvlm_preproc rvlm_preproc;
int rvlm_preproc_create(int size, int max_input_dim, rvlm_preproc** out);
int rvlm_preproc_destroy(rvlm_preproc* p);
int rvlm_preproc_run(rvlm_preproc* p, const uint8_t* img_hwc, int H, int W, float* out_chw, rvlm_stream_t stream);
/* A batch of n decoded images of any mix of shapes in ONE kernel launch (the reference's DataLoader hands over batches of
 * 128, train/adversarial_training_clip.py:119-148): imgs_hwc / H / W are HOST arrays of n device pointers / heights /
 * widths; out: device float32 [n, 3, size, size].  Tables are staged through pinned memory and copied on `stream`; the call
 * does not synchronise the stream.  Same arithmetic as rvlm_preproc_run (bit-identical outputs). */
int rvlm_preproc_run_batch(rvlm_preproc* p, const uint8_t* const* imgs_hwc, const int* H, const int* W, int n, float* out,
                           rvlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel trainer: gradient all-reduce over RCCL / xGMI, one process per GPU.
 * Replaces the reduction the reference gets from torch.nn.DataParallel (train/adversarial_training_clip.py:184-191).
 * Rendezvous: rank 0 calls rvlm_comm_unique_id and hands the 128 bytes to every rank by whatever channel the host has
 * (launcher environment, file, socket); every rank then calls rvlm_comm_create on its own GPU (collective call).
 * rvlm_allreduce_grads: in-place SUM of buf[0 .. count) (device pointer) over the ranks, asynchronous on `stream`; the
 * 1 / world factor goes into rvlm_adamw_step (grad_scale).  The attack entry points need none of this: every quantity of
 * pgd / apgd_train / APGDAttack is per sample.  librccl is loaded at the first rvlm_comm_* call (RVLM_ERR_UNSUPPORTED if
 * the host has none).
 * ------------------------------------------------------------------------------------------- */
#define RVLM_COMM_ID_BYTES 128
#define RVLM_DTYPE_F32 0
#define RVLM_DTYPE_BF16 1
typedef struct rvlm_comm rvlm_comm;
int rvlm_comm_unique_id(uint8_t* out_id /* [RVLM_COMM_ID_BYTES], host memory */);
int rvlm_comm_create(const uint8_t* id, int rank, int world, rvlm_comm** out);
int rvlm_comm_destroy(rvlm_comm* comm);
int rvlm_comm_info(const rvlm_comm* comm, int* rank, int* world);
int rvlm_allreduce_grads(rvlm_comm* comm, void* buf, size_t count, int dtype, rvlm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement support: per-kernel-class HIP-event timing on the engine's stream.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    char name[48];
    double total_ms;  /* sum of launch durations */
    double flops;     /* algorithmic FLOPs of those launches (2/MAC, GEMM-shaped work only) */
    double bytes;     /* algorithmic bytes (HBM-bound kernels) */
    int64_t launches;
} rvlm_profile_entry;
int rvlm_vit_set_profiling(rvlm_vit* h, int enabled);
/* Synchronises the device; writes up to *n entries, sets *n to the count. */
int rvlm_vit_get_profile(rvlm_vit* h, rvlm_profile_entry* out, int* n);
int rvlm_vit_reset_profile(rvlm_vit* h);

const char* rvlm_last_error(void);
int rvlm_version(void);   /* = RVLM_VERSION.  101: rvlm_loss_spec.y_target, rvlm_loss_grad(y_target), DLR losses;
                           * 102: square-attack kernels; 103: rvlm_vit_fwd_inputgrad, rvlm_vit_backward_params_stages,
                           * rvlm_ce_logits, rvlm_head_logits(_bwd), double hyper-parameters in rvlm_adamw_step,
                           * rvlm_pgd_l2_update, rvlm_pgd_run_norm, rvlm_apgd_l2_step, rvlm_apgd_run_norm,
                           * rvlm_project_perturbation, rvlm_normalize_grad; 104: rvlm_preproc_run_batch, rvlm_ce_logits takes B = 1,
                           * rvlm_vit_backward_params_stages refuses out-of-order stages; 105: rvlm_apgd_controller_rho,
                           * rvlm_vit_set_apgd_rho, rvlm_comm_* / rvlm_allreduce_grads; 106: rvlm_pgd_run_mixed, fp32 mode on v_mfma_f32_32x32x2_f32;
                           * 107: rvlm_l2_random_start; 108: RVLM_PREC_F32X3 (split-bf16 linears over fp32 storage);
                           * 109: rvlm_vit_backward_input_from, rvlm_vit_forward_for, rvlm_vit_set_flash_inference, rvlm_pgd_run_mixed_fwd (fp32-storage forward, bf16 backward) */

#ifdef __cplusplus
}
#endif
#endif /* RVLM_H */
