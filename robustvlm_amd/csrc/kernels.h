// Internal launch API of the librvlm kernels (host side).  Each returns an rvlm_status.
#pragma once
#include "common.h"

namespace rvlm {

// ---------------------------------------------------------------------------------------------
// Generic strided / batched fp32 GEMM (VALU, fmaf chain over k in order): the fp32 parity path
// and all the small GEMMs (head projection, CE logits).
//   C[b1,b2][m,n] = epi( alpha * sum_k A[b1,b2][m*sam + k*sak] * B[b1,b2][n*sbn + k*sbk] )
// epi: +bias[n] -> (C2 = pre-activation) -> act -> *mul[m,n] (act' form) -> +residual[m,n]
// ---------------------------------------------------------------------------------------------
struct GemmF32 {
    const float* A = nullptr; long sam = 0, sak = 0, sab1 = 0, sab2 = 0;
    const float* B = nullptr; long sbn = 0, sbk = 0, sbb1 = 0, sbb2 = 0;
    float* C = nullptr;       long scm = 0, scn = 1, scb1 = 0, scb2 = 0;
    int M = 0, N = 0, K = 0, nb1 = 1, nb2 = 1;
    float alpha = 1.0f;
    const float* bias = nullptr;       // [N]
    int act = -1;                      // -1 none, else RVLM_ACT_*
    float* C_pre = nullptr;            // optional copy of the pre-activation (same layout as C)
    const float* dact_h = nullptr;     // optional: multiply by act'(dact_h[m,n]) (layout of C)
    int dact_kind = RVLM_ACT_QUICK_GELU;
    const float* residual = nullptr;   // optional: + residual[m,n] (layout of C)
    // bit 0 (A) / bit 1 (B): along its contiguous dimension the operand is READABLE up to the next multiple of 4 elements,
    // and where that dimension is k the padding holds zeros (the engine's [.., S, round_up(S, 4)] score matrices): lets the
    // S = 257 attention products of the fp32 mode use 16-byte loads
    int pad4 = 0;
};
int gemm_f32(const GemmF32& p, hipStream_t s);
void gemm_f32_set_valu(int on);   // test hook: 1 = the VALU fmaf-chain tiles instead of the fp32 MFMA tiles

// ---------------------------------------------------------------------------------------------
// bf16 MFMA GEMM, NT form: C[M,N] = A[M,K] (row-major, lda) x Bw[N,K]^T (row-major, ldb)
// fp32 accumulate.  Requirements: K % 64 == 0, A/Bw rows readable up to round_up(M|N,128)
// (buffers are padded), N % 4 == 0.
// ---------------------------------------------------------------------------------------------
enum GemmEpi {
    EPI_BF16 = 0,          // out bf16 = acc (+bias)
    EPI_F32_RESID = 1,     // out f32  = acc (+bias) (+residual f32)
    EPI_BF16_ACT = 2,      // h = acc+bias: out bf16 = act(h); out_pre bf16 = act'(h) (all the backward needs of h)
    EPI_BF16_DACT = 3,     // out bf16 = acc * h_pre, h_pre bf16 = the act'(h) an EPI_BF16_ACT launch stored
    EPI_F32 = 4            // out f32 = acc (+bias)
};
struct GemmBf16 {
    const bf16_t* A = nullptr; long lda = 0;
    const bf16_t* Bw = nullptr; long ldb = 0;
    int M = 0, N = 0, K = 0;
    int a_rows = 0;                     // allocated (readable) rows of A, >= M; 0 -> M
    int epi = EPI_BF16;
    const float* bias = nullptr;        // [N] or null
    void* out = nullptr; long ldo = 0;  // bf16 or f32 per epi
    bf16_t* out_pre = nullptr;          // EPI_BF16_ACT: act'(h) (ld = ldo); null: not written (forward-only callers)
    const bf16_t* h_pre = nullptr;      // EPI_BF16_DACT: act'(h) of the matching forward (ld = ldo)
    const float* residual = nullptr;    // EPI_F32_RESID (ld = ldo)
    int act = RVLM_ACT_QUICK_GELU;
    unsigned long long* trace = nullptr;   // persistent kernel only: per-tile s_memtime stamps (test hook)
    float* splitk = nullptr;            // fp32 slab scratch for the split-K paths (few-row GEMMs); null: no split-K
    size_t splitk_bytes = 0;
    int stagger = 0;                    // persistent kernel only: odd workgroup groups start stagger x ~4 us late
    int group_m = 4;                    // persistent kernel only: m-tiles per tile-order group (RVLM_GEMM_GROUP_M)
    int wave_prio = 0;                  // persistent kernel only: s_setprio for waves 4-7 (RVLM_GEMM_PRIO experiment)
    int krot = 0;                       // persistent kernel only: K-step rotation per workgroup (RVLM_GEMM_KROT experiment)
    int batch_m_rows = 0;               // persistent kernel only, > 0: batched form - rows [b*batch_m_rows, ...) of A meet
                                        // rows [b*N, (b+1)*N) of Bw (the split-K weight-gradient GEMM, gemm_bf16_wgrad)
    int tn = 0;                         // persistent kernel only, 1: CONTRACTION-major operands - A = [k_rows][lda] (columns = output
    int k_rows = 0;                     // rows of a batch), Bw = [k_rows][ldb] (columns = output columns); batch b contracts k-rows
                                        // [b*K, (b+1)*K), rows >= k_rows read as zero (gemm_bf16_wgrad_tn)
};
int gemm_bf16_nt(const GemmBf16& p, hipStream_t s);
// default routing to the phase-shifted persistent kernel (gemm_bf16_256x.hip): epilogue-kind mask and K limit, from
// same-box A/Bs on the encoder's shapes (profiles/r03_*); RVLM_GEMM_PINGPONG / RVLM_GEMM_PINGPONG_KMAX override
#ifndef RVLM_PINGPONG_DEFAULT_MASK
#define RVLM_PINGPONG_DEFAULT_MASK 0
#endif
#ifndef RVLM_PINGPONG_DEFAULT_KMAX
#define RVLM_PINGPONG_DEFAULT_KMAX 1024
#endif

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (eps 1e-5, biased variance), one wave per row.
// ---------------------------------------------------------------------------------------------
template <typename TO>
int layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, TO* y, long ldy,
                  float* mean, float* rstd, int M, int W, hipStream_t s);
// dres[m,:] (+)= LN'(dy[m,:]) ; optionally also writes dres as TB (A operand of the next dgrad GEMM)
// accumulate=0 overwrites dres; accumulate=-S accumulates on rows that are multiples of S only (the other rows of
// dres are overwritten: the class-token tail leaves a gradient on the class rows and garbage elsewhere).
template <typename TI, typename TB>
int layernorm_bwd(const TI* dy, long lddy, const float* x, long ldx, const float* gamma,
                  const float* mean, const float* rstd, float* dres, long lddres, TB* dres_lp,
                  long ldlp, int accumulate, int M, int W, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Patch embedding glue
// ---------------------------------------------------------------------------------------------
// A0[(b*g*g + py*g + px), c*P*P + i*P + j] = ((x(+delta))[b,c,py*P+i,px*P+j] - mean_c)/std_c ; cols
// >= 3*P*P zero-filled up to Kpad.
template <typename T>
int im2col_normalize(const float* x, const float* delta, int B, int img, int P, const float* mean3,
                     const float* std3, T* A0, long lda, int Kpad, hipStream_t s);
// grad_x[b,c,y,x] = dA0[row, col] / std_c
template <typename T>
int col2im_grad(const T* dA0, long lda, int B, int img, int P, const float* std3, float* grad_x,
                hipStream_t s);
// tokens = [cls ; patch_out] + pos ; x0 = ln_pre(tokens)
template <typename T>
int embed_lnpre_fwd(const T* patch_out, long ldp, const float* cls, const float* pos,
                    const float* gamma, const float* beta, float* x0, long ldx, float* mean,
                    float* rstd, int B, int S, int W, hipStream_t s, float* tokens_out = nullptr);
// d_patch[b*(S-1)+s-1,:] = LN'(dx0[b*S+s,:]) for s >= 1
template <typename T>
int embed_lnpre_bwd(const float* dx0, long lddx, const float* patch_out, long ldp, const float* cls,
                    const float* pos, const float* gamma, const float* mean, const float* rstd,
                    T* d_patch, long lddp, int B, int S, int W, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Head: pooled = ln_post(x[b,0,:]); emb = pooled @ proj; optional L2 normalise
// ---------------------------------------------------------------------------------------------
int l2_normalize_fwd(const float* e, float* out, float* inv_norm, int B, int D, hipStream_t s);
// d_raw = (d - ehat*(ehat.d)) * inv_norm   (F.normalize backward, eps 1e-12)
int l2_normalize_bwd(const float* d_out, const float* e_raw, const float* inv_norm, float* d_raw,
                     int B, int D, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Softmax rows for the fp32 attention path (scores materialised): in place
// ---------------------------------------------------------------------------------------------
// rows of `cols` values, `ld` floats apart.  lse2 != null: the rows are the [cols x cols] score matrices of (image, head) pairs and
// lse2[(row / cols) * lse_ld + row % cols] = log2(sum_j exp(s_j)) - the bf16 flash kernels' log-sum-exp convention (attention_bf16.hip),
// so that their backward can run on a forward this path computed (the handoff of engine.hip)
int softmax_rows_fwd(float* s, long rows, int cols, int ld, hipStream_t st, float* lse2 = nullptr, int lse_ld = 0);
// ds = p * (dp - sum_j p_j dp_j) * scale, in place over dp
int softmax_rows_bwd(const float* p, float* dp, long rows, int cols, int ld, float scale, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// bf16 flash attention (head_dim 64), qkv packed [M, 3W] bf16 (q | k | v), tokens of image b at
// rows b*S .. b*S+S-1.
// ---------------------------------------------------------------------------------------------
int attn_fwd_bf16(const bf16_t* qkv, long ldqkv, bf16_t* o, long ldo, float* lse, int B, int H,
                  int S, hipStream_t s);
int attn_bwd_bf16(const bf16_t* qkv, long ldqkv, const bf16_t* o, long ldo, const bf16_t* d_o,
                  long lddo, const float* lse, float* dsum_scratch, bf16_t* dqkv, long lddqkv,
                  int B, int H, int S, hipStream_t s);
// fp32 flash forward (attention_f32.hip): O = softmax(0.125 Q K^T) V without materialised scores, on v_mfma_f32_32x32x2_f32; optional
// lse2 rows (the bf16 flash kernels' convention) and bf16 copies of q | k | v and o.  false: shape not covered.
bool attn_fwd_f32_flash_covers(int S);
bool attn_fwd_f32_flash(const float* qkv, float* o, float* lse2, int lse_ld, bf16_t* qkv_bf, bf16_t* o_bf, int B, int H, int S, hipStream_t s,
                        int* rc_out);
// fp32 flash backward (attention_f32.hip): dqkv from qkv, o, d_o and the forward's lse2 rows - no kept probabilities; dsum: scratch
// [B * H, lse_ld].  Covered: S = 32 NK + 1..4, NK <= 8 (S = 257).
bool attn_bwd_f32_flash_covers(int S);
bool attn_bwd_f32_flash(const float* qkv, const float* o, const float* d_o, const float* lse2, int lse_ld, float* dsum, float* dqkv, int B,
                        int H, int S, hipStream_t s, int* rc_out);
// class-token attention of the last block (only the class token's query row is live there): o [B, W] (row b = image b),
// lse [B*H] natural log; bwd writes dqkv [B*S, 3W] in full (dQ rows of the other tokens are zero)
int attn_cls_fwd_bf16(const bf16_t* qkv, long ldqkv, bf16_t* o, long ldo, float* lse, int B, int H, int S,
                      hipStream_t s);
int attn_cls_bwd_bf16(const bf16_t* qkv, long ldqkv, const bf16_t* o, long ldo, const bf16_t* d_o, long lddo,
                      const float* lse, bf16_t* dqkv, long lddqkv, int B, int H, int S, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------------------------
int convert_f32_to_bf16(const float* src, long lds_, bf16_t* dst, long ldd, int rows, int cols,
                        int transpose, hipStream_t s);
bool convert_f32_to_bf16_pair(const float* src, long lds_, bf16_t* nk, long ld_nk, bf16_t* t, long ld_t, int rows,
                              int cols, hipStream_t s);  // dst[c,r] if transpose
int scale_copy_f32(const float* src, float* dst, size_t n, float alpha, hipStream_t s);
int fill_f32(float* dst, size_t n, float v, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// training-step support (train_kernels.hip)
// ---------------------------------------------------------------------------------------------
template <typename T>
int transpose_pad(const T* in, long ldi, int R, int C, T* out, long ldo, int Rp, hipStream_t s);
// split-K weight gradient (bf16): plan -> (splits, Kc), 0 when the shape is not covered; operands transposed into
// [splits][C][Kc] token chunks (optionally emitting the bias gradient = column sums); batched persistent GEMM + reduce
int wgrad_split_plan(int M, int N, int K, size_t slab_bytes, int* splits, int* Kc);
size_t wgrad_slab_bytes(int N, int K);
// split-bf16 ("x3") linears of the fp32-storage engine (x3_kernels.hip)
int x3_split_rows(const float* A, long lda, bf16_t* A3, int M, int rows_out, int K, hipStream_t s);
int x3_prepare_weight(const float* src, int rows, int cols, bf16_t* nk3, bf16_t* t3, hipStream_t s);
bool x3_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, bf16_t* A3, float* mean, float* rstd, int M,
                      int rows_out, int W, hipStream_t s);
// dact_bf != null (mode 0 only): also writes bf16(act'(h)) there (row stride ld_dact) - the bf16 backward's stored derivative (handoff)
int x3_act(const float* hbuf, long ldh, float* out, long ldo, bf16_t* A3, int M, int rows_out, int N, int act, int mode, hipStream_t s,
           bf16_t* dact_bf = nullptr, long ld_dact = 0);
// handoff of a saved fp32-storage forward to the bf16 backward: out = bf16(A) (dact = 0) or bf16(act'(A)) (dact = 1)
int x3_export_bf16(const float* A, long lda, bf16_t* out, long ldo, int M, int N, int act, int dact, hipStream_t s);
int transpose_split(const bf16_t* in, long ldi, int R, int C, bf16_t* out, int Kc, int splits, float* dbias,
                    int accumulate, float* red, size_t red_floats, hipStream_t s);
// the same from the operands AS THEY LIE (dY [M, N] ld lddy, X [M, K] ld ldx, token-major): no transposed copies
int gemm_bf16_wgrad_tn(const bf16_t* dY, long lddy, const bf16_t* X, long ldx, int M, int splits, int Kc, int N, int K, float* dW,
                       long lddw, int accumulate, float* slab, size_t slab_bytes, hipStream_t s);
int gemm_bf16_wgrad_split(const bf16_t* tA, const bf16_t* tB, int splits, int Kc, int N, int K, float* dW, long lddw,
                          int accumulate, float* slab, size_t slab_bytes, hipStream_t s);
// `red` / `red_floats`: caller-owned scratch for the partial column sums (no process-wide state)
template <typename T>
int colsum(const T* in, long ld, int R, int C, float* out, int accumulate, float* red, size_t red_floats,
           hipStream_t s);
template <typename T>
int ln_param_grad(const T* dy, long lddy, const float* x, long ldx, const float* mean, const float* rstd, int R,
                  int C, float* dgamma, float* dbeta, int accumulate, float* red, size_t red_floats, hipStream_t s);
int pos_cls_grad(const float* dtok, long ld, int B, int S, int W, float* dpos, float* dcls, int accumulate,
                 hipStream_t s);
template <typename T>
int gather_patch_rows(const float* dtok, long ld, int B, int S, int W, T* d_patch, long ldp, hipStream_t s);

}  // namespace rvlm
