/* rvlm_kernels.h - kernel-level entry points of librvlm.so (unit-test / benchmarking surface).
 *
 * Not part of the drop-in boundary (that is rvlm.h); these expose the individual HIP kernels the
 * encoder is built from so tests/ can check each one against the oracle in isolation.
 * bf16 buffers are passed as uint16_t* (raw bfloat16 bits).  All pointers are device pointers.
 */
#ifndef RVLM_KERNELS_H
#define RVLM_KERNELS_H
#include "rvlm.h"
#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] = epi(A[M,K] @ Bw[N,K]^T): epi 0 bf16 out(+bias); 1 f32 out(+bias)(+residual f32);
 * 2 h=acc+bias: out=bf16 act(h), out_pre=bf16 act'(h); 3 out=bf16 acc*h_pre (h_pre = a stored act'(h));
 * 4 f32 out(+bias).
 * A must be readable for round_up(M,128) rows. */
int rvlm_k_gemm_bf16_nt(const uint16_t* A, long lda, const uint16_t* Bw, long ldb, int M, int N, int K,
                        int a_rows, int epi, const float* bias, void* out, long ldo, uint16_t* out_pre,
                        const uint16_t* h_pre, const float* residual, int act, rvlm_stream_t stream);
/* generic strided batched fp32 GEMM (see robustvlm_amd/csrc/kernels.h: GemmF32) */
int rvlm_k_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C,
                    long scm, long scn, int M, int N, int K, float alpha, const float* bias,
                    rvlm_stream_t stream);
/* the same with one batch dimension (element strides sab / sbb / scb) and the whole epilogue: act >= 0 applies RVLM_ACT_*
 * (C_pre, if given, receives the pre-activation), dact_h multiplies by act'(dact_h[m, n]) (QuickGELU), residual adds */
int rvlm_k_gemm_f32_ex(const float* A, long sam, long sak, long sab, const float* B, long sbn, long sbk, long sbb,
                       float* C, long scm, long scn, long scb, int M, int N, int K, int nb, float alpha,
                       const float* bias, int act, float* C_pre, const float* dact_h, const float* residual,
                       rvlm_stream_t stream);
/* 1: the VALU fmaf-chain tiles for every later fp32 GEMM instead of the v_mfma_f32_32x32x2_f32 tiles (the two are
 * bit-identical: both evaluate every output element as the k-ordered fp32 fma chain); 0: default */
int rvlm_k_gemm_f32_set_valu(int on);
/* fp32 attention path: in-place row softmax of s (backward = 0), or ds = p * (ds - sum(p * ds)) * scale in place (backward = 1);
 * rows of `cols` values, `ld` floats apart */
int rvlm_k_softmax_rows(const float* p, float* s, long rows, int cols, int ld, float scale, int backward, rvlm_stream_t stream);
/* bf16 flash attention on packed qkv [B*S, 3W] (head_dim 64); lse2 [B*H*round_up(S,32)] */
/* fp32 flash forward (csrc/attention_f32.hip): qkv fp32 [B*S, 3*64*H], o fp32 [B*S, 64*H]; optional lse2 [B*H, round_up(S, 32)] (log2
 * domain of 0.125 log2(e) q.k) and bf16 copies qkv_bf / o_bf in the same layouts; S <= 288 */
int rvlm_k_attn_fwd_f32_flash(const float* qkv, float* o, float* lse2, uint16_t* qkv_bf, uint16_t* o_bf, int B, int H, int S,
                              rvlm_stream_t stream);
/* fp32 flash backward: dqkv fp32 [B*S, 3*64*H] from qkv, o, d_o (fp32) and the forward's lse2 rows; dsum [B*H, round_up(S, 32)] scratch;
 * S = 32 NK + 1..4, NK <= 8 */
int rvlm_k_attn_bwd_f32_flash(const float* qkv, const float* o, const float* d_o, const float* lse2, float* dsum, float* dqkv,
                              int B, int H, int S, rvlm_stream_t stream);
int rvlm_k_attn_fwd_bf16(const uint16_t* qkv, uint16_t* o, float* lse2, int B, int H, int S,
                         rvlm_stream_t stream);
int rvlm_k_attn_bwd_bf16(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse2,
                         float* dsum_scratch, uint16_t* dqkv, int B, int H, int S, rvlm_stream_t stream);
/* 1: ds_read_b64_tr_b16 transposed fragments (default); 0: scalar-LDS-read validation variant */
int rvlm_k_attn_set_use_tr(int on);
/* 0: 128x128 GEMM kernel only; 1 / 2: the production dispatch (persistent 256x256 kernel unless the fill rule or the
 * few-row rule sends the shape to the 128x128 kernel); 3: persistent kernel for every shape it can take (M >= 256,
 * N % 256 == 0, K % 128 == 0) regardless of those rules; -1: env RVLM_GEMM_VARIANT (default 2) */
int rvlm_k_gemm_set_variant(int v);
/* Bit mask of the kernel families the last rvlm_k_gemm_bf16_nt call (or the engine's last GEMM) launched */
#define RVLM_GEMM_K_128 1         /* gemm_bf16_nt_kernel, 128x128 tiles */
#define RVLM_GEMM_K_256 2         /* one-tile-per-workgroup 256x256 kernel (EXPERIMENTAL builds only) */
#define RVLM_GEMM_K_PERSISTENT 4  /* gemm_bf16_nt_256p_kernel, the persistent 256x256 kernel */
#define RVLM_GEMM_K_256Q 8        /* 4-wave persistent kernel (EXPERIMENTAL builds only) */
#define RVLM_GEMM_K_SPLITK 16     /* split-K slabs on the 128x128 kernel + splitk_reduce_kernel */
#define RVLM_GEMM_K_STRIP 32      /* remainder rows computed by the persistent kernel's strip phase (same launch) */
#define RVLM_GEMM_K_PINGPONG 64   /* gemm_bf16_nt_256x_kernel, the persistent kernel with phase-shifted wave groups */
int rvlm_k_gemm_last_kernels(void);
/* persistent 256x256 kernel: device buffer of 256*8*4 uint64 receiving per-tile s_memtime stamps (tile start, first
 * K-step done, mainloop done, epilogue issued); NULL switches tracing off */
int rvlm_k_gemm_set_trace(void* ptr);
/* Phase-shifted (ping-pong) persistent kernel, gemm_bf16_256x.hip: `mask` bit e sends epilogue kind e (0 bf16, 1 fp32 +
 * residual, 2 activation pair, 3 x stored act', 4 fp32) to it for K <= kmax when the shape qualifies; mask < 0 restores
 * the environment / default (RVLM_GEMM_PINGPONG, RVLM_GEMM_PINGPONG_KMAX).  rvlm_k_gemm_x_set_trace: device buffer of
 * 256 * 8 * 26 uint64 receiving, per workgroup and wave, s_memtime at kernel start / end and at the start of the MFMAs,
 * the end of the MFMAs and the end of the epilogue of the wave's first 8 tiles; NULL switches tracing off. */
int rvlm_k_gemm_set_pingpong(int mask, int kmax);
int rvlm_k_gemm_x_set_trace(void* ptr);
/* persistent kernel timing experiments (results become garbage): bit 0 no operand DMA, bit 1 no MFMA, bit 2 no LDS
 * fragment reads; 0 = normal */
int rvlm_k_gemm_set_ablate(int v);
/* MFMA shape of the persistent kernel's tile phase: 1 = v_mfma_f32_16x16x32_bf16 (8 x 4 accumulator tiles per wave; the
 * shipped form since round 4), 0 = v_mfma_f32_32x32x16_bf16 (4 x 2; EXPERIMENTAL builds only, RVLM_ERR_UNSUPPORTED otherwise),
 * -1 = environment / build default (RVLM_GEMM_M16).  Same results to fp32 summation order. */
int rvlm_k_gemm_set_m16(int v);
/* measurement only: the persistent GEMM's operand request stream (same tile order, 8 waves x (4 + 4) pieces of 8 rows x
 * 128 B per K-step) with no MFMA and no LDS, `depth` K-steps (depth x 64 KiB per CU) in flight in registers.
 * mode bit 0: one s_barrier per K-step (a wave waits for the slowest wave's pieces, like a GEMM K-step); bit 1: pieces by
 * LDS-DMA into LDS instead of into registers (depth <= 2).  A [M, K], Bw [N, K] bf16 row-major; M, N % 256 == 0,
 * K % 64 == 0; `out` >= 512 words (never written in practice). */
int rvlm_k_probe_operand_stream(const uint16_t* A, const uint16_t* Bw, int M, int N, int K, int depth, int mode,
                                uint32_t* out, rvlm_stream_t stream);
/* workgroups per CU reported by the runtime for attn fwd (96-VGPR build), attn fwd (default), attn dq */
int rvlm_k_attn_occupancy(int S, int* out3);
int rvlm_k_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                             float* mean, float* rstd, int M, int W, rvlm_stream_t stream);
/* bf16 dy and bf16 copy of the residual gradient; dres == NULL: the copy is the stream itself (accumulate reads it); accumulate < 0:
 * only rows that are multiples of -accumulate accumulate, the others are overwritten */
int rvlm_k_layernorm_bwd_bf16(const uint16_t* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                              float* dres, uint16_t* dres_lp, int accumulate, int M, int W, rvlm_stream_t stream);
int rvlm_k_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                             const float* rstd, float* dres, int accumulate, int M, int W,
                             rvlm_stream_t stream);
/* micro-probe of ds_read_b64_tr_b16: out[lane*4+j] for a 64-lane wave reading the bf16 buffer `src`
 * (>= 2048 elements, copied to LDS) at per-lane byte offsets `offs[64]`. */
int rvlm_k_probe_tr16(const uint16_t* src, const int32_t* offs, uint16_t* out, rvlm_stream_t stream);

/* Weight gradient of one linear layer on the split-K path of the training step (train/adversarial_training_clip.py:
 * 356-364, loss.backward()): dW[N,K] (+)= dY[M,N]^T X[M,K] (bf16 operands, fp32 result), dbias[N] (+)= column sums of
 * dY (NULL: skipped).  N, K multiples of 256, M >= 256.  `work`: device scratch of rvlm_k_wgrad_work_bytes(M, N, K). */
size_t rvlm_k_wgrad_work_bytes(int M, int N, int K);
int rvlm_k_wgrad_bf16(const uint16_t* dY, long lddy, const uint16_t* X, long ldx, int M, int N, int K, float* dW,
                      long lddw, int accumulate, float* dbias, void* work, size_t work_bytes, rvlm_stream_t stream);
/* 1: rvlm_k_wgrad_bf16 runs the form the training step used until round 4 (token-chunk transposes + NT GEMM) instead of the
 * copy-free contraction-major GEMM - the A/B arm of tests and benches; 0 (default): the training step's path. */
void rvlm_k_wgrad_set_transposed(int v);

#ifdef __cplusplus
}
#endif
#endif
