// Kernel-level C entry points (include/rvlm_kernels.h) used by tests/ to check each HIP kernel in
// isolation against the oracle.
#include "kernels.h"
#include "../../include/rvlm_kernels.h"

namespace rvlm {
void attn_set_use_tr(int on);
void gemm_set_variant(int v);
int gemm_last_kernels();
void gemm_set_trace(unsigned long long* ptr);
void gemm_set_pingpong(int mask, int kmax);
bool gemm_has_pingpong();
#ifdef RVLM_EXPERIMENTAL_GEMM
void gemm_x_set_trace(unsigned long long* ptr);
#endif
void gemm_set_ablate(int v);
void gemm_set_m16(int v);
bool gemm_has_m32();
int attn_occupancy(int S, int* out3);

__global__ void probe_tr16_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ offs,
                                  bf16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) tile[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)((__attribute__((address_space(3))) char*)tile + offs[lane]));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

// ---------------------------------------------------------------------------------------------------------------
// Operand-stream probe (measurement only, no product path uses it): the persistent GEMM's global-memory request
// stream - 256 workgroups x 8 waves, tile -> workgroup order of gemm_bf16_nt_256p_kernel, per K-step every wave
// fetches 4 + 4 pieces of 8 rows x 128 B of its A and B tiles - with NO MFMA and NO LDS, loaded straight into
// registers, DEPTH K-steps (DEPTH x 64 KiB per CU) in flight.  Answers one question: does L2 -> CU delivery per
// clock grow with more bytes in flight than the GEMM's 160 KiB LDS ring can hold (<= 96 KiB)?
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4p gload16(const char* sbase, unsigned voff) {
    u32x4p r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}
// MODE bit 2: the 8 lanes of a row read its eight 16-B chunks in the XOR-permuted order the GEMM's LDS swizzle imposes on
// the global side ((lane & 7) ^ ((row >> 1) & 7)); bit 3: permuted within 64-B halves only ((row >> 1) & 3);
// MODE bit 4 (with bit 1): the LDS-DMA as MUBUF buffer_load ... lds with a descriptor, a per-lane offset and an SGPR offset
// (the GEMM's form) instead of global_load_lds;
// MODE bit 0: one s_barrier per K-step (every wave then waits for the slowest wave's pieces, as a GEMM K-step does);
// MODE bit 1: the pieces go to LDS by LDS-DMA (global_load_lds_dwordx4) instead of to registers (DEPTH <= 2: 64 KiB each)
template <int DEPTH, int MODE>
__global__ void __launch_bounds__(512)
probe_stream_kernel(const char* __restrict__ A, const char* __restrict__ Bw, long ld_bytes, int tiles_m, int tiles_n,
                    int nk, unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(1024))) char plds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = tiles_m * tiles_n;
    constexpr bool DMA = (MODE & 2) != 0;
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = w * 32 + j * 8 + (lane >> 3);
        const int chunk = (MODE & 4) ? ((lane & 7) ^ ((row >> 1) & 7)) : (MODE & 8) ? ((lane & 7) ^ ((row >> 1) & 3)) : (lane & 7);
        voff[j] = (unsigned)(row * ld_bytes + chunk * 16);
    }
    u32x4p P[DMA ? 1 : DEPTH][8], sink = {0u, 0u, 0u, 0u};
    const auto a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, 0x7fffffff, 0x00020000);
    const auto b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bw), 0, 0x7fffffff, 0x00020000);
    auto fetch = [&](int d, const char* a, const char* b, int k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (DMA && (MODE & 16)) {
                char* dst = plds + d * 65536 + w * 8192 + j * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void*)dst, 16, (int)voff[j],
                                                         __builtin_amdgcn_readfirstlane((int)(a - A) + k * 128), 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void*)(dst + 4096), 16, (int)voff[j],
                                                         __builtin_amdgcn_readfirstlane((int)(b - Bw) + k * 128), 0, 0);
            } else if (DMA) {
                char* dst = plds + d * 65536 + w * 8192 + j * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + (long)k * 128 + voff[j]),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + (long)k * 128 + voff[j]),
                                                 (__attribute__((address_space(3))) void*)(dst + 4096), 16, 0, 0);
            } else {
                P[d][j] = gload16(a + (long)k * 128, voff[j]);
                P[d][4 + j] = gload16(b + (long)k * 128, voff[j]);
            }
        }
    };
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = tile & 7, loc = tile >> 3;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        const int group_size = 8 * tiles_n, first_m = (t / group_size) * 8, gm = min(tiles_m - first_m, 8);
        const long m0 = (long)(first_m + (t % group_size) % gm) * 256, n0 = (long)((t % group_size) / gm) * 256;
        const char* a = A + m0 * ld_bytes;
        const char* b = Bw + n0 * ld_bytes;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d, a, b, d);            // prologue: DEPTH K-steps in flight
        for (int kt = DEPTH; kt < nk + DEPTH; kt += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                // the oldest generation has landed when at most 8 * (DEPTH - 1) younger pieces are outstanding
                if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                if (MODE & 1) __builtin_amdgcn_s_barrier();
                if (!DMA) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) sink ^= P[d][j];
                }
                const int k = kt + d;
                if (k < nk) fetch(d, a, b, k);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (sink.x == 0x12345678u && sink.y == 0x9abcdef0u) out[threadIdx.x] = sink.z ^ sink.w;
}
// The persistent GEMM's ring protocol itself, nothing else: 160 KiB of LDS as an A ring of 3 half-stage slots and a B ring
// of 2; at the barrier that ends K-step g every wave requests B(g+2) then A(g+3); before the barrier it waits with
// vmcnt(KEEP) (the GEMM keeps the 4 youngest pieces = the A half in flight).  ORDER 1 requests A before B.
template <int KEEP, int ORDER>
__global__ void __launch_bounds__(512)
probe_ring_kernel(const char* __restrict__ A, const char* __restrict__ Bw, long ld_bytes, int tiles_m, int tiles_n, int nk) {
    extern __shared__ __attribute__((aligned(1024))) char plds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ntiles = tiles_m * tiles_n;
    unsigned voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) voff[j] = (unsigned)((w * 32 + j * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16);
    const int ntw = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto origin = [&](int tile, long& m0, long& n0) {
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = tile & 7, loc = tile >> 3;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        const int group_size = 8 * tiles_n, first_m = (t / group_size) * 8, gm = min(tiles_m - first_m, 8);
        m0 = (long)(first_m + (t % group_size) % gm) * 256; n0 = (long)((t % group_size) / gm) * 256;
    };
    // per-stream cursors over all tiles of this workgroup (next stage to request), advanced incrementally
    const int total = ntw * nk;
    int a_ti = 0, a_kt = 0, a_slot = 0, b_ti = 0, b_kt = 0, b_slot = 0;
    const char *a_src, *b_src;
    {
        long m0, n0;
        origin(blockIdx.x, m0, n0);
        a_src = A + m0 * ld_bytes; b_src = Bw + n0 * ld_bytes;
    }
    auto issue = [&](bool is_a, int) {
        int& ti = is_a ? a_ti : b_ti; int& kt = is_a ? a_kt : b_kt; int& slot = is_a ? a_slot : b_slot;
        const char*& src = is_a ? a_src : b_src;
        if (ti >= ntw) return;
        char* dst = plds + (is_a ? slot * 32768 : 98304 + slot * 32768) + w * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)kt * 128 + voff[j]),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
        slot = is_a ? (slot == 2 ? 0 : slot + 1) : (slot ^ 1);
        if (++kt == nk) {
            kt = 0;
            if (++ti < ntw) {
                long m0, n0;
                origin(blockIdx.x + ti * gridDim.x, m0, n0);
                src = is_a ? A + m0 * ld_bytes : Bw + n0 * ld_bytes;
            }
        }
    };
    int ga = 0, gb = 0;
    issue(true, ga++); issue(false, gb++);
    issue(true, ga++); issue(false, gb++);
    issue(true, ga++);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int g = 0; g < total; ++g) {
        if (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ORDER == 0) { issue(false, gb++); issue(true, ga++); }
        else { issue(true, ga++); issue(false, gb++); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace rvlm
using namespace rvlm;

extern "C" int rvlm_k_probe_operand_stream(const uint16_t* A, const uint16_t* Bw, int M, int N, int K, int depth, int mode,
                                           uint32_t* out, rvlm_stream_t stream) {
    if (M % 256 || N % 256 || K % 64 || depth < 1 || depth > 4 || K / 64 < depth || mode < 0 || mode > 35 ||
        (mode < 32 && (mode & 2) && depth > 2))
        return fail(RVLM_ERR_ARG, "rvlm_k_probe_operand_stream: M, N % 256, K % 64, depth 1..4 (<= 2 with LDS-DMA), mode 0..3");
    const int tm = M / 256, tn = N / 256, nk = K / 64, grid = std::min(tm * tn, 256);
    const long ldb = (long)K * 2;
    hipStream_t s = (hipStream_t)stream;
    if (mode >= 32) {
        const size_t ring = 163840;
#define RING(KP, OR)                                                                                                    \
        do {                                                                                                             \
            (void)hipFuncSetAttribute((const void*)probe_ring_kernel<KP, OR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ring); \
            hipLaunchKernelGGL((probe_ring_kernel<KP, OR>), dim3(grid), dim3(512), ring, s, (const char*)A, (const char*)Bw, ldb, tm, tn, nk); \
        } while (0)
        if (mode == 32) RING(4, 0); else if (mode == 33) RING(4, 1); else if (mode == 34) RING(0, 0); else RING(8, 0);
#undef RING
        RVLM_CHECK_LAUNCH();
        return RVLM_OK;
    }
    const size_t lds = (mode & 2) ? (size_t)depth * 65536 : 0;
#define PROBE(D, MD)                                                                                                  \
    do {                                                                                                               \
        if (lds) (void)hipFuncSetAttribute((const void*)probe_stream_kernel<D, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((probe_stream_kernel<D, MD>), dim3(grid), dim3(512), lds, s, (const char*)A, (const char*)Bw, ldb, tm, tn, nk, out); \
    } while (0)
    switch (depth * 32 + mode) {
        case 32: PROBE(1, 0); break;
        case 33: PROBE(1, 1); break;
        case 34: PROBE(1, 2); break;
        case 35: PROBE(1, 3); break;
        case 64: PROBE(2, 0); break;
        case 65: PROBE(2, 1); break;
        case 66: PROBE(2, 2); break;
        case 67: PROBE(2, 3); break;
        case 96: PROBE(3, 0); break;
        case 97: PROBE(3, 1); break;
        case 128: PROBE(4, 0); break;
        case 129: PROBE(4, 1); break;
        case 36: PROBE(1, 4); break;
        case 68: PROBE(2, 4); break;
        case 39: PROBE(1, 7); break;
        case 71: PROBE(2, 7); break;
        case 40: PROBE(1, 8); break;
        case 72: PROBE(2, 8); break;
        case 43: PROBE(1, 11); break;
        case 75: PROBE(2, 11); break;
        case 50: PROBE(1, 18); break;
        case 82: PROBE(2, 18); break;
        case 51: PROBE(1, 19); break;
        case 83: PROBE(2, 19); break;
        default: return fail(RVLM_ERR_ARG, "rvlm_k_probe_operand_stream: depth / mode combination");
    }
#undef PROBE
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

extern "C" int rvlm_k_gemm_bf16_nt(const uint16_t* A, long lda, const uint16_t* Bw, long ldb, int M, int N,
                                   int K, int a_rows, int epi, const float* bias, void* out, long ldo,
                                   uint16_t* out_pre, const uint16_t* h_pre, const float* residual, int act,
                                   rvlm_stream_t stream) {
    // split-K slabs for few-row problems (test surface only; allocated once, never freed)
    static float* scratch = nullptr;
    const size_t bytes = (size_t)8 * 512 * 4096 * sizeof(float);
    if (!scratch && hipMalloc((void**)&scratch, bytes) != hipSuccess) scratch = nullptr;
    GemmBf16 g;
    g.A = (const bf16_t*)A; g.lda = lda; g.Bw = (const bf16_t*)Bw; g.ldb = ldb; g.M = M; g.N = N; g.K = K;
    g.a_rows = a_rows; g.epi = epi; g.bias = bias; g.out = out; g.ldo = ldo; g.out_pre = (bf16_t*)out_pre;
    g.h_pre = (const bf16_t*)h_pre; g.residual = residual; g.act = act;
    g.splitk = scratch; g.splitk_bytes = scratch ? bytes : 0;
    return gemm_bf16_nt(g, (hipStream_t)stream);
}
extern "C" int rvlm_k_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk,
                               float* C, long scm, long scn, int M, int N, int K, float alpha,
                               const float* bias, rvlm_stream_t stream) {
    GemmF32 g;
    g.A = A; g.sam = sam; g.sak = sak; g.B = B; g.sbn = sbn; g.sbk = sbk; g.C = C; g.scm = scm; g.scn = scn;
    g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.bias = bias;
    return gemm_f32(g, (hipStream_t)stream);
}
extern "C" int rvlm_k_gemm_f32_ex(const float* A, long sam, long sak, long sab, const float* B, long sbn, long sbk, long sbb,
                                  float* C, long scm, long scn, long scb, int M, int N, int K, int nb, float alpha,
                                  const float* bias, int act, float* C_pre, const float* dact_h, const float* residual,
                                  rvlm_stream_t stream) {
    GemmF32 g;
    g.A = A; g.sam = sam; g.sak = sak; g.sab2 = sab; g.B = B; g.sbn = sbn; g.sbk = sbk; g.sbb2 = sbb;
    g.C = C; g.scm = scm; g.scn = scn; g.scb2 = scb; g.M = M; g.N = N; g.K = K; g.nb1 = 1; g.nb2 = nb; g.alpha = alpha;
    g.bias = bias; g.act = act; g.C_pre = C_pre; g.dact_h = dact_h; g.residual = residual;
    return gemm_f32(g, (hipStream_t)stream);
}
extern "C" int rvlm_k_softmax_rows(const float* p, float* s, long rows, int cols, int ld, float scale, int backward,
                                   rvlm_stream_t stream) {
    return backward ? softmax_rows_bwd(p, s, rows, cols, ld, scale, (hipStream_t)stream) : softmax_rows_fwd(s, rows, cols, ld, (hipStream_t)stream);
}
extern "C" int rvlm_k_gemm_f32_set_valu(int on) { gemm_f32_set_valu(on); return RVLM_OK; }
extern "C" int rvlm_k_attn_fwd_f32_flash(const float* qkv, float* o, float* lse2, uint16_t* qkv_bf, uint16_t* o_bf, int B, int H, int S,
                                         rvlm_stream_t stream) {
    int rc = RVLM_OK;
    if (!attn_fwd_f32_flash(qkv, o, lse2, (int)round_up(S, 32), (bf16_t*)qkv_bf, (bf16_t*)o_bf, B, H, S, (hipStream_t)stream, &rc))
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_k_attn_fwd_f32_flash: sequence length not covered");
    return rc;
}
extern "C" int rvlm_k_attn_bwd_f32_flash(const float* qkv, const float* o, const float* d_o, const float* lse2, float* dsum, float* dqkv,
                                         int B, int H, int S, rvlm_stream_t stream) {
    int rc = RVLM_OK;
    if (!attn_bwd_f32_flash(qkv, o, d_o, lse2, (int)round_up(S, 32), dsum, dqkv, B, H, S, (hipStream_t)stream, &rc))
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_k_attn_bwd_f32_flash: sequence length not covered");
    return rc;
}
extern "C" int rvlm_k_attn_fwd_bf16(const uint16_t* qkv, uint16_t* o, float* lse2, int B, int H, int S,
                                    rvlm_stream_t stream) {
    return attn_fwd_bf16((const bf16_t*)qkv, 3L * H * 64, (bf16_t*)o, H * 64L, lse2, B, H, S, (hipStream_t)stream);
}
extern "C" int rvlm_k_attn_bwd_bf16(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o,
                                    const float* lse2, float* dsum_scratch, uint16_t* dqkv, int B, int H, int S,
                                    rvlm_stream_t stream) {
    return attn_bwd_bf16((const bf16_t*)qkv, 3L * H * 64, (const bf16_t*)o, H * 64L, (const bf16_t*)d_o, H * 64L,
                         lse2, dsum_scratch, (bf16_t*)dqkv, 3L * H * 64, B, H, S, (hipStream_t)stream);
}
extern "C" int rvlm_k_attn_set_use_tr(int on) { attn_set_use_tr(on); return RVLM_OK; }
extern "C" int rvlm_k_gemm_set_variant(int v) { gemm_set_variant(v); return RVLM_OK; }
extern "C" int rvlm_k_gemm_last_kernels(void) { return gemm_last_kernels(); }
extern "C" int rvlm_k_gemm_set_ablate(int v) { gemm_set_ablate(v); return RVLM_OK; }
extern "C" int rvlm_k_gemm_set_m16(int v) {
    if (v == 0 && !gemm_has_m32())
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_k_gemm_set_m16: this build holds the 16x16x32 form only (make EXPERIMENTAL=1)");
    gemm_set_m16(v);
    return RVLM_OK;
}
extern "C" int rvlm_k_gemm_set_trace(void* ptr) { gemm_set_trace((unsigned long long*)ptr); return RVLM_OK; }
// the shipped library does not contain the ping-pong kernel (make EXPERIMENTAL=1 builds it): a non-zero mask is refused
extern "C" int rvlm_k_gemm_set_pingpong(int mask, int kmax) {
    if (mask > 0 && !gemm_has_pingpong())
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_k_gemm_set_pingpong: this build has no ping-pong kernel (make EXPERIMENTAL=1)");
    gemm_set_pingpong(mask, kmax);
    return RVLM_OK;
}
extern "C" int rvlm_k_gemm_x_set_trace(void* ptr) {
#ifdef RVLM_EXPERIMENTAL_GEMM
    gemm_x_set_trace((unsigned long long*)ptr);
    return RVLM_OK;
#else
    return ptr ? fail(RVLM_ERR_UNSUPPORTED, "rvlm_k_gemm_x_set_trace: this build has no ping-pong kernel (make EXPERIMENTAL=1)") : RVLM_OK;
#endif
}
extern "C" int rvlm_k_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                                        float* mean, float* rstd, int M, int W, rvlm_stream_t stream) {
    return layernorm_fwd<float>(x, W, gamma, beta, y, W, mean, rstd, M, W, (hipStream_t)stream);
}
// bf16 dy / bf16 copy; dres == NULL: the bf16 copy is the residual-gradient stream itself (read to accumulate, written back)
extern "C" int rvlm_k_layernorm_bwd_bf16(const uint16_t* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                         float* dres, uint16_t* dres_lp, int accumulate, int M, int W, rvlm_stream_t stream) {
    return layernorm_bwd<bf16_t, bf16_t>((const bf16_t*)dy, W, x, W, gamma, mean, rstd, dres, W, (bf16_t*)dres_lp, W, accumulate, M, W,
                                         (hipStream_t)stream);
}
extern "C" int rvlm_k_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                        const float* rstd, float* dres, int accumulate, int M, int W,
                                        rvlm_stream_t stream) {
    return layernorm_bwd<float, float>(dy, W, x, W, gamma, mean, rstd, dres, W, nullptr, W, accumulate, M, W,
                                       (hipStream_t)stream);
}
extern "C" int rvlm_k_probe_tr16(const uint16_t* src, const int32_t* offs, uint16_t* out, rvlm_stream_t stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)src, offs,
                       (bf16_t*)out);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
extern "C" int rvlm_k_attn_occupancy(int S, int* out3) { return attn_occupancy(S, out3); }

static int g_wgrad_transposed = 0;
// dW[N,K] (+)= dY[M,N]^T X[M,K], dbias[N] (+)= column sums of dY, on the split-K path of the training step.  `work` is a
// caller-provided device buffer of rvlm_k_wgrad_work_bytes(M, N, K) bytes (token-chunk operands, fp32 slabs, partials).
extern "C" size_t rvlm_k_wgrad_work_bytes(int M, int N, int K) {
    const size_t Mpt = (size_t)round_up(M, 128) + 16 * 128;
    return Mpt * (size_t)(N + K) * 2 + (Mpt / 64) * (size_t)N * 4 + (size_t)16 * N * K * 4 + 4096;
}
extern "C" int rvlm_k_wgrad_bf16(const uint16_t* dY, long lddy, const uint16_t* X, long ldx, int M, int N, int K,
                                 float* dW, long lddw, int accumulate, float* dbias, void* work, size_t work_bytes,
                                 rvlm_stream_t stream) {
    int splits = 0, Kc = 0;
    const size_t Mpt = (size_t)round_up(M, 128) + 16 * 128;
    char* w = (char*)work;
    bf16_t* tA = (bf16_t*)w; w += Mpt * N * 2;
    bf16_t* tB = (bf16_t*)w; w += Mpt * K * 2;
    float* red = (float*)w; const size_t redf = (Mpt / 64) * (size_t)N; w += redf * 4;
    float* slab = (float*)w; const size_t slab_bytes = (size_t)16 * N * K * 4;
    if ((size_t)(w - (char*)work) + slab_bytes > work_bytes) return fail(RVLM_ERR_ARG, "rvlm_k_wgrad_bf16: work buffer too small");
    int rc = RVLM_OK;
    if (!wgrad_split_plan(M, N, K, slab_bytes, &splits, &Kc)) rc = fail(RVLM_ERR_UNSUPPORTED, "rvlm_k_wgrad_bf16: N, K % 256, M >= 256");
    if (g_wgrad_transposed) {       // the token-chunk transposes + NT form the training step used until round 4 (A/B arm)
        if (!rc) rc = transpose_split((const bf16_t*)dY, lddy, M, N, tA, Kc, splits, dbias, accumulate, red, redf, (hipStream_t)stream);
        if (!rc) rc = transpose_split((const bf16_t*)X, ldx, M, K, tB, Kc, splits, nullptr, 0, nullptr, 0, (hipStream_t)stream);
        if (!rc) rc = gemm_bf16_wgrad_split(tA, tB, splits, Kc, N, K, dW, lddw, accumulate, slab, slab_bytes, (hipStream_t)stream);
        return rc;
    }
    // the training step's path (engine.hip wgrad<bf16_t>): column sums + the contraction-major persistent GEMM on the operands as they lie
    if (!rc && dbias) rc = colsum<bf16_t>((const bf16_t*)dY, lddy, M, N, dbias, accumulate, red, redf, (hipStream_t)stream);
    if (!rc) rc = gemm_bf16_wgrad_tn((const bf16_t*)dY, lddy, (const bf16_t*)X, ldx, M, splits, Kc, N, K, dW, lddw, accumulate, slab,
                                     slab_bytes, (hipStream_t)stream);
    return rc;
}
extern "C" void rvlm_k_wgrad_set_transposed(int v) { g_wgrad_transposed = v; }
