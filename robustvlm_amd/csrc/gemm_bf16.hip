// bf16 MFMA GEMM for gfx950, NT form: C[M,N] = A[M,K] x Bw[N,K]^T, fp32 accumulate, fused epilogues.
//
// All the dense work of the encoder (QKV / out-proj / fc1 / fc2 / patch-embed, forward and dgrad)
// runs through this kernel: dgrad uses the same NT form on the pre-transposed weight copy.
//
// Structure (v1):  128x128 block tile, BK = 64, 4 waves (2x2), each wave a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16_bf16; operands staged HBM -> LDS with global_load_lds_dwordx4 (16 B/lane,
// no VGPR round trip), double-buffered, one barrier per K-step; LDS rows are 128 B (= BK bf16) with
// the 16-B chunk index XOR-swizzled by ((row>>1)&7) so the ds_read_b128 fragment reads are
// bank-conflict free (the swizzle is applied on the per-lane GLOBAL source address because the DMA
// writes LDS lane-linearly).  The MFMA is issued as mfma(Bfrag, Afrag) so every lane ends up with 4
// consecutive output columns of one row -> 8/16-byte epilogue accesses.  Workgroup ids are remapped
// XCD-aware (each XCD's L2 sees a contiguous group of tiles) with GROUP_M=8 tile grouping.
#include "kernels.h"
#include "gemm_epilogue.h"
#include "../../include/rvlm_kernels.h"

namespace rvlm {

constexpr int GB_M = 128, GB_N = 128, GB_K = 64;
constexpr int GB_TILE_BYTES = GB_M * GB_K * 2;  // 16 KiB per operand per buffer

__device__ __forceinline__ void glds16(const void* gptr, void* lds_ptr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_ptr, 16, 0, 0);
}

template <int EPI>
__global__ void __launch_bounds__(256)
gemm_bf16_nt_kernel(GemmBf16 p, int tiles_m, int tiles_n, int a_rows, int desync) {
    __shared__ __attribute__((aligned(16))) char lds[4 * GB_TILE_BYTES];
    if (gridDim.y > 1) {   // split-K slice blockIdx.y: fp32 partial sums into its own slab of p.out
        const int ks = p.K / (int)gridDim.y;
        p.A += (long)blockIdx.y * ks;
        p.Bw += (long)blockIdx.y * ks;
        p.K = ks;
        p.out = (float*)p.out + (long)blockIdx.y * p.M * p.ldo;
    }

    // ---- XCD-aware, grouped tile order ----------------------------------------------------
    const int nwg = gridDim.x, pid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = pid & 7, loc = pid >> 3;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int group_size = 8 * tiles_n;
    const int first_m = (t / group_size) * 8;
    const int gm = min(tiles_m - first_m, 8);
    const int tm = first_m + (t % group_size) % gm;
    const int tn = (t % group_size) / gm;
    const int m0 = tm * GB_M, n0 = tn * GB_N;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;

    // ---- phase offset (performance only): co-resident workgroups otherwise run in lockstep, so their
    // epilogue store bursts and pipeline fills coincide chip-wide instead of hiding under the partner's
    // MFMA phase.  Half of the first wave of workgroups starts `desync` x 3.4 us late.
    if (desync > 0 && pid < 2 * 256 && ((pid >> 3) & 1)) {
        for (int i = 0; i < desync; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- staging: wave w loads rows [32w, 32w+32) of both operand tiles, 8 rows per DMA ----
    const int srow = lane >> 3, cphys = lane & 7;
    const bf16_t* a_src[4];
    const bf16_t* b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = w * 32 + j * 8 + srow;
        const int clog = cphys ^ ((r >> 1) & 7);
        const int ar = min(m0 + r, a_rows - 1);
        const int br = min(n0 + r, p.N - 1);
        a_src[j] = p.A + (long)ar * p.lda + clog * 8;
        b_src[j] = p.Bw + (long)br * p.ldb + clog * 8;
    }
    char* const stage_base = lds + (w * 32) * 128;

    // ---- fragment read offsets (bytes inside an operand tile) ------------------------------
    const int l31 = lane & 31, hi = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + hi) ^ swz) << 4;
    const int a_row_off = (wm * 64 + l31) * 128;
    const int b_row_off = (wn * 64 + l31) * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / GB_K;
    // prologue: stage tile 0 into buffer 0
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        glds16(a_src[j], stage_base + j * 1024);
        glds16(b_src[j], stage_base + GB_TILE_BYTES + j * 1024);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        __syncthreads();  // drains this wave's DMA (vmcnt(0)) and orders LDS reuse
        if (kt + 1 < nk) {
            char* dst = stage_base + (buf ^ 1) * 2 * GB_TILE_BYTES;
            const int ko = (kt + 1) * GB_K;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                glds16(a_src[j] + ko, dst + j * 1024);
                glds16(b_src[j] + ko, dst + GB_TILE_BYTES + j * 1024);
            }
        }
        const char* At = lds + buf * 2 * GB_TILE_BYTES;
        const char* Bt = At + GB_TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a0 = *(const bf16x8*)(At + a_row_off + koff[kk]);
            bf16x8 a1 = *(const bf16x8*)(At + a_row_off + 32 * 128 + koff[kk]);
            bf16x8 b0 = *(const bf16x8*)(Bt + b_row_off + koff[kk]);
            bf16x8 b1 = *(const bf16x8*)(Bt + b_row_off + 32 * 128 + koff[kk]);
            // swapped operands: D[n][m] -> lane holds 4 consecutive n for m = lane&31
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
        }
    }

    // ---- epilogue (wave-private LDS transpose -> full-line global accesses) ----
    __syncthreads();   // every wave is done reading the pipeline buffers
    if (desync == -1 && acc[0][0][0] != 123456.789f) return;   // ablation: no stores
    gemm_epilogue<EPI, 2, 2>(acc, p, m0 + wm * 64, n0 + wn * 64, lane, lds + w * EPI_LDS_BYTES_PER_WAVE);
}

// ---- split-K for few-row GEMMs ------------------------------------------------------------------------------
// Few-row problems (the class-token tail: M = batch rows; with RVLM_GEMM_TAIL=0 also the <=255 remainder rows of a big
// problem, which the persistent kernel otherwise computes in its own launch) are a handful of 128x128 tiles: as N/128
// workgroups walking the whole K serially they cost 46-61 us at K = 3072/4096 (latency-bound: ~1 us per 64-deep
// K-step).  They are split SPLITK ways along K into fp32 slabs (deterministic: no atomics) and combined by a tiny
// reduce+epilogue kernel.
template <int EPI>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ slabs, int splitk, GemmBf16 p) {
    const int n4 = p.N >> 2;
    const long total = (long)p.M * n4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / n4), n = (int)(idx - (long)m * n4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int sidx = 0; sidx < splitk; ++sidx) {
            const float4 t = *(const float4*)(slabs + ((long)sidx * p.M + m) * p.N + n);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (p.bias) { const float4 b = *(const float4*)(p.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        const long o = (long)m * p.ldo + n;
        float f[4] = {v.x, v.y, v.z, v.w};
        if (EPI == EPI_BF16) {
            bf16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)f[e];
            *(bf16x4*)((bf16_t*)p.out + o) = ov;
        } else if (EPI == EPI_F32_RESID) {
            if (p.residual) { const float4 r = *(const float4*)(p.residual + o); f[0] += r.x; f[1] += r.y; f[2] += r.z; f[3] += r.w; }
            *(float4*)((float*)p.out + o) = make_float4(f[0], f[1], f[2], f[3]);
        } else if (EPI == EPI_BF16_ACT) {
            bf16x4 pv, ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float av, dv;
                act_pair(f[e], p.act, av, dv);
                pv[e] = (bf16_t)dv;
                ov[e] = (bf16_t)av;
            }
            if (p.out_pre) *(bf16x4*)(p.out_pre + o) = pv;
            *(bf16x4*)((bf16_t*)p.out + o) = ov;
        } else if (EPI == EPI_BF16_DACT) {
            const bf16x4 hv = *(const bf16x4*)(p.h_pre + o);
            bf16x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(f[e] * (float)hv[e]);
            *(bf16x4*)((bf16_t*)p.out + o) = ov;
        } else {
            *(float4*)((float*)p.out + o) = make_float4(f[0], f[1], f[2], f[3]);
        }
    }
}


int gemm_bf16_nt_256p(const GemmBf16& p, int* rows_done, hipStream_t s);
int gemm_bf16_nt_256x(const GemmBf16& p, int* rows_done, hipStream_t s);
#ifdef RVLM_EXPERIMENTAL_GEMM   // ablation kernels (make EXPERIMENTAL=1): not part of the shipped library
int gemm_bf16_nt_256(const GemmBf16& p, int* rows_done, hipStream_t s);
int gemm_bf16_nt_256q(const GemmBf16& p, int* rows_done, hipStream_t s);
#endif

// Which kernel families the last gemm_bf16_nt() call launched (test surface: rvlm_k_gemm_last_kernels): the parity
// tests assert that the family they name is the one that ran - a dispatch rule must not silently move them.
static int g_last_kernels = 0;
int gemm_last_kernels() { return g_last_kernels; }

static int g_waves = -1;   // persistent kernel flavour: 8 waves x 128x64 (default) or 4 waves x 128x128
static int gemm_waves() {
    if (g_waves < 0) { const char* e = getenv("RVLM_GEMM_WAVES"); g_waves = e ? atoi(e) : 8; }
    return g_waves;
}

static int g_persist = -1;
static int gemm_persist() {
    if (g_persist < 0) { const char* e = getenv("RVLM_GEMM_PERSIST"); g_persist = e ? atoi(e) : 1; }
    return g_persist;
}

// 0: 128x128 kernel only; 1 / 2 (default): persistent 256x256 kernel where the shape qualifies and the fill rule
// below does not send it to the 128x128 kernel; 3: persistent kernel for EVERY shape it can take (M >= 256,
// N % 256 == 0, K % 128 == 0), fill rule and few-row rule off - the parity tests of the persistent kernel run under 3.
// (EXPERIMENTAL builds with RVLM_GEMM_PERSIST=0: the older per-shape choice between the one-tile-per-workgroup 256x256
// kernel and the 128x128 kernel.)
static int g_gemm_variant = -1;
void gemm_set_variant(int v) { g_gemm_variant = v; }
// RVLM_GEMM_PINGPONG: bit mask over epilogue kinds (GemmEpi) that take gemm_bf16_256x.hip; RVLM_GEMM_PINGPONG_KMAX: largest K
static int g_pingpong_mask = -1, g_pingpong_kmax = -1;
static int gemm_pingpong_mask() {
    if (g_pingpong_mask < 0) { const char* e = getenv("RVLM_GEMM_PINGPONG"); g_pingpong_mask = e ? atoi(e) : RVLM_PINGPONG_DEFAULT_MASK; }
    return g_pingpong_mask;
}
static int gemm_pingpong_kmax() {
    if (g_pingpong_kmax < 0) { const char* e = getenv("RVLM_GEMM_PINGPONG_KMAX"); g_pingpong_kmax = e ? atoi(e) : RVLM_PINGPONG_DEFAULT_KMAX; }
    return g_pingpong_kmax;
}
// mask < 0: back to the environment / default for BOTH values (whatever kmax says)
void gemm_set_pingpong(int mask, int kmax) {
    if (mask < 0) { g_pingpong_mask = -1; g_pingpong_kmax = -1; }
    else { g_pingpong_mask = mask; g_pingpong_kmax = kmax; }
}
bool gemm_has_pingpong() {
#ifdef RVLM_EXPERIMENTAL_GEMM
    return true;
#else
    return false;
#endif
}
static int gemm_variant() {
    if (g_gemm_variant < 0) {
        const char* e = getenv("RVLM_GEMM_VARIANT");
        g_gemm_variant = e ? atoi(e) : 2;
    }
    return g_gemm_variant;
}

static int g_desync = -1;
static int gemm_desync() {
    if (g_desync < 0) { const char* e = getenv("RVLM_GEMM_DESYNC"); g_desync = e ? atoi(e) : 0; }
    return g_desync;
}

static int gemm_bf16_nt_128(const GemmBf16& p, hipStream_t s) {
    const int desync = gemm_desync();
    const int tiles_m = cdiv(p.M, GB_M), tiles_n = cdiv(p.N, GB_N);
    const int a_rows = p.a_rows > 0 ? p.a_rows : p.M;
    dim3 grid(tiles_m * tiles_n), block(256);
    switch (p.epi) {
        case EPI_BF16:
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_BF16>), grid, block, 0, s, p, tiles_m, tiles_n, a_rows, desync);
            break;
        case EPI_F32_RESID:
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_F32_RESID>), grid, block, 0, s, p, tiles_m, tiles_n, a_rows, desync);
            break;
        case EPI_BF16_ACT:
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_BF16_ACT>), grid, block, 0, s, p, tiles_m, tiles_n, a_rows, desync);
            break;
        case EPI_BF16_DACT:
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_BF16_DACT>), grid, block, 0, s, p, tiles_m, tiles_n, a_rows, desync);
            break;
        case EPI_F32:
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_F32>), grid, block, 0, s, p, tiles_m, tiles_n, a_rows, desync);
            break;
        default:
            return fail(RVLM_ERR_ARG, "gemm_bf16_nt: unknown epilogue");
    }
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// Split-K weight gradient on the persistent kernel.  dW[N,K] (+)= sum_b tA_b tB_b^T with tA = [splits][N][Kc] and
// tB = [splits][K][Kc] (token chunks of the transposed operands, written by transpose_split): one batched launch of
// splits * (N/256) * (K/256) work items, fp32 slabs [splits][N][K], then the deterministic slab reduce.
// Plan: enough splits to give every CU one work item (the W x W out-projection gradient has 16 output tiles).
// fp32 slab bytes the plan below asks for at its largest split count (M >= 128 * 16 tokens): what a trainable handle must own
// for the copy-free weight gradient of an [N, K] linear (0: the shape never takes that path)
size_t wgrad_slab_bytes(int N, int K) {
    if (N % 256 != 0 || K % 256 != 0) return 0;
    const int tiles = (N / 256) * (K / 256);
    return (size_t)std::max(1, std::min(16, 256 / tiles)) * N * K * sizeof(float);
}
int wgrad_split_plan(int M, int N, int K, size_t slab_bytes, int* splits, int* Kc) {
    if (N % 256 != 0 || K % 256 != 0 || M < 256) return 0;
    const int tiles = (N / 256) * (K / 256), nk128 = cdiv(M, 128);
    int sp = std::max(1, std::min(16, 256 / tiles));
    sp = std::min(sp, nk128);
    // (also for one split: the copy-free contraction-major form always writes fp32 slabs and reduces them - a plan whose slab
    // does not fit is no plan, and the caller falls back to the transposing path instead of failing in the launch)
    if ((size_t)sp * N * K * sizeof(float) > slab_bytes) return 0;
    *splits = sp;
    *Kc = cdiv(nk128, sp) * 128;
    return 1;
}
int gemm_bf16_wgrad_split(const bf16_t* tA, const bf16_t* tB, int splits, int Kc, int N, int K, float* dW, long lddw,
                          int accumulate, float* slab, size_t slab_bytes, hipStream_t s) {
    if (splits > 1 && (!slab || (size_t)splits * N * K * sizeof(float) > slab_bytes))
        return fail(RVLM_ERR_STATE, "gemm_bf16_wgrad_split: slab scratch too small");
    GemmBf16 g;
    g.A = tA; g.lda = Kc; g.Bw = tB; g.ldb = Kc;
    g.M = splits * N; g.N = K; g.K = Kc;
    g.batch_m_rows = N;
    if (splits == 1) {
        g.epi = accumulate ? EPI_F32_RESID : EPI_F32; g.residual = accumulate ? dW : nullptr;
        g.out = dW; g.ldo = lddw;
    } else {
        g.epi = EPI_F32; g.out = slab; g.ldo = K;
    }
    int done = 0;
    int rc = gemm_bf16_nt_256p(g, &done, s);
    if (rc) return rc;
    if (done != g.M) return fail(RVLM_ERR_UNSUPPORTED, "gemm_bf16_wgrad_split: shape not covered by the persistent kernel");
    if (splits > 1) {
        GemmBf16 r;
        r.M = N; r.N = K; r.out = dW; r.ldo = lddw; r.residual = accumulate ? dW : nullptr;
        const int rb = cdiv((long)N * (K / 4), 256);
        if (accumulate)
            hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32_RESID>), dim3(rb), dim3(256), 0, s, slab, splits, r);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32>), dim3(rb), dim3(256), 0, s, slab, splits, r);
        RVLM_CHECK_LAUNCH();
    }
    return RVLM_OK;
}

// Weight gradient without operand copies: the persistent kernel's contraction-major form (gemm_bf16_256p.hip, TN) reads dY and X
// token-major as the backward left them; split-K over token chunks of Kc (the tail chunk's rows beyond M read as zero), fp32
// slabs, deterministic reduce.
int gemm_bf16_wgrad_tn(const bf16_t* dY, long lddy, const bf16_t* X, long ldx, int M, int splits, int Kc, int N, int K, float* dW,
                       long lddw, int accumulate, float* slab, size_t slab_bytes, hipStream_t s) {
    // The kernel addresses its operands with 32-bit byte offsets: token ranges beyond 2 GiB of either operand go in row chunks, the
    // later ones accumulating (dY and X are token-major, so a chunk is a pointer offset; the slabs are reused in stream order).
    const long row_bytes = 2 * std::max(lddy, ldx);
    const long addressable = ((1L << 31) - 1) / row_bytes;                    // k-rows a launch can reach (its last chunk is padded)
    const long max_rows = (addressable - 17 * 128) / 256 * 256;                // chunk size: its padded split stays addressable
    if ((long)splits * Kc > addressable) {
        if (max_rows < 256) return fail(RVLM_ERR_UNSUPPORTED, "gemm_bf16_wgrad_tn: leading dimension too large");
        for (long r0 = 0; r0 < M; r0 += max_rows) {
            const int rows = (int)std::min<long>(max_rows, M - r0);
            int sp = 0, kc = 0;
            if (!wgrad_split_plan(std::max(rows, 256), N, K, slab_bytes, &sp, &kc))       // (a tail of < 256 tokens: one zero-padded chunk)
                return fail(RVLM_ERR_UNSUPPORTED, "gemm_bf16_wgrad_tn: N, K % 256");
            int rc = gemm_bf16_wgrad_tn(dY + r0 * lddy, lddy, X + r0 * ldx, ldx, rows, sp, kc, N, K, dW, lddw, r0 > 0 ? 1 : accumulate, slab,
                                        slab_bytes, s);
            if (rc) return rc;
        }
        return RVLM_OK;
    }
    if (!slab || (size_t)splits * N * K * sizeof(float) > slab_bytes)
        return fail(RVLM_ERR_STATE, "gemm_bf16_wgrad_tn: slab scratch too small");
    if ((long)splits * Kc < M) return fail(RVLM_ERR_ARG, "gemm_bf16_wgrad_tn: the token chunks do not cover M");
    GemmBf16 g;
    g.A = dY; g.lda = lddy; g.Bw = X; g.ldb = ldx;
    g.M = splits * N; g.N = K; g.K = Kc;
    g.batch_m_rows = N; g.tn = 1; g.k_rows = M;
    g.epi = EPI_F32; g.out = slab; g.ldo = K;
    int done = 0;
    int rc = gemm_bf16_nt_256p(g, &done, s);
    if (rc) return rc;
    if (done != g.M) return fail(RVLM_ERR_UNSUPPORTED, "gemm_bf16_wgrad_tn: shape not covered by the persistent kernel");
    GemmBf16 r;
    r.M = N; r.N = K; r.out = dW; r.ldo = lddw; r.residual = accumulate ? dW : nullptr;
    const int rb = cdiv((long)N * (K / 4), 256);
    if (accumulate)
        hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32_RESID>), dim3(rb), dim3(256), 0, s, slab, splits, r);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32>), dim3(rb), dim3(256), 0, s, slab, splits, r);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

int gemm_bf16_nt(const GemmBf16& p, hipStream_t s) {
    if (!p.A || !p.Bw || !p.out || p.M <= 0 || p.N <= 0 || p.K <= 0)
        return fail(RVLM_ERR_ARG, "gemm_bf16_nt: bad arguments");
    if (p.K % GB_K != 0 || p.N % 4 != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.ldo % 4 != 0)
        return fail(RVLM_ERR_UNSUPPORTED, "gemm_bf16_nt: need K%64==0, N%4==0, lda/ldb%8==0, ldo%4==0");
    if (p.epi == EPI_BF16_DACT && !p.h_pre) return fail(RVLM_ERR_ARG, "gemm_bf16_nt: h_pre");
    // deep-K, few-tile problems (the weight-gradient GEMMs: K = all tokens, <= 256 output tiles): split K so
    // that >= 2 workgroups per CU are busy, fp32 slabs + deterministic reduce/epilogue kernel
    if (p.K >= 8192 && (p.epi == EPI_F32 || p.epi == EPI_F32_RESID) && p.splitk) {
        const int tiles = cdiv(p.M, GB_M) * cdiv(p.N, GB_N);
        int splitk = 1;
        const int nk64 = p.K / GB_K;
        for (int cand : {2, 3, 4, 6, 12}) {
            if (nk64 % cand == 0 && (size_t)cand * p.M * p.N * sizeof(float) <= p.splitk_bytes) {
                splitk = cand;
                if (tiles * cand >= 512) break;
            }
        }
        if (splitk > 1) {
            GemmBf16 part = p;
            part.epi = EPI_F32; part.bias = nullptr; part.residual = nullptr; part.out = p.splitk;
            part.ldo = p.N; part.out_pre = nullptr; part.h_pre = nullptr;
            part.a_rows = p.a_rows > 0 ? p.a_rows : p.M;
            const int tiles_m = cdiv(p.M, GB_M), tiles_n = cdiv(p.N, GB_N);
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_F32>), dim3(tiles_m * tiles_n, splitk), dim3(256), 0, s, part,
                               tiles_m, tiles_n, part.a_rows, 0);
            RVLM_CHECK_LAUNCH();
            const int rb = cdiv((long)p.M * (p.N / 4), 256);
            if (p.epi == EPI_F32_RESID)
                hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32_RESID>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, p);
            else
                hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, p);
            RVLM_CHECK_LAUNCH();
            g_last_kernels = RVLM_GEMM_K_SPLITK;
            return RVLM_OK;
        }
    }
    int done = 0;
    const int variant = gemm_variant();
    g_last_kernels = 0;
    // few-row problems (the class-token rows of the last block: M = batch): a handful of 256x256 tiles cannot fill the
    // chip - they take the 128x128 kernel with split-K below
    const bool small_m = p.M <= 512 && variant != 3;
    // 1 or 2 (default): the persistent 256x256 kernel wherever the shape qualifies (N % 256 == 0, K % 128 == 0); it
    // beats both older kernels on every encoder shape (scripts/gemm_bench.py, profiles/).  RVLM_GEMM_PERSIST=0
    // restores the per-shape choice between the one-tile-per-workgroup 256x256 kernel and the 128x128 kernel.
    // how full the persistent kernel's rounds of <= 256 tiles would be: a small problem (ViT-B/32 at batch 128: 75-300
    // tiles) leaves most CUs idle or runs a nearly empty last round - the 128x128 kernel (4x the tiles, 2 workgroups per
    // CU) then wins (+17 % on that model) although its mainloop is slower
    bool sparse = false;
    if (p.N % 256 == 0 && p.M >= 256) {
        const long t256 = (long)(p.M / 256) * (p.N / 256);
        static float min_fill = -1.0f;
        if (min_fill < 0.0f) { const char* e = getenv("RVLM_GEMM_MIN_FILL"); min_fill = e ? (float)atof(e) : 0.7f; }
        const float fill = (float)t256 / (float)(((t256 + 255) / 256) * 256);
        // deep-K tiles amortise the persistent kernel's per-tile cost: there only a really empty chip (< 45 %) switches
        sparse = variant != 3 && (fill < 0.45f || (fill < min_fill && p.K <= 1024));
    }
    if (variant != 0 && gemm_persist() && !small_m && !sparse) {
        int rc;
#ifdef RVLM_EXPERIMENTAL_GEMM
        const bool no_pre = p.epi == EPI_BF16_ACT && !p.out_pre;   // only the default kernel knows the one-output form
        if (gemm_waves() == 4 && !no_pre) { rc = gemm_bf16_nt_256q(p, &done, s); if (done) g_last_kernels |= RVLM_GEMM_K_256Q; }
        else
#endif
        {
            rc = RVLM_OK;
#ifdef RVLM_EXPERIMENTAL_GEMM
            // the phase-shifted (ping-pong) form of the persistent kernel (experimental/gemm_bf16_256x.hip: bit-identical,
            // 3-15 % slower, round 3), per epilogue kind (bit e of the mask, e = the kind the kernel will RUN: fp32 +
            // residual without a residual is the plain fp32 kind) and up to a K limit
            const int epi_run = (p.epi == EPI_F32_RESID && !p.residual) ? (int)EPI_F32 : (int)p.epi;
            if (((gemm_pingpong_mask() >> epi_run) & 1) && p.K <= gemm_pingpong_kmax()) {
                rc = gemm_bf16_nt_256x(p, &done, s);
                if (done) g_last_kernels |= RVLM_GEMM_K_PINGPONG | (done > (p.M / 256) * 256 ? RVLM_GEMM_K_STRIP : 0);
            }
#endif
            if (!rc && !done) {
                rc = gemm_bf16_nt_256p(p, &done, s);
                if (done) g_last_kernels |= RVLM_GEMM_K_PERSISTENT | (done > (p.M / 256) * 256 ? RVLM_GEMM_K_STRIP : 0);
            }
        }
        if (rc) return rc;
        if (done >= p.M) return RVLM_OK;
    }
#ifdef RVLM_EXPERIMENTAL_GEMM
    const bool big = variant == 1 || (variant == 2 && (p.epi == EPI_BF16 || p.K >= 2048));
    if (done == 0 && big && !small_m && !sparse) {
        int rc = gemm_bf16_nt_256(p, &done, s);
        if (rc) return rc;
        if (done) g_last_kernels |= RVLM_GEMM_K_256;
        if (done >= p.M) return RVLM_OK;
    }
#endif
    GemmBf16 r = p;
    if (done > 0) {   // remainder rows [done, M) on the 128x128 kernel
        const bool f32out = (p.epi == EPI_F32_RESID || p.epi == EPI_F32);
        r.A = p.A + (long)done * p.lda;
        r.out = (char*)p.out + (long)done * p.ldo * (f32out ? 4 : 2);
        if (p.out_pre) r.out_pre = p.out_pre + (long)done * p.ldo;
        if (p.h_pre) r.h_pre = p.h_pre + (long)done * p.ldo;
        if (p.residual) r.residual = p.residual + (long)done * p.ldo;
        r.M = p.M - done;
        r.a_rows = (p.a_rows > 0 ? p.a_rows : p.M) - done;
    }
    if ((done > 0 || small_m) && variant != 0) {
        // split-K: the strip is a handful of 128x128 tiles walking K serially (~1 us per 64-deep step, latency-bound).
        // Cut K so that >= 128 workgroups share the walk, each keeping >= 2 K-steps; fp32 slabs + reduce/epilogue kernel.
        int splitk = 1;
        {
            const int nk64 = p.K / GB_K, tiles = cdiv(r.M, GB_M) * cdiv(p.N, GB_N);
            for (int cand : {2, 3, 4, 6, 8, 12, 16}) {
                if (nk64 % cand != 0 || nk64 / cand < 2 || tiles * cand > 512) continue;
                if ((size_t)cand * r.M * p.N * sizeof(float) > p.splitk_bytes) continue;
                splitk = cand;
                if (tiles * cand >= 128) break;
            }
        }
        const size_t need = (size_t)splitk * r.M * p.N * sizeof(float);
        if (splitk > 1 && r.M <= 512 && p.splitk && need <= p.splitk_bytes && p.K % (splitk * GB_K) == 0) {
            GemmBf16 part = r;
            part.epi = EPI_F32; part.bias = nullptr; part.residual = nullptr; part.out = p.splitk;
            part.ldo = p.N; part.out_pre = nullptr; part.h_pre = nullptr;
            part.a_rows = r.a_rows > 0 ? r.a_rows : r.M;
            const int tiles_m = cdiv(part.M, GB_M), tiles_n = cdiv(part.N, GB_N);
            hipLaunchKernelGGL((gemm_bf16_nt_kernel<EPI_F32>), dim3(tiles_m * tiles_n, splitk), dim3(256), 0, s, part,
                               tiles_m, tiles_n, part.a_rows, 0);
            RVLM_CHECK_LAUNCH();
            const int rb = cdiv((long)r.M * (p.N / 4), 256);
            switch (p.epi) {
                case EPI_BF16: hipLaunchKernelGGL((splitk_reduce_kernel<EPI_BF16>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, r); break;
                case EPI_F32_RESID: hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32_RESID>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, r); break;
                case EPI_BF16_ACT: hipLaunchKernelGGL((splitk_reduce_kernel<EPI_BF16_ACT>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, r); break;
                case EPI_BF16_DACT: hipLaunchKernelGGL((splitk_reduce_kernel<EPI_BF16_DACT>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, r); break;
                default: hipLaunchKernelGGL((splitk_reduce_kernel<EPI_F32>), dim3(rb), dim3(256), 0, s, p.splitk, splitk, r); break;
            }
            RVLM_CHECK_LAUNCH();
            g_last_kernels |= RVLM_GEMM_K_SPLITK;
            return RVLM_OK;
        }
    }
    g_last_kernels |= RVLM_GEMM_K_128;
    return gemm_bf16_nt_128(r, s);
}

}  // namespace rvlm
