// bf16 MFMA GEMM for gfx950, PERSISTENT 256x256 variant ("256p", 8 waves x 128x64 wave tiles).
//
// Why: with one 256x256 workgroup per CU and one tile per workgroup, every CU of the chip runs its pipeline fill and
// its epilogue at the same moment.  Measured on the encoder's shapes (M = 32 896, K = 1024) the per-tile fixed cost
// was 13 us (plain bf16 store) to 33 us (fp32 residual read + write) on a 28 us mainloop.
//
// Structure:
//   * <= 256 workgroups (one per CU, 160 KiB of LDS each) walk the tile list; tile -> (m, n) keeps the XCD-aware
//     grouped order of the other kernels.
//   * BK = 64 (128-B LDS rows = one full L2 line per operand row per K-step; 64-B rows halve the payload per request
//     and measured 10 TB/s against 15 TB/s of operand DMA).  LDS = A ring of 3 half-stage slots + B ring of 2: the A
//     halves run three K-steps ahead of the MFMAs, the B halves two, continuously across tile seams.  One barrier
//     per K-step, fragments double-buffered in registers and read between the MFMAs of the previous k-slice, counted
//     vmcnt (the A half's youngest stage stays in flight).
//   * the two waves of a SIMD take opposite roles in the operand traffic: waves 4-7 request all of B right after the
//     barrier, waves 0-3 all of A at the end of their step, so one of them issues MFMAs while the other waits in the
//     texture queue (role split below; +4-6 % on every shape over the lockstep form).
//   * epilogue = wave-private LDS transpose (in the B slot the tile's last K-step freed, through inline-asm DS ops so
//     that hipcc does not drain the operand stream) -> full-line 16-B-per-lane buffer stores; fp32 residual / h_pre
//     are read in the same coalesced pattern, prefetched four 32x32 sub-tiles ahead; bias joins the accumulators
//     before the staging; accumulators are re-zeroed sub-tile by sub-tile.
//   * every global address is a buffer descriptor + a 32-bit per-lane offset that is constant for the whole kernel
//     (nothing lane-dependent is recomputed or spilled around the tile loop).
//
// Requirements: M % 256 == 0 rows handled here, N % 256 == 0, K % 128 == 0, all byte extents < 2 GiB.
#include "kernels.h"
#include "gemm_epilogue.h"
#include "gemm_persist.h"
#include "gemm_strip.h"
#include <type_traits>


// THIS FILE HOLDS THE MEASURED FORM ONLY (round 5): split roles, fragment reads between the MFMAs, the v_mfma_f32_16x16x32_bf16
// tile phase, two side-input slots, in-place last phase.  Every A/B arm it was measured against - the 32x32x16 shape, the
// lockstep and grouped-read schedules, the ablation / wait-sum / timeline instantiations, start stagger, K rotation, static
// priority, the register-path operand stream - lives in experimental/gemm_bf16_256p_abl.patch, a patch over THIS file that
// `make EXPERIMENTAL=1` applies (-> build_exp/gen/gemm_bf16_256p_abl.hip) and builds INSTEAD of this file (librvlm_exp.so): one
// source, so a change here reaches the A/B library too or fails its build where an arm no longer fits.  The production kernels
// of the two are the same instruction streams (the assembly of the nine instantiations was diffed when this file was cut).
namespace rvlm {

// Contraction-major form: both transposing reads of one fragment - tile t (its pair index XORed into the per-lane base + slot
// address `base`) of the 32-deep slice ks.  (A namespace-scope function: as a lambda called from the kernel's other lambdas it
// made hipcc drop the instantiation's host stub.)
static __device__ __forceinline__ void tr_frag(unsigned base, int t, int ks, i32x4& out) {
    const unsigned ad = base ^ (unsigned)(t << 5);
    i32x2 lo, hi2;
    if (ks == 0) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ad));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi2) : "v"(ad));
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:16384" : "=v"(lo) : "v"(ad));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:18432" : "=v"(hi2) : "v"(ad));
    }
    out = i32x4{lo[0], lo[1], hi2[0], hi2[1]};
}

// FORM = 8192 (the tile phase's MFMA shape v_mfma_f32_16x16x32_bf16: kept in the kernel's NAME so that the rocprofv3 rows of
// rounds 4 and 5 compare) | 4096 (fp32 outputs stored nt: K >= 2048) | 16384 (contraction-major operands, "TN", below)
template <int EPI, int ACT, int FORM>
__global__ void __launch_bounds__(512)
gemm_bf16_nt_256p_kernel(GemmBf16 p, int tiles_m, int tiles_n, int m_total) {
    static_assert((FORM & ~(4096 | 16384)) == 8192, "gemm_bf16_nt_256p_kernel: unknown form");
    // 160 KiB: A ring 3 x 32 KiB | B ring 2 x 32 KiB (the B slot a tile's last K-step frees doubles as the
    // epilogue staging buffer, 4 KiB per wave)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int ntiles = tiles_m * tiles_n;
    constexpr bool OUT_F32 = (EPI == EPI_F32_RESID || EPI == EPI_F32);
    // Role split inside a K-step.  A `buffer_load ... lds` holds its wave until the CU's texture queue takes it (~15 cycles per
    // 1-KB piece, queue shared by all 8 waves), and a waiting wave issues no MFMAs.  With every wave requesting 4 A + 4 B
    // pieces right after the barrier, both waves of each SIMD sat in that queue together: ~890 cycles per K-step with
    // the matrix pipe idle (per-wave s_memtime sums, profiles/r02_gemm_gemm_waits*.log).  Here one half of the
    // workgroup requests ALL of B right after the barrier and the other half ALL of A at the END of its step, so on
    // every SIMD one wave is in its MFMAs while its partner is in the queue.  The B half is waves 4-7: the
    // second-dispatched wave of a SIMD loses MFMA arbitration to its partner anyway, so its early stall costs least
    // (the other assignment measured +1-2 % against +4-6 %).
    // MFMA shape (round 4): the wave tile (128 x 64) is 8 x 4 tiles of v_mfma_f32_16x16x32_bf16 (rounds 1-3: 4 x 2 of
    // v_mfma_f32_32x32x16_bf16) - same FLOPs, LDS fragment bytes and accumulator count per K-step, half the accumulator traffic per
    // FLOP inside the matrix pipe.  A register-only probe of the two instruction streams under the socket power cap
    // (profiles/r04_mfma_shape_power.log) holds 2 169 against 1 919 TFLOP/s (2.17 vs 1.93 GHz at ~1.33 kW).  A K-step is
    // 2 slices of 32 x 2 halves of 4 m-tiles = 4 phases of 16 MFMAs; accumulator layout:
    // tile (mt, nt), lane (i16 = lane & 15, G = lane >> 4) holds row m = 16 mt + i16, columns n = 16 nt + 4 G + {0..3}.
    // Operand layout (round 4, the weight-gradient GEMM of the training step).  FORM & 16384 ("TN"): both operands are stored
    // CONTRACTION-major - A = dY[k][m] (ld = lda), Bw = X[k][n] (ld = ldb), k = token - and C[m][n] = sum_k A[k][m] Bw[k][n] is
    // computed from them as they lie: a stage's LDS image is [64 k][256 rows] (512-B k-rows, the 32-B pairs of a k-row XORed with
    // f(k) = (k & 3) | ((k >> 3) & 1) << 2), filled by the same 16-B-per-lane DMA (one instruction = 2 k-rows x 512 B, fully
    // coalesced) and read through ds_read_b64_tr_b16 (2 per fragment: a 16-lane group gathers [4 k][16 rows] -> lane = row, 4 k
    // each), so the token-chunk transposes the NT form needs in front (9 ms per training step) are gone.  Batched form: batch b =
    // k-rows [b K, (b+1) K) (zero beyond k_rows: buffer range check), output rows [b batch_m_rows, ...).  
    constexpr bool TN = (FORM & 16384) != 0;
    constexpr bool NT_STORE = (FORM & 4096) != 0;
    constexpr int NPIECE = 8;      // DMA pieces per wave and operand half
    // Fragment reads go out BETWEEN the MFMAs of the previous phase (one ds_read_b128 behind each of its first MFMAs): with the
    // role split each wave runs alone on its SIMD while its partner sits in the texture queue or at the barrier, and every
    // non-MFMA issue slot between groups is then pipe idle time.

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    // Desynchronised halves (RVLM_GEMM_STRIP_FIRST, default 130 = the workgroups of every second XCD; 0 = off): those
    // workgroups compute their remainder-row unit BEFORE their tiles instead of after them.  No extra work, but they then
    // run a strip-time (4-14 us) behind the others for the whole launch, so the chip-wide store bursts - every workgroup's
    // epilogue at the same moment, 67-134 MB at once - and the final drain fall into two groups.  Same-box A/B, four
    // alternations (profiles/r03_ab_strip_first_4reps.log): 259.7 -> 264.5 img/s, all-GEMM 1 043 -> 1 075 TFLOP/s; fc2 dgrad
    // 65.7 -> 60.7 ms, fc2 forward 60.2 -> 57.2; every second workgroup of an XCD instead: 262.5; one XCD of eight: 261.4
    const int sf_mod = (p.stagger >> 16) & 127;
    const int sf_id = ((p.stagger >> 16) & 128) ? ((int)blockIdx.x & 7) : ((int)blockIdx.x >> 3);   // +128: whole XCDs instead
    const bool strip_first = sf_mod >= 2 && (sf_id % sf_mod) == sf_mod - 1 && m_total > p.M;
    if (strip_first) {
        strip_tail<EPI, ACT>(p, p.M, m_total, lds, w, lane);
        __syncthreads();      // the reduce buffer overlaps the first ring slots
    }
    const int lda = (int)p.lda, ldb = (int)p.ldb, ldo = (int)p.ldo;   // byte offsets fit 31 bits (host check)

    // buffer descriptors (wave-uniform, built from kernel arguments only): every global access below is
    // descriptor + 32-bit lane offset (VGPR, constant for the whole kernel) + 32-bit scalar offset
    const int tn_cols = p.batch_m_rows > 0 ? p.batch_m_rows : p.M;        // TN: columns of A = output rows per batch
    const auto a_rs = make_rsrc(p.A, TN ? (unsigned)((p.k_rows - 1) * lda + tn_cols) * 2u : (unsigned)((p.M - 1) * lda + p.K) * 2u);
    // batched form (batch_m_rows > 0, the split-K weight-gradient GEMM): rows [b*batch_m_rows, (b+1)*batch_m_rows) of A
    // meet rows [b*N, (b+1)*N) of Bw; the output keeps A's row index (a stack of per-batch [batch_m_rows, N] slabs)
    const int nbatch = p.batch_m_rows > 0 ? p.M / p.batch_m_rows : 1;
    const auto b_rs = make_rsrc(p.Bw, TN ? (unsigned)((p.k_rows - 1) * ldb + p.N) * 2u : (unsigned)((nbatch * p.N - 1) * ldb + p.K) * 2u);
    auto b_row0 = [&](int m0, int n0) { return p.batch_m_rows > 0 ? n0 + (m0 / p.batch_m_rows) * p.N : n0; };
    const unsigned out_elems = (unsigned)((p.M - 1) * ldo + p.N);
    const auto o_rs = make_rsrc(p.out, out_elems * (OUT_F32 ? 4u : 2u));
    const auto pre_rs = make_rsrc(EPI == EPI_BF16_ACT ? (const void*)p.out_pre : (const void*)p.out, out_elems * 2u);
    const auto h_rs = make_rsrc(EPI == EPI_BF16_DACT ? (const void*)p.h_pre : (const void*)p.out, out_elems * 2u);
    const auto bias_rs = make_rsrc(p.bias ? (const void*)p.bias : (const void*)p.out, p.bias ? (unsigned)p.N * 4u : 0u);
    const auto r_rs = make_rsrc(EPI == EPI_F32_RESID ? (const void*)p.residual : (const void*)p.out, out_elems * 4u);

    auto tile_origin = [&](int tile, int& m0, int& n0) {
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = tile & 7, loc = tile >> 3;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        const int group_size = p.group_m * tiles_n;
        const int first_m = (t / group_size) * p.group_m;
        const int gm = min(tiles_m - first_m, p.group_m);
        m0 = __builtin_amdgcn_readfirstlane((first_m + (t % group_size) % gm) * P_M);
        n0 = __builtin_amdgcn_readfirstlane(((t % group_size) / gm) * P_N);
    };
    const int nk = p.K / P_K;
    const int ntw = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup

    // ---- operand streams.  Stage g (global K-step counter over all tiles of this workgroup) has its A half in A
    // slot g % 3 and its B half in B slot g % 2; A runs three stages ahead of the MFMAs, B two.  One DMA instruction
    // moves 8 rows x 128 B.  Waves 4-7 request the B half (wave w rows [64(w&3), +64) = 8 pieces), waves
    // 0-3 the A half likewise.  Row of piece j:
    // first row + 8j + (lane>>3); its 16-B chunk (lane&7) holds logical chunk (lane&7) ^ ((row>>1)&7), which only
    // depends on the parity of j.
    int a_loff[2], b_loff[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int clog = (lane & 7) ^ (((jp * 8 + (lane >> 3)) >> 1) & 7);
        a_loff[jp] = ((lane >> 3) * lda + clog * 8) * 2;
        b_loff[jp] = ((lane >> 3) * ldb + clog * 8) * 2;
    }
    const int wrow = (w & 3) * 64;     // first row (of each 256-row half) this wave requests
    // TN: wave (w & 3) = 2 b + r1 requests k-rows 32 a + 16 a' + 8 b + 4 c + 2 r1 + {0, 1} (piece j = 4 a + 2 a' + c; lanes 0-31 the
    // even row, 32-63 the odd one), so f(k) = (lane >> 5) | (w & 3) << 1 is one constant per lane: slot chunk z = lane & 31 of a
    // k-row holds logical 16-B chunk (((z >> 1) ^ f) << 1) | (z & 1) (the XOR stays inside the 256-B bank row: f < 8)
    const int tn_kw = 8 * ((w & 3) >> 1) + 2 * (w & 1);
    if (TN) {
        const int z = lane & 31, f = (lane >> 5) | ((w & 3) << 1);
        const int clog = (((z >> 1) ^ f) << 1) | (z & 1);
        a_loff[0] = a_loff[1] = ((lane >> 5) * lda + clog * 8) * 2;
        b_loff[0] = b_loff[1] = ((lane >> 5) * ldb + clog * 8) * 2;
    }
    const int stage_wave_off = TN ? tn_kw * 512 : wrow * 128;
    // (the helpers below are macros: as lambdas called from the role lambdas they made hipcc drop every instantiation's host stub)
    // byte offset of a tile's operand block, this wave's share folded in (NT: row block; TN: column block + the batch's k-rows)
#define RVLM_A_TILE_OFF(m0) (TN ? (((m0) % tn_cols) + (((m0) / tn_cols) * p.K + tn_kw) * lda) * 2 : ((m0) + wrow) * lda * 2)
#define RVLM_B_TILE_OFF(m0, n0) (TN ? ((n0) + (((m0) / tn_cols) * p.K + tn_kw) * ldb) * 2 : (b_row0(m0, n0) + wrow) * ldb * 2)
    // piece j of a stage: LDS offset inside the wave's share / global offset (K-step kt); TN k-row of piece j: 32 a + 16 a' + 4 c
#define RVLM_PIECE_KROW(j) (32 * ((j) >> 2) + 16 * (((j) >> 1) & 1) + 4 * ((j) & 1))
#define RVLM_PIECE_LDS(j) (TN ? RVLM_PIECE_KROW(j) * 512 : (j) * 1024)
#define RVLM_PIECE_GOFF(j, kt, ld) (TN ? ((kt) * P_K + RVLM_PIECE_KROW(j)) * (ld) * 2 : (kt) * (P_K * 2) + (j) * 16 * (ld))
    // cursors: next stage to request = K-step a_kt of this workgroup's tile number a_ti (likewise b_*)
    int a_ti = 0, a_kt = 0, a_soff = 0, a_slot = 0;
    int b_ti = 0, b_kt = 0, b_soff = 0, b_slot = 0;
    {
        int m0, n0;
        tile_origin(blockIdx.x, m0, n0);
        a_soff = RVLM_A_TILE_OFF(m0);
        b_soff = RVLM_B_TILE_OFF(m0, n0);
    }
    // The tile loop exists twice, once per role (ROLE 1 = A half, 2 = B half): a wave picks its copy once, so inside the loop
    // the role is a compile-time fact and costs no branch.
    auto run = [&](auto role_c) __attribute__((always_inline)) {
    constexpr int ROLE = decltype(role_c)::value;
    constexpr bool own_a = ROLE != 2, own_b = ROLE != 1;
    // one DMA piece (8 rows x 128 B per wave instruction) of the next A / B stage, and the cursor step behind its last
    auto a_piece = [&](int j) __attribute__((always_inline)) {
        if (!own_a || a_ti >= ntw) return;
        __attribute__((address_space(3))) char* dst =
            (__attribute__((address_space(3))) char*)lds + (a_slot * PA_SLOT + stage_wave_off);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            a_rs, (lds_ptr_t)(dst + RVLM_PIECE_LDS(j)), 16, a_loff[j & 1],
            __builtin_amdgcn_readfirstlane(a_soff + RVLM_PIECE_GOFF(j, a_kt, lda)), 0, 0);
    };
    auto a_advance = [&]() __attribute__((always_inline)) -> bool {
        if (a_ti >= ntw) return false;
        a_slot = (a_slot == 2) ? 0 : a_slot + 1;
        if (++a_kt == nk) {
            a_kt = 0;
            if (++a_ti < ntw) {
                int m0, n0;
                tile_origin(blockIdx.x + a_ti * gridDim.x, m0, n0);
                a_soff = RVLM_A_TILE_OFF(m0);
            }
        }
        return true;
    };
    auto b_piece = [&](int j) __attribute__((always_inline)) {
        if (!own_b || b_ti >= ntw) return;
        __attribute__((address_space(3))) char* dst =
            (__attribute__((address_space(3))) char*)lds + (PB_BASE + b_slot * PB_SLOT + stage_wave_off);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            b_rs, (lds_ptr_t)(dst + RVLM_PIECE_LDS(j)), 16, b_loff[j & 1],
            __builtin_amdgcn_readfirstlane(b_soff + RVLM_PIECE_GOFF(j, b_kt, ldb)), 0, 0);
    };
    auto b_advance = [&]() __attribute__((always_inline)) -> bool {
        if (b_ti >= ntw) return false;
        b_slot ^= 1;
        if (++b_kt == nk) {
            b_kt = 0;
            if (++b_ti < ntw) {
                int m0, n0;
                tile_origin(blockIdx.x + b_ti * gridDim.x, m0, n0);
                b_soff = RVLM_B_TILE_OFF(m0, n0);
            }
        }
        return true;
    };
    auto issue_a = [&]() -> bool {     // a whole stage in one burst
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) a_piece(j);
        return a_advance();
    };
    auto issue_b = [&]() -> bool {
#pragma unroll
        for (int j = 0; j < NPIECE; ++j) b_piece(j);
        return b_advance();
    };

    // ---- fragment addresses = per-lane part (fa16, by k-slice) + ring slot offset.  The slot offsets are kept
    // opaque to the optimiser: otherwise it precomputes every (slot, slice, operand) sum into VGPRs that stay live over the
    // whole tile loop (and spill around it).
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int ab_delta = PB_BASE + (wn * 64 - wm * 128) * 128;   // B row block of this wave relative to its A rows
    int ca_slot = 0, cb_slot = 0;                                 // slots of the stage being consumed

    // lane (i16, G) reads row (row block) + 16 t + i16, logical 16-B chunk 4 ks + G of the 32-deep slice ks; the ring's
    // swizzle key (row >> 1) & 7 only depends on i16.  (Conflict-free under ds_read_b128's 4 x 16 lane groups: every group sees
    // each (row parity, chunk) pair once - checked by hand for the four groups.)
    const int i16 = lane & 15, G = lane >> 4;
    unsigned fa16[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fa16[ks] = lds_base + (wm * 128 + i16) * 128 + (((ks * 4 + G) ^ ((i16 >> 1) & 7)) << 4);
    f32x4 acc16[8][4];
    // TN: lane (i16, G) of a fragment of slice ks holds row (tile) + i16, k = 32 ks + 8 G + 0..7 = two transposing reads (k-rows
    // 32 ks + 8 G + 4 hh + (i16 >> 2), hh = 0 / 1: +2048 B), each lane pointing at the 8 bytes (i16 & 3) of its k-row's 32-B
    // pair p ^ f(k), p = the tile's pair index (A: 8 wm + t, B: 4 wn + nt), f(k) = (i16 >> 2) | (G & 1) << 2 - independent of
    // ks / hh / tile, so tile t is one v_xor of (t << 5) into the per-lane base and (ks, hh) are immediates.  A 32-lane half
    // (two groups G) reads 8 k-rows x 32 B at 8 different pair positions of the 256-B bank row: conflict-free.
    const unsigned tr_lane = (8 * G + (i16 >> 2)) * 512 + (i16 & 3) * 8 + (((i16 >> 2) | ((G & 1) << 2)) << 5);
    const unsigned trA = lds_base + (tr_lane ^ (wm * 256));
    const unsigned trB = lds_base + PB_BASE + (tr_lane ^ (wn * 128));
    // X fragments (activation rows) of m-tiles 4 h .. 4 h + 3 of slice ks
    auto load_x16 = [&](int sa, int ks, int h, i32x4 (&x)[4]) __attribute__((always_inline)) {
        int oa = sa * PA_SLOT;
        asm volatile("" : "+s"(oa));
        if (TN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tr_frag(trA + oa, 4 * h + q, ks, x[q]);
            return;
        }
        const unsigned aa = fa16[ks] + oa + h * 8192;
        asm volatile("ds_read_b128 %0, %1" : "=v"(x[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(x[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(x[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(x[3]) : "v"(aa));
    };
    // W fragments (weight rows) of the wave's 4 n-tiles of slice ks
    auto load_w16 = [&](int sb, int ks, i32x4 (&wf)[4]) __attribute__((always_inline)) {
        int ob = sb * PB_SLOT + (TN ? 0 : ab_delta);
        asm volatile("" : "+s"(ob));
        if (TN) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) tr_frag(trB + ob, nt, ks, wf[nt]);
            return;
        }
        const unsigned bb = fa16[ks] + ob;
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(wf[1]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(wf[2]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(wf[3]) : "v"(bb));
    };
    // one phase: the 16 MFMAs of m-tiles 4 h .. 4 h + 3 x 4 n-tiles; with LOADS, the next phase's fragment reads go out one
    // behind each of the first MFMAs (nks, nh: the next phase; its W fragments only when it opens a new slice)
    auto mma16_phase = [&](const i32x4 (&x)[4], const i32x4 (&wf)[4], int h, bool loads, int sa, int sb, int nks, int nh,
                           i32x4 (&nx)[4], i32x4 (&nw)[4]) __attribute__((always_inline)) {
        int oa = sa * PA_SLOT, ob = sb * PB_SLOT + (TN ? 0 : ab_delta);
        asm volatile("" : "+s"(oa), "+s"(ob));
        const unsigned aa = TN ? trA + oa : fa16[nks] + oa + nh * 8192, bb = TN ? trB + ob : fa16[nks] + ob;
        // (measured and not kept, profiles/r04_ab_mfma16_*.log: two reads per slot, a read behind every second MFMA, n-tile-outer order)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc16[4 * h + q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, x[q]), acc16[4 * h + q][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (loads) {
                    // fragment read r of the next phase (X: 0..3, W: 4..7) goes out behind MFMA r of this one
                    const int r = q * 4 + nt;
                    if (TN) {            // both transposing reads of a fragment behind one MFMA
                        if (r < 4) tr_frag(aa, 4 * nh + r, nks, nx[r]);
                        else if (nh == 0 && r < 8) tr_frag(bb, r - 4, nks, nw[r - 4]);
                    } else
                    if (r == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(nx[0]) : "v"(aa));
                    else if (r == 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(nx[1]) : "v"(aa));
                    else if (r == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(nx[2]) : "v"(aa));
                    else if (r == 3) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(nx[3]) : "v"(aa));
                    else if (nh == 0 && r == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(nw[0]) : "v"(bb));
                    else if (nh == 0 && r == 5) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(nw[1]) : "v"(bb));
                    else if (nh == 0 && r == 6) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(nw[2]) : "v"(bb));
                    else if (nh == 0 && r == 7) asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(nw[3]) : "v"(bb));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    // zero the accumulators of the 32 x 32 sub-tile (mi, ni) of the wave tile
    auto init_acc = [&](int mi, int ni) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) acc16[2 * mi + (pc >> 1)][2 * ni + (pc & 1)] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { init_acc(mi, 0); init_acc(mi, 1); }

    // ---- prologue: stages 0 and 1 complete, A of stage 2 ----
    issue_a(); issue_b();
    issue_a(); issue_b();
    bool a_ahead = issue_a();       // was the A half of stage g+2 requested at the previous barrier?
    // (B half: 16 pieces, all needed; A half: 24, the 8 of stage 2 may stay in flight)
    if (own_a && a_ahead) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    i32x4 a0[4], a1[4];          // X fragments of the current / next phase
    i32x4 w0[4], w1[4];          // W fragments of the current / next 32-deep slice

    auto stamp = [&](int ti, int k) {   // optional per-tile timeline (test hook rvlm_k_gemm_set_trace), wave 0 only
        if (p.trace && w == 0 && ti < 7) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.trace[((long)blockIdx.x * 8 + ti) * 4 + k] = t;
        }
    };
    if (p.trace && w == 0 && lane == 0) {   // clock calibration: s_memtime vs the constant 100 MHz s_memrealtime
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 0] = __builtin_amdgcn_s_memtime();
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }

    for (int ti = 0; ti < ntw; ++ti) {
        int m0, n0;
        tile_origin(blockIdx.x + ti * gridDim.x, m0, n0);
        const bool last_tile = (ti + 1 == ntw);
        stamp(ti, 0);
        load_x16(ca_slot, 0, 0, a0); load_w16(cb_slot, 0, w0);

        // One K-step.  At its barrier the next stage has landed for every wave and every wave is done reading this
        // stage, whose two slots are refilled right away: B of stage g+2, then A of stage g+3 (in that order: the
        // next step may leave exactly the youngest pieces, the A half, in flight).
        auto k_step = [&](bool first_of_tile, bool last_of_tile) {
            const int na_slot = (ca_slot == 2) ? 0 : ca_slot + 1, nb_slot = cb_slot ^ 1;
            // phases 0 - 2 of the K-step (slice 0 half 0, slice 0 half 1, slice 1 half 0); phase 3 runs behind the barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma16_phase(a0, w0, 0, true, ca_slot, cb_slot, 0, 1, a1, w1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma16_phase(a1, w0, 1, true, ca_slot, cb_slot, 1, 0, a0, w1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma16_phase(a0, w1, 0, true, ca_slot, cb_slot, 1, 1, a1, w0);
            // the A half requests the stage for the slot freed at the PREVIOUS barrier (nothing is free yet in the
            // very first step); its 8 youngest pieces may stay in flight, the B half waits for all of its own
            a_ahead = (first_of_tile && ti == 0) ? false : issue_a();
            if (own_a && a_ahead) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (first_of_tile) stamp(ti, 1);
            // (B is deferred past the epilogue in a tile's last step: staging uses that slot; the next stage's first fragments
            // go out among the caller's MFMAs)
            if (!last_of_tile) issue_b();
            ca_slot = na_slot;
            cb_slot = nb_slot;
        };
        k_step(true, false);
        __builtin_amdgcn_sched_barrier(0);
        // + the next stage's first fragments (its barrier is behind us)
        mma16_phase(a1, w1, 1, true, ca_slot, cb_slot, 0, 0, a0, w0);
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 1; kt < nk - 1; ++kt) {
            k_step(false, false);
            __builtin_amdgcn_sched_barrier(0);
            mma16_phase(a1, w1, 1, true, ca_slot, cb_slot, 0, 0, a0, w0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int stage_slot = cb_slot;   // B slot of the tile's last stage = epilogue staging area after its barrier
        k_step(false, true);

        // Lane constants of the epilogue, re-derived PER TILE from an opaque copy of the lane id (round 4): as kernel-scope
        // constants hipcc kept ~15 of them (and sums it precomputes from them) live through the whole mainloop, where the
        // 16x16x32 form has no register to spare - they were spilled and reloaded from scratch in every tile's epilogue, behind
        // the stores (VMEM returns in order).  ~20 VALU per tile instead.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int i16 = lane_e & 15, G = lane_e >> 4;
        // lane offsets of the epilogue's coalesced accesses (bytes); one instruction = 8 rows (16 for the 64-B rows)
        const int st16_loff = ((lane_e >> 3) * ldo + (lane_e & 7) * 8) * 2;   // bf16: 128 B (64 columns) per row
        const int st32_loff = ((lane_e >> 3) * ldo + (lane_e & 7) * 4) * 4;   // fp32: 128 B (32 columns) per row
        const int h16_loff = ((lane_e >> 2) * ldo + (lane_e & 3) * 8) * 2;    // bf16 32-column sub-tile: 16 rows x 64 B per instruction
        const int r0 = lane_e >> 3;                                           // flush: row r0 + 8*it, 16-B slot lane & 7

        float4 bv[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // (a null bias has a zero-length descriptor: out-of-range loads return 0)
                if (g & 1) continue;            // bv[ni][g], g even = the 4 columns of n-tile 2 ni + (g >> 1); 4 vectors, not 8
                bv[ni][g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                    bias_rs, G * 16, __builtin_amdgcn_readfirstlane((n0 + wn * 64 + (2 * ni + (g >> 1)) * 16) * 4), 0));
            }
        // Side input of the epilogue (fp32 residual / bf16 h_pre), read in the SAME coalesced pattern as the output is
        // stored (after the LDS transpose) and prefetched SIDE_DEPTH 32x32 sub-tiles ahead: sub-tile s = 2*mi + ni lives
        // in side[s % SIDE_DEPTH].  The first ones are requested under the last MFMAs of the tile.
        const int m_base = m0 + wm * 128, n_base = n0 + wn * 64;
        // (2 slots.  Before the tile's last phase accumulated in place hipcc parked five of the
        // twelve side loads of 4 slots in scratch, each behind its own s_waitcnt vmcnt(0) - an HBM round trip apiece, per tile;
        // with the in-place phase nothing spills at any depth and 2 / 3 / 4 slots measure within 0.3 % of each other end to end
        // (278.0 / 277.5 / 277.4 img/s on one box, profiles/r04_ab_mfma16_inplace.log): the shallowest one ships)
        constexpr int SIDE_DEPTH = 2;
        u32x4 side[SIDE_DEPTH][4];
        auto load_side = [&](int sub) {
            if (EPI == EPI_F32_RESID) {    // sub = 2*mi + ni: 32 rows x 128 B, 4 loads of 8 rows
                const int mi = sub >> 1, ni = sub & 1;
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 4);
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    side[sub % SIDE_DEPTH][it] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, st32_loff, so + it * 32 * ldo, 0);
            } else {                        // sub = mi: act'(h) next to the 32 x 64 bf16 block, 4 loads of 8 rows x 128 B
                const int so = __builtin_amdgcn_readfirstlane(((m_base + sub * 32) * ldo + n_base) * 2);
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    side[sub % SIDE_DEPTH][it] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, st16_loff, so + it * 16 * ldo, 0);
            }
        };
        if (EPI == EPI_F32_RESID) load_side(0);                            // a0/b0 are dead
        if (EPI == EPI_BF16_DACT) { load_side(0); load_side(1); }
        __builtin_amdgcn_sched_barrier(0);
        {
            // The tile's LAST phase (no fragment reads) sits in a loop of (opaque) trip count 1: the accumulators are then
            // loop-carried and the MFMAs accumulate in place, as in the K loop.  As straight-line code hipcc gave every result of
            // the phase fresh registers (D != C, 64 of them over the phase) right where the epilogue's side-input prefetch wants
            // its registers, and parked side loads - or accumulators, behind s_nop 7 each - in scratch.  (Inline-asm MFMAs with
            // "+v" do the same but hide the MFMA -> VALU wait states from the compiler: a spill it placed behind one read garbage.)
            int once = 1;
            asm volatile("" : "+s"(once));
#pragma unroll 1
            for (int rep = 0; rep < once; ++rep) mma16_phase(a1, w1, 1, false, 0, 0, 0, 0, a0, w0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // bias joins the accumulators here, so that its registers are free for the side-input prefetch
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 bq = bv[nt >> 1][(nt & 1) * 2];
                acc16[mt][nt] += f32x4{bq.x, bq.y, bq.z, bq.w};
            }
        __builtin_amdgcn_sched_barrier(0);
        if (EPI == EPI_F32_RESID) {
#pragma unroll
            for (int sd = 1; sd < SIDE_DEPTH; ++sd) load_side(sd);
        }
        if (EPI == EPI_BF16_DACT) {
#pragma unroll
            for (int sd = 2; sd < SIDE_DEPTH; ++sd) load_side(sd);
        }
        stamp(ti, 2);

        // ---- epilogue of (m0, n0): staging through the freed B slot, accumulators re-zeroed sub-tile by sub-tile ----
        const unsigned ebuf = lds_base + PB_BASE + stage_slot * PB_SLOT + w * P_EPI_WAVE;
        // staging write addresses: piece pc = (a, b) of a 32 x 32 sub-tile is accumulator tile (2 mi + a, 2 ni + b), row 16 a + i16
        // (128-B rows), columns 16 b + 4 G .. + 3, i.e. 4-column chunk 4 b + G; chunk index = (constant per piece) ^ (lane part),
        // keys of row & 15 = i16.  (The staged layouts are those of the 32x32x16 form of rounds 1-3: element (row, col) of the
        // 32-row block lands at the same LDS address, so the read-back and store side below did not change with the MFMA shape.)
        const unsigned w16n_pre = ebuf + i16 * 128 + ((G ^ i16) << 3);                // bf16: 8-B chunk ^ (row & 15)
        const unsigned wpn_pre = ebuf + i16 * 128 + ((G ^ pair_key(i16)) << 3);       // activation pair: see pair_key()
        const unsigned w32n_pre = ebuf + i16 * 128 + ((G ^ (i16 & 7)) << 4);          // fp32: 16-B chunk ^ (row & 7)
        // piece pc of sub-tile (mi, ni): its 4 values and its staging addresses
        auto piece = [&](int mi, int ni, int pc) -> f32x4 { return acc16[2 * mi + (pc >> 1)][2 * ni + (pc & 1)]; };
        auto w16_addr = [&](int ni, int pc) -> unsigned {      // bf16, 64-column rows
            return (w16n_pre ^ ((ni * 8 + 4 * (pc & 1)) << 3)) + (pc >> 1) * 2048;
        };
        auto wp_addr = [&](int which, int pc) -> unsigned {    // activation pair: which = 0 act', 1 act
            return (wpn_pre ^ ((which * 8 + 4 * (pc & 1)) << 3)) + (pc >> 1) * 2048;
        };
        auto w32_addr = [&](int pc) -> unsigned {              // fp32, 32-column rows
            return (w32n_pre ^ ((pc & 1) << 6)) + (pc >> 1) * 2048;
        };
        const unsigned r16_a = ebuf + r0 * 128 + (((lane_e & 7) ^ (r0 >> 1)) << 4);       // bf16, it even
        const unsigned r16_b = ebuf + r0 * 128 + (((lane_e & 7) ^ (r0 >> 1) ^ 4) << 4);   // bf16, it odd ((row & 15) >> 1 flips bit 2)
        const unsigned r32 = ebuf + r0 * 128 + (((lane_e & 7) ^ r0) << 4);                // fp32 ((r0 + 8*it) & 7 == r0)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (!OUT_F32) {
                // bf16 output(s): stage 32 rows x 64 columns (128-B rows, 8-B chunk index XOR (row & 15)), then 4
                // stores of 8 full 128-B rows each.
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base) * 2);
                auto stage_flush = [&](__amdgpu_buffer_rsrc_t rs, int what) {   // what: 0 value, 1 act(value), 2 act'(value), 3 value * side
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = piece(mi, ni, g);
                            bf16x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float av, dv;
                                actp_pair<ACT>(v[e], av, dv);
                                o[e] = (bf16_t)(what == 1 ? av : what == 2 ? dv : v[e]);
                            }
                            lds_w64(w16_addr(ni, g), __builtin_bit_cast(u32x2, o));
                        }
                    u32x4 t0 = lds_r128<0>(r16_a), t1 = lds_r128<8 * 128>(r16_b), t2 = lds_r128<16 * 128>(r16_a),
                          t3 = lds_r128<24 * 128>(r16_b);
                    lds_wait();
                    if (r0 & 1) {   // a lane's 16 B cover two 8-B chunks, swapped when its row is odd
                        t0 = __builtin_shufflevector(t0, t0, 2, 3, 0, 1); t1 = __builtin_shufflevector(t1, t1, 2, 3, 0, 1);
                        t2 = __builtin_shufflevector(t2, t2, 2, 3, 0, 1); t3 = __builtin_shufflevector(t3, t3, 2, 3, 0, 1);
                    }
                    if (what == 3) {   // dgrad through the activation: times the act'(h) the forward stored (bf16 x bf16 in fp32)
                        auto mul8 = [&](u32x4& t, const u32x4& hq) {
                            const bf16x8 a = __builtin_bit_cast(bf16x8, t), b = __builtin_bit_cast(bf16x8, hq);
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)a[e] * (float)b[e]);
                            t = __builtin_bit_cast(u32x4, o);
                        };
                        mul8(t0, side[mi % SIDE_DEPTH][0]); mul8(t1, side[mi % SIDE_DEPTH][1]);
                        mul8(t2, side[mi % SIDE_DEPTH][2]); mul8(t3, side[mi % SIDE_DEPTH][3]);
                    }
                    if (what == 2) {     // act'(h): next read by the backward pass - streamed past the L2
                        store16<2>(t0, rs, st16_loff, so);
                        store16<2>(t1, rs, st16_loff, so + 16 * ldo);
                        store16<2>(t2, rs, st16_loff, so + 32 * ldo);
                        store16<2>(t3, rs, st16_loff, so + 48 * ldo);
                    } else {
                        store16(t0, rs, st16_loff, so);
                        store16(t1, rs, st16_loff, so + 16 * ldo);
                        store16(t2, rs, st16_loff, so + 32 * ldo);
                        store16(t3, rs, st16_loff, so + 48 * ldo);
                    }
                };
                if (EPI == EPI_BF16_ACT && p.out_pre) {
                    // Activation PAIR: act(h) and act'(h) of one 32 x 32 sub-tile at a time, both staged at once (act'(h) in
                    // the left 64 B of the staged rows, act(h) in the right) and flushed as 16 rows x 64 B per store.  The
                    // round-2 form flushed the 32 x 64 block twice (act' then act): hipcc kept the first pass's act values live
                    // across the LDS round trip and the stores of the second, ran out of registers (20 spilled dwords) and
                    // reloaded them from scratch BEHIND the four stores it had just issued - VMEM returns in order, so every
                    // reload waited for those stores to be acknowledged, six times per tile.
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 oa, od;
                            const f32x4 hv = piece(mi, ni, g);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float av, dv;
                                actp_pair<ACT>(hv[e], av, dv);
                                oa[e] = (bf16_t)av; od[e] = (bf16_t)dv;
                            }
                            // 8-B chunk index within the 128-B row: (which * 8 + 2 g + hi) ^ pair_key(row); which = 0 act', 1 act
                            lds_w64(wp_addr(0, g), __builtin_bit_cast(u32x2, od));
                            lds_w64(wp_addr(1, g), __builtin_bit_cast(u32x2, oa));
                        }
                        init_acc(mi, ni);
                        // read back: 16 rows per instruction, 4 lanes x 16 B per row and output; lane -> row rr (+ 16), 16-B
                        // slot q of output `which`: 8-B chunks (which * 8 + 2 q, + 1) ^ (row & 15)
                        const int rr = lane_e >> 2, q = lane_e & 3;
                        u32x2 rq[8];
#pragma unroll
                        for (int which = 0; which < 2; ++which)
#pragma unroll
                            for (int half = 0; half < 2; ++half) {
                                const int row = rr + 16 * half;
                                const unsigned rbase = ebuf + row * 128;
                                const int c8 = (which * 8 + 2 * q) ^ pair_key(row);
                                asm volatile("ds_read_b64 %0, %1" : "=v"(rq[(which * 2 + half) * 2]) : "v"(rbase + (c8 << 3)) : "memory");
                                asm volatile("ds_read_b64 %0, %1" : "=v"(rq[(which * 2 + half) * 2 + 1]) : "v"(rbase + ((c8 ^ 1) << 3)) : "memory");
                            }
                        lds_wait();
                        auto join = [](u32x2 lo, u32x2 hi2) { u32x4 t; t.x = lo.x; t.y = lo.y; t.z = hi2.x; t.w = hi2.y; return t; };
                        const int so2 = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 2);
                        store16<2>(join(rq[0], rq[1]), pre_rs, h16_loff, so2);              // act'(h): next read by the backward pass
                        store16<2>(join(rq[2], rq[3]), pre_rs, h16_loff, so2 + 32 * ldo);
                        store16(join(rq[4], rq[5]), o_rs, h16_loff, so2);
                        store16(join(rq[6], rq[7]), o_rs, h16_loff, so2 + 32 * ldo);
                    }
                } else {
                    // (forward-only callers - no backward to come - pass out_pre = null: act'(h) is then not written)
                    stage_flush(o_rs, EPI == EPI_BF16_ACT ? 1 : EPI == EPI_BF16_DACT ? 3 : 0);
                    if (EPI == EPI_BF16_DACT && mi + SIDE_DEPTH < 4) load_side(mi + SIDE_DEPTH);   // (only with fewer than 4 slots)
                    init_acc(mi, 0);
                    init_acc(mi, 1);
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    // fp32 staging of one 32x32 sub-tile: 128-B rows, 16-B chunk index XOR (row & 7)
                    const int sub = mi * 2 + ni;
#pragma unroll
                    for (int g = 0; g < 4; ++g) lds_w128(w32_addr(g), __builtin_bit_cast(u32x4, piece(mi, ni, g)));
                    init_acc(mi, ni);
                    {
                        const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 4);
                        u32x4 t[4] = {lds_r128<0>(r32), lds_r128<8 * 128>(r32), lds_r128<16 * 128>(r32),
                                      lds_r128<24 * 128>(r32)};
                        lds_wait();
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            if (EPI == EPI_F32_RESID) {
                                const float4 x = __builtin_bit_cast(float4, t[it]);
                                const float4 r = __builtin_bit_cast(float4, side[sub % SIDE_DEPTH][it]);
                                t[it] = __builtin_bit_cast(u32x4, make_float4(x.x + r.x, x.y + r.y, x.z + r.z, x.w + r.w));
                            }
                            if (NT_STORE) store16<2>(t[it], o_rs, st32_loff, so + it * 32 * ldo);   // K >= 2048: large operand set
                            else store16(t[it], o_rs, st32_loff, so + it * 32 * ldo);
                        }
                    }
                    if (EPI == EPI_F32_RESID && sub + SIDE_DEPTH < 8) load_side(sub + SIDE_DEPTH);
                }
            }
        }
        stamp(ti, 3);
        if (!last_tile) {
            // every wave is done with the staging slot: request the B half that was held back, fetch the first fragments
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue_b();
        }
    }
    };
    if (w >= 4) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 1>{});
    if (m_total > p.M && !strip_first) strip_tail<EPI, ACT>(p, p.M, m_total, lds, w, lane);
    if (p.trace && w == 0 && lane == 0) {
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 2] = __builtin_amdgcn_s_memtime();
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

#undef RVLM_A_TILE_OFF
#undef RVLM_B_TILE_OFF
#undef RVLM_PIECE_KROW
#undef RVLM_PIECE_LDS
#undef RVLM_PIECE_GOFF

// (the test hooks of the A/B arms: no-ops here - the shipped library instantiates the measured form only)
int g_persist_ablate = 0;
void gemm_set_ablate(int) { g_persist_ablate = 0; }
bool gemm_has_ablate() { return false; }
unsigned long long* g_persist_trace = nullptr;   // [256 workgroups][8 tiles][4 stamps] or null
void gemm_set_trace(unsigned long long* ptr) { g_persist_trace = ptr; }
void gemm_set_m16(int) {}
bool gemm_has_m32() { return false; }

template <int EPI, int ACT, int FORM>
static int launch_256p_form(const GemmBf16& p, int tiles_m, int tiles_n, int m_total, hipStream_t s) {
    static unsigned long long attr_devices = 0;
    const int lds_bytes = 3 * PA_SLOT + 2 * PB_SLOT;
    RVLM_ONCE_PER_DEVICE(attr_devices, {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256p_kernel<EPI, ACT, FORM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
    });
    const int grid = std::min(tiles_m * tiles_n, 256);
    GemmBf16 q = p;
    q.trace = g_persist_trace;
    hipLaunchKernelGGL((gemm_bf16_nt_256p_kernel<EPI, ACT, FORM>), dim3(grid), dim3(512), lds_bytes, s, q, tiles_m, tiles_n, m_total);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template <int EPI, int ACT>
static int launch_256p_act(const GemmBf16& p, int tiles_m, int tiles_n, int m_total, hipStream_t s) {
    if constexpr (EPI == EPI_F32_RESID) {
        // fp32 output stored nt from K = 2048 on (profiles/r03_ab_nt_threshold.log)
        if (p.K >= 2048) return launch_256p_form<EPI, ACT, 4096 | 8192>(p, tiles_m, tiles_n, m_total, s);
    }
    return launch_256p_form<EPI, ACT, 8192>(p, tiles_m, tiles_n, m_total, s);
}
template <int EPI>
static int launch_256p(const GemmBf16& p, int tiles_m, int tiles_n, int m_total, hipStream_t s) {
    if constexpr (EPI == EPI_BF16_ACT || EPI == EPI_BF16_DACT) {
        if (p.act != RVLM_ACT_QUICK_GELU) return launch_256p_act<EPI, RVLM_ACT_GELU>(p, tiles_m, tiles_n, m_total, s);
    }
    return launch_256p_act<EPI, RVLM_ACT_QUICK_GELU>(p, tiles_m, tiles_n, m_total, s);
}

// rows [0, 256*floor(M/256)) of the problem; *rows_done = 0 when the shape does not qualify
int gemm_bf16_nt_256p(const GemmBf16& p, int* rows_done, hipStream_t s) {
    *rows_done = 0;
    if (p.M < P_M || p.N % P_N != 0 || p.K % (2 * P_K) != 0 || p.K < 2 * P_K) return RVLM_OK;
    // buffer descriptors address with 32-bit offsets
    const long lim = 1L << 31;
    if (p.tn) {
        // contraction-major operands (weight gradient): fp32 slabs only, whole 256-row batches, 16-B aligned k-rows
        if (p.epi != EPI_F32 || p.batch_m_rows <= 0 || p.batch_m_rows % P_M != 0 || p.M % p.batch_m_rows != 0 || p.k_rows <= 0 ||
            p.lda % 8 != 0 || p.ldb % 8 != 0)
            return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256p: contraction-major form needs epi F32, batch_m_rows % 256, k_rows, ld % 8");
        if ((long)p.k_rows * p.lda * 2 >= lim || (long)p.k_rows * p.ldb * 2 >= lim || (long)p.M * p.ldo * 4 >= lim ||
            (long)(p.M / p.batch_m_rows) * p.K * std::max(p.lda, p.ldb) * 2 >= lim) return RVLM_OK;
        GemmBf16 q = p;
        q.stagger = 0; q.wave_prio = 0; q.krot = 0; q.group_m = 4;
        int rc = launch_256p_form<EPI_F32, RVLM_ACT_QUICK_GELU, 8192 | 16384>(q, p.M / P_M, p.N / P_N, p.M, s);
        if (rc) return rc;
        *rows_done = p.M;
        return RVLM_OK;
    }
    if ((long)p.M * p.lda * 2 >= lim || (long)p.N * p.ldb * 2 >= lim || (long)p.M * p.ldo * 4 >= lim) return RVLM_OK;
    if (p.batch_m_rows > 0) {
        if (p.batch_m_rows % P_M != 0 || p.M % p.batch_m_rows != 0)
            return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256p: batch_m_rows must be a multiple of 256 dividing M");
        if ((long)(p.M / p.batch_m_rows) * p.N * p.ldb * 2 >= lim) return RVLM_OK;
    }
    GemmBf16 q = p;
    const int tiles_m = p.M / P_M, tiles_n = p.N / P_N;
    q.M = tiles_m * P_M;
    if (q.epi == EPI_F32_RESID && !q.residual) q.epi = EPI_F32;
    // the remainder rows ride along in the same launch (strip_tail) unless RVLM_GEMM_TAIL=0 / an ablation build is on
    static int tail_on = -1;
    if (tail_on < 0) { const char* e = getenv("RVLM_GEMM_TAIL"); tail_on = e ? atoi(e) : 1; }
    // the measured settings of the schedule (same-box A/Bs of round 3): the workgroups of every second
    // XCD take their remainder-row unit first (130), tile-order groups of 4 m-tiles
    q.stagger = (130 & 255) << 16; q.wave_prio = 0; q.krot = 0;
    q.group_m = 4;
    const bool tail = tail_on && p.batch_m_rows == 0 && p.M > q.M;
    const int m_total = tail ? p.M : q.M;
    int rc;
    switch (q.epi) {
        case EPI_BF16: rc = launch_256p<EPI_BF16>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_F32_RESID: rc = launch_256p<EPI_F32_RESID>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_BF16_ACT: rc = launch_256p<EPI_BF16_ACT>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_BF16_DACT: rc = launch_256p<EPI_BF16_DACT>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_F32: rc = launch_256p<EPI_F32>(q, tiles_m, tiles_n, m_total, s); break;
        default: return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256p: unknown epilogue");
    }
    if (rc) return rc;
    *rows_done = m_total;
    return RVLM_OK;
}

}  // namespace rvlm
