// bf16 MFMA GEMM for gfx950, PERSISTENT 256x256 tiles with the two wave groups of a workgroup ALTERNATING between MFMAs
// and everything else ("256x": ping-pong form of gemm_bf16_256p.hip).
//
// Why: the 256p kernel's mainloop is within 4 % of what two waves per SIMD and one
// barrier per K-step give, but its fused epilogues - 4 k (bf16) to 17 k (fp32 + residual) shader cycles beside 38 k of
// mainloop at K = 1024 - run with all 8 waves in lockstep, i.e. with the matrix pipe idle.
//
// Structure.  A workgroup is two groups of four waves, one wave of each group per SIMD; group X owns rows [128 X, +128) of
// every 256x256 tile of the workgroup's list (wave tile 128 x 64 as in 256p, 128 accumulators per lane).  Time is cut
// into steps of one 64-deep K slice, one s_barrier per step for all 8 waves.  In every step exactly ONE group issues
// MFMAs: group 0 runs its half of tile i for nk = K / 64 steps, then group 1 runs its half of tile i for nk steps,
// and so on.  The group that is not in MFMAs
//   * requests the operands of the stage two steps ahead (B: the 256 weight rows, A: the 128 rows of the computing
//     group) - the computing group issues nothing but MFMAs and LDS fragment reads, so no wave of the matrix pipe's
//     feeder ever waits in the CU's texture queue, and
//   * runs the epilogue of the half tile it has just finished, cut into E steps (bias, activation pair / x act' /
//     fp32 residual, wave-private LDS transpose, full-line stores), then idles (requests only) until its next turn.
// A lone wave issues an MFMA every ~39 cycles (profiles/r02_gemm_timeline_split_fine.log), two sharing a SIMD one every
// 37.5: the pipe is fed at the same rate, and the epilogue no longer costs pipe time.  The price: the weight panel is
// fetched once per HALF tile (operand bytes per FLOP x 1.5; they come from L2, the panel is shared by the XCD's
// workgroups), and 4 waves instead of 8 hide MFMA / LDS latency.
//   * LDS 160 KiB: 3 stages x (A 16 KiB + B 32 KiB), i.e. BOTH operands two steps ahead (the first form of this kernel kept
//     256p's rings - B one step ahead - and shared the stream between the groups: a request then had to land inside a
//     ~1 300-cycle step and every step took 2 700-3 100 cycles, profiles/r03_gemm_pingpong_v1_shared_stream.log) + 16 KiB of
//     epilogue staging (only one group is in its epilogue at a time).
//   * The K order of a tile is 0 .. nk-1 as in 256p and the MFMA order inside a slice is the same: results are
//     bit-identical to gemm_bf16_nt_256p_kernel.
//
// Requirements (else the dispatcher keeps 256p): N % 256 == 0, K % 128 == 0, K >= 512, no batched form.
#include "../kernels.h"
#include "../gemm_persist.h"
#include "../gemm_strip.h"
#include <type_traits>

namespace rvlm {

constexpr int X_A_BYTES = 128 * 128;                 // one stage of A: the computing group's 128 rows x 64 k
constexpr int X_STAGE = X_A_BYTES + P_OPER_BYTES;    // + 256 weight rows x 64 k = 48 KiB
constexpr int X_STAGING = 3 * X_STAGE;               // 4 waves x 4 KiB behind the ring

template <int EPI, int ACT, bool HAS_PRE>
__global__ void __launch_bounds__(512)
gemm_bf16_nt_256x_kernel(GemmBf16 p, int tiles_m, int tiles_n, int m_total) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr bool OUT_F32 = (EPI == EPI_F32_RESID || EPI == EPI_F32);
    // epilogue steps: one 32 x 64 bf16 block (4 full-line stores) or one 32 x 32 fp32 sub-tile per step; the activation
    // pair (two outputs, VALU-bound) per 32 x 32 sub-tile
    constexpr int E = (OUT_F32 || (EPI == EPI_BF16_ACT && HAS_PRE)) ? 8 : 4;
    const int ntiles = tiles_m * tiles_n;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int X = w >> 2, wi = w & 3;                     // group, wave within the group (= its 64-column block)
    const int l31 = lane & 31, hi = lane >> 5;
    const int lda = (int)p.lda, ldb = (int)p.ldb, ldo = (int)p.ldo;

    const auto a_rs = make_rsrc(p.A, (unsigned)((p.M - 1) * lda + p.K) * 2u);
    const auto b_rs = make_rsrc(p.Bw, (unsigned)((p.N - 1) * ldb + p.K) * 2u);
    const unsigned out_elems = (unsigned)((p.M - 1) * ldo + p.N);
    const auto o_rs = make_rsrc(p.out, out_elems * (OUT_F32 ? 4u : 2u));
    const auto pre_rs = make_rsrc(EPI == EPI_BF16_ACT && HAS_PRE ? (const void*)p.out_pre : (const void*)p.out, out_elems * 2u);
    const auto h_rs = make_rsrc(EPI == EPI_BF16_DACT ? (const void*)p.h_pre : (const void*)p.out, out_elems * 2u);
    const auto bias_rs = make_rsrc(p.bias ? (const void*)p.bias : (const void*)p.out, p.bias ? (unsigned)p.N * 4u : 0u);
    const auto r_rs = make_rsrc(EPI == EPI_F32_RESID ? (const void*)p.residual : (const void*)p.out, out_elems * 4u);

    // tile -> (m, n): 256p's XCD-aware grouped order
    auto tile_origin = [&](int tile, int& m0, int& n0) __attribute__((always_inline)) {
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

    // ---- the operand stream.  Stage t (t = 0, 1, ...) is K slice (t mod nk) of half tile (t div nk): group
    // (t div nk) & 1 of tile (t div 2 nk).  It lives in ring slot t mod 3 and is requested during step t - 2 by the
    // group that is not in MFMAs then.  One DMA instruction moves 8 rows x 128 B; row r of an operand lives at r * 128 B,
    // its 16-B chunk c holds logical chunk c ^ ((r >> 1) & 7) (first rows below are multiples of 16, so the swizzle of
    // piece j only depends on the parity of j).
    int a_loff[2], b_loff[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int clog = (lane & 7) ^ (((jp * 8 + (lane >> 3)) >> 1) & 7);
        a_loff[jp] = ((lane >> 3) * lda + clog * 8) * 2;
        b_loff[jp] = ((lane >> 3) * ldb + clog * 8) * 2;
    }
    // frontier = the next stage to request: its K slice, group, tile and that tile's origin; its ring slot
    int f_k = 0, f_grp = 0, f_tile = 0, f_m0 = 0, f_n0 = 0, f_slot = 0;
    if (ntw > 0) tile_origin(blockIdx.x, f_m0, f_n0);
    auto frontier_step = [&]() __attribute__((always_inline)) {
        f_slot = (f_slot == 2) ? 0 : f_slot + 1;
        if (++f_k == nk) {
            f_k = 0;
            f_grp ^= 1;
            if (f_grp == 0 && ++f_tile < ntw) tile_origin(blockIdx.x + f_tile * gridDim.x, f_m0, f_n0);
        }
    };
    // NPA / NPB pieces of the frontier stage: rows [arow, +8 NPA) of its A half, rows [brow, +8 NPB) of the weight panel
    auto request = [&](int arow, int brow, auto npa_c, auto npb_c) __attribute__((always_inline)) -> bool {
        constexpr int NPA = decltype(npa_c)::value, NPB = decltype(npb_c)::value;
        if (f_tile >= ntw) return false;
        __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)lds + f_slot * X_STAGE;
        const int sob = __builtin_amdgcn_readfirstlane(((f_n0 + brow) * ldb + f_k * P_K) * 2);
#pragma unroll
        for (int j = 0; j < NPB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_ptr_t)(dst + X_A_BYTES + brow * 128 + j * 1024), 16, b_loff[j & 1],
                                                     sob + j * 16 * ldb, 0, 0);
        const int soa = __builtin_amdgcn_readfirstlane(((f_m0 + f_grp * 128 + arow) * lda + f_k * P_K) * 2);
#pragma unroll
        for (int j = 0; j < NPA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(dst + arow * 128 + j * 1024), 16, a_loff[j & 1],
                                                     soa + j * 16 * lda, 0, 0);
        return true;
    };
    const std::integral_constant<int, 2> np2;
    const std::integral_constant<int, 4> np4;
    const std::integral_constant<int, 8> np8;

    // ---- fragments: per-lane part by k-slice + ring slot offset kept opaque to the optimiser (as in 256p)
    const int swz = (l31 >> 1) & 7;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned fa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fa[kk] = lds_base + l31 * 128 + (((kk * 2 + hi) ^ swz) << 4);
    const int ab_delta = X_A_BYTES + wi * 64 * 128;
    int c_slot = 0;                                      // ring slot of the stage of the current step
    f32x16 acc[4][2];
    i32x4 a0[4], b0[2], a1[4], b1[2];
    auto load_frags = [&](int slot, int kk, i32x4 (&a)[4], i32x4 (&b)[2]) __attribute__((always_inline)) {
        int oa = slot * X_STAGE;
        asm volatile("" : "+s"(oa));
        const unsigned aa = fa[kk] + oa, bb = aa + ab_delta;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a[3]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(b[1]) : "v"(bb));
    };
    auto mma = [&](const i32x4 (&a)[4], const i32x4 (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                    __builtin_bit_cast(bf16x8, a[i]), acc[i][j], 0, 0, 0);
    };
    // 8 MFMAs of one k-slice with the 6 fragment reads of slice kk of the stage in `slot` issued between them
    auto mma_lf = [&](const i32x4 (&a)[4], const i32x4 (&b)[2], int slot, int kk, i32x4 (&na)[4], i32x4 (&nb_)[2])
                      __attribute__((always_inline)) {
        int oa = slot * X_STAGE;
        asm volatile("" : "+s"(oa));
        const unsigned aa = fa[kk] + oa, bb = aa + ab_delta;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                    __builtin_bit_cast(bf16x8, a[i]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0 && j == 0) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(nb_[0]) : "v"(bb));
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(nb_[1]) : "v"(bb));
                } else if (i == 0 && j == 1) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(na[0]) : "v"(aa));
                    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(na[1]) : "v"(aa));
                } else if (i == 1 && j == 0) {
                    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(na[2]) : "v"(aa));
                    asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(na[3]) : "v"(aa));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto init_acc = [&](int mi, int ni) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
    };
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) { init_acc(mi, 0); init_acc(mi, 1); }

    // optional timeline (test hook rvlm_k_gemm_x_set_trace): per wave, s_memtime at kernel start / end and at the start of
    // the MFMAs, the end of the MFMAs and the end of the epilogue of each of its first 8 tiles: [wg][wave][2 + 3 * 8]
    auto stamp = [&](int k) {
        if (p.trace) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.trace[((long)blockIdx.x * 8 + w) * 26 + k] = t;
        }
    };
    stamp(0);

    // ---- prologue: stages 0 and 1, requested by all 8 waves (2 + 4 pieces each per stage) ----
    request(w * 16, w * 32, np2, np4);
    frontier_step();
    request(w * 16, w * 32, np2, np4);
    frontier_step();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // lane offsets of the epilogue's coalesced accesses (bytes), as in 256p
    const int st16_loff = ((lane >> 3) * ldo + (lane & 7) * 8) * 2;   // bf16: 8 rows x 128 B (64 columns) per instruction
    const int st32_loff = ((lane >> 3) * ldo + (lane & 7) * 4) * 4;   // fp32: 8 rows x 128 B (32 columns)
    const int h16_loff = ((lane >> 2) * ldo + (lane & 3) * 8) * 2;    // bf16 32-column sub-tile: 16 rows x 64 B
    const int r0 = lane >> 3;
    const unsigned ebuf = lds_base + X_STAGING + wi * P_EPI_WAVE;
    const unsigned w16_pre = ebuf + l31 * 128 + ((hi ^ (l31 & 15)) << 3);
    const unsigned wp_pre = ebuf + l31 * 128 + ((hi ^ pair_key(l31)) << 3);
    const unsigned w32_pre = ebuf + l31 * 128 + ((hi ^ (l31 & 7)) << 4);
    const unsigned r16_a = ebuf + r0 * 128 + (((lane & 7) ^ (r0 >> 1)) << 4);
    const unsigned r16_b = ebuf + r0 * 128 + (((lane & 7) ^ (r0 >> 1) ^ 4) << 4);
    const unsigned r32 = ebuf + r0 * 128 + (((lane & 7) ^ r0) << 4);

    // end of a step (every wave): advance the step bookkeeping behind the barrier
    auto step_end = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        c_slot = (c_slot == 2) ? 0 : c_slot + 1;
        frontier_step();
    };
    // a step of the group that is NOT in MFMAs, without epilogue work: request the frontier stage (12 pieces per wave),
    // make sure the previous step's requests have landed, barrier
    auto idle_step = [&]() __attribute__((always_inline)) {
        if (request(wi * 32, wi * 64, np4, np8)) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        step_end();
    };

    for (int s = 0; s < (X ? nk : 0); ++s) idle_step();      // group 1's first turn comes after group 0's first half tile

    for (int ti = 0; ti < ntw; ++ti) {
        int m0, n0;
        tile_origin(blockIdx.x + ti * gridDim.x, m0, n0);
        if (ti < 8) stamp(2 + 3 * ti);
        // ---- nk steps of MFMAs: nothing but MFMAs and fragment reads (vmcnt(0): requests this wave made in its last idle
        // steps must have landed before the barrier that publishes them)
        load_frags(c_slot, 0, a0, b0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_lf(a0, b0, c_slot, 1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_lf(a1, b1, c_slot, 2, a0, b0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_lf(a0, b0, c_slot, 3, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            step_end();
            if (kt + 1 < nk) {       // the last k-slice's MFMAs with the next stage's first fragments (its barrier is behind us)
                mma_lf(a1, b1, c_slot, 0, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- the other group's turn: E epilogue steps (the first one opens with the requests, then the tile's last 8
        // MFMAs), then request-only steps.  The last half tile of the workgroup (group 1) has only its epilogue left.
        const int off_steps = (X == 1 && ti + 1 == ntw) ? E : nk;
        const int m_base = m0 + X * 128, n_base = n0 + wi * 64;
        float4 bv[2][4];
        // side input of the epilogue (fp32 residual / stored act'(h)) of chunk cc, read in the store pattern.  VMEM returns
        // in order, so these loads are issued IN FRONT of the step's operand requests (behind them they would not return
        // before the whole stage has landed) and one step ahead of their use: side[cc & 1]
        u32x4 side[2][4];
        auto load_side = [&](int cc) __attribute__((always_inline)) {
            if (EPI == EPI_BF16_DACT) {
                const int so = __builtin_amdgcn_readfirstlane(((m_base + cc * 32) * ldo + n_base) * 2);
#pragma unroll
                for (int it = 0; it < 4; ++it) side[cc & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, st16_loff, so + it * 16 * ldo, 0);
            } else if (EPI == EPI_F32_RESID) {
                const int so = __builtin_amdgcn_readfirstlane(((m_base + (cc >> 1) * 32) * ldo + n_base + (cc & 1) * 32) * 4);
#pragma unroll
                for (int it = 0; it < 4; ++it) side[cc & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, st32_loff, so + it * 32 * ldo, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < E; ++c) {
            if (c == 0) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)   // (a null bias has a zero-length descriptor: out-of-range loads return 0)
                        bv[ni][gq] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                            bias_rs, hi * 16, __builtin_amdgcn_readfirstlane((n0 + wi * 64 + ni * 32 + 8 * gq) * 4), 0));
                load_side(0);
            }
            if (c + 1 < E) load_side(c + 1);
            __builtin_amdgcn_sched_barrier(0);
            const bool issued = request(wi * 32, wi * 64, np4, np8);
            __builtin_amdgcn_sched_barrier(0);
            if (c == 0) {
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                if (ti < 8) stamp(3 + 3 * ti);
            }
            if (E == 4) {
                // one 32 x 64 bf16 block: bias, (x act'(h) of the forward), LDS transpose, 4 stores of 8 full lines
                const int mi = c;
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base) * 2);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float v[4] = {acc[mi][ni][gq * 4 + 0] + bv[ni][gq].x, acc[mi][ni][gq * 4 + 1] + bv[ni][gq].y,
                                            acc[mi][ni][gq * 4 + 2] + bv[ni][gq].z, acc[mi][ni][gq * 4 + 3] + bv[ni][gq].w};
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float av = v[e], dv;
                            if (EPI == EPI_BF16_ACT) actp_pair<ACT>(v[e], av, dv);
                            o[e] = (bf16_t)av;
                        }
                        lds_w64(w16_pre ^ ((ni * 8 + 2 * gq) << 3), __builtin_bit_cast(u32x2, o));
                    }
                init_acc(mi, 0);
                init_acc(mi, 1);
                u32x4 t0 = lds_r128<0>(r16_a), t1 = lds_r128<8 * 128>(r16_b), t2 = lds_r128<16 * 128>(r16_a),
                      t3 = lds_r128<24 * 128>(r16_b);
                lds_wait();
                if (r0 & 1) {   // a lane's 16 B cover two 8-B chunks, swapped when its row is odd
                    t0 = __builtin_shufflevector(t0, t0, 2, 3, 0, 1); t1 = __builtin_shufflevector(t1, t1, 2, 3, 0, 1);
                    t2 = __builtin_shufflevector(t2, t2, 2, 3, 0, 1); t3 = __builtin_shufflevector(t3, t3, 2, 3, 0, 1);
                }
                if (EPI == EPI_BF16_DACT) {
                    auto mul8 = [&](u32x4& t, const u32x4& hq) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, t), b = __builtin_bit_cast(bf16x8, hq);
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)a[e] * (float)b[e]);
                        t = __builtin_bit_cast(u32x4, o);
                    };
                    mul8(t0, side[c & 1][0]); mul8(t1, side[c & 1][1]); mul8(t2, side[c & 1][2]); mul8(t3, side[c & 1][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                store16(t0, o_rs, st16_loff, so);
                store16(t1, o_rs, st16_loff, so + 16 * ldo);
                store16(t2, o_rs, st16_loff, so + 32 * ldo);
                store16(t3, o_rs, st16_loff, so + 48 * ldo);
            } else if (!OUT_F32) {
                // activation pair, one 32 x 32 sub-tile: act'(h) in the left 64 B of the staged rows, act(h) in the right
                const int mi = c >> 1, ni = c & 1;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float v[4] = {acc[mi][ni][gq * 4 + 0] + bv[ni][gq].x, acc[mi][ni][gq * 4 + 1] + bv[ni][gq].y,
                                        acc[mi][ni][gq * 4 + 2] + bv[ni][gq].z, acc[mi][ni][gq * 4 + 3] + bv[ni][gq].w};
                    bf16x4 oa, od;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float av, dv;
                        actp_pair<ACT>(v[e], av, dv);
                        oa[e] = (bf16_t)av; od[e] = (bf16_t)dv;
                    }
                    // 8-B chunk index within the 128-B row: (which * 8 + 2 gq + hi) ^ pair_key(row); which = 0 act', 1 act
                    lds_w64(wp_pre ^ ((2 * gq) << 3), __builtin_bit_cast(u32x2, od));
                    lds_w64(wp_pre ^ ((8 + 2 * gq) << 3), __builtin_bit_cast(u32x2, oa));
                }
                init_acc(mi, ni);
                // read back: 16 rows per instruction, 4 lanes x 16 B per row and output; lane -> row rr = lane >> 2 (+16),
                // 16-B slot q = lane & 3 of output `which`: 8-B chunks (which * 8 + 2 q, + 1) ^ (rr & 15)
                const int rr = lane >> 2, q = lane & 3;
                u32x2 rq[8];
#pragma unroll
                for (int which = 0; which < 2; ++which)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int row = rr + 16 * half;
                        const unsigned base = ebuf + row * 128;
                        const int c8 = (which * 8 + 2 * q) ^ pair_key(row);      // first 8-B chunk; its pair is c8 ^ 1
                        asm volatile("ds_read_b64 %0, %1" : "=v"(rq[(which * 2 + half) * 2]) : "v"(base + (c8 << 3)) : "memory");
                        asm volatile("ds_read_b64 %0, %1" : "=v"(rq[(which * 2 + half) * 2 + 1]) : "v"(base + ((c8 ^ 1) << 3)) : "memory");
                    }
                lds_wait();
                auto join = [](u32x2 lo, u32x2 hi2) { u32x4 t; t.x = lo.x; t.y = lo.y; t.z = hi2.x; t.w = hi2.y; return t; };
                const u32x4 p0 = join(rq[0], rq[1]), p1 = join(rq[2], rq[3]), q0 = join(rq[4], rq[5]), q1 = join(rq[6], rq[7]);
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 2);
                __builtin_amdgcn_sched_barrier(0);
                store16<2>(p0, pre_rs, h16_loff, so);
                store16<2>(p1, pre_rs, h16_loff, so + 32 * ldo);
                store16(q0, o_rs, h16_loff, so);
                store16(q1, o_rs, h16_loff, so + 32 * ldo);
            } else {
                // fp32 output, one 32 x 32 sub-tile: (+ fp32 residual read in the store pattern), 4 stores of 8 full lines
                const int mi = c >> 1, ni = c & 1;
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 4);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 v = make_float4(acc[mi][ni][gq * 4 + 0] + bv[ni][gq].x, acc[mi][ni][gq * 4 + 1] + bv[ni][gq].y,
                                                 acc[mi][ni][gq * 4 + 2] + bv[ni][gq].z, acc[mi][ni][gq * 4 + 3] + bv[ni][gq].w);
                    lds_w128(w32_pre ^ (gq << 5), __builtin_bit_cast(u32x4, v));
                }
                init_acc(mi, ni);
                u32x4 t[4] = {lds_r128<0>(r32), lds_r128<8 * 128>(r32), lds_r128<16 * 128>(r32), lds_r128<24 * 128>(r32)};
                lds_wait();
                if (EPI == EPI_F32_RESID) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const float4 x = __builtin_bit_cast(float4, t[it]);
                        const float4 r = __builtin_bit_cast(float4, side[c & 1][it]);
                        t[it] = __builtin_bit_cast(u32x4, make_float4(x.x + r.x, x.y + r.y, x.z + r.z, x.w + r.w));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 4; ++it) store16(t[it], o_rs, st32_loff, so + it * 32 * ldo);
            }
            __builtin_amdgcn_sched_barrier(0);
            // in flight behind this wait at most: this step's 12 requests + its 4 stores - the previous step's requests (the
            // stage of the NEXT step) and stores are complete
            if (issued) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            step_end();
        }
        if (ti < 8) stamp(4 + 3 * ti);
        for (int s = E; s < off_steps; ++s) idle_step();
    }
    for (int s = 0; s < (X ? 0 : E); ++s) idle_step();       // group 0 waits out the epilogue of group 1's last half tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (m_total > p.M) strip_tail<EPI, ACT>(p, p.M, m_total, lds, w, lane);
    stamp(1);
}

unsigned long long* g_x_trace = nullptr;    // [256 workgroups][8 waves][26 stamps] or null
void gemm_x_set_trace(unsigned long long* ptr) { g_x_trace = ptr; }

template <int EPI, int ACT, bool HAS_PRE>
static int launch_256x(const GemmBf16& p, int tiles_m, int tiles_n, int m_total, hipStream_t s) {
    static bool attr_set = false;
    const int lds_bytes = X_STAGING + 4 * P_EPI_WAVE;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256x_kernel<EPI, ACT, HAS_PRE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_set = true;
    }
    GemmBf16 q = p;
    q.trace = g_x_trace;
    const int grid = std::min(tiles_m * tiles_n, 256);
    hipLaunchKernelGGL((gemm_bf16_nt_256x_kernel<EPI, ACT, HAS_PRE>), dim3(grid), dim3(512), lds_bytes, s, q, tiles_m, tiles_n, m_total);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// rows [0, 256 * floor(M / 256)) of the problem (+ the remainder rows as the strip phase); *rows_done = 0 when the shape
// does not qualify (the caller then takes gemm_bf16_nt_256p)
int gemm_bf16_nt_256x(const GemmBf16& p, int* rows_done, hipStream_t s) {
    *rows_done = 0;
    if (p.batch_m_rows > 0 || p.M < P_M || p.N % P_N != 0 || p.K % (2 * P_K) != 0 || p.K < 8 * P_K) return RVLM_OK;
    const long lim = 1L << 31;
    if ((long)p.M * p.lda * 2 >= lim || (long)p.N * p.ldb * 2 >= lim || (long)p.M * p.ldo * 4 >= lim) return RVLM_OK;
    const int tiles_m = p.M / P_M, tiles_n = p.N / P_N;
    GemmBf16 q = p;
    q.M = tiles_m * P_M;
    if (q.epi == EPI_F32_RESID && !q.residual) q.epi = EPI_F32;
    static int tail_on = -1;
    if (tail_on < 0) { const char* e = getenv("RVLM_GEMM_TAIL"); tail_on = e ? atoi(e) : 1; }
    const int m_total = (tail_on && p.M > q.M) ? p.M : q.M;
    static int group_m = -1;
    if (group_m < 0) { const char* e = getenv("RVLM_GEMM_GROUP_M"); group_m = e ? std::max(1, atoi(e)) : 4; }
    q.group_m = group_m;
    const bool gelu = p.act != RVLM_ACT_QUICK_GELU;
    int rc;
    switch (q.epi) {
        case EPI_BF16: rc = launch_256x<EPI_BF16, RVLM_ACT_QUICK_GELU, false>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_F32_RESID: rc = launch_256x<EPI_F32_RESID, RVLM_ACT_QUICK_GELU, false>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_F32: rc = launch_256x<EPI_F32, RVLM_ACT_QUICK_GELU, false>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_BF16_DACT: rc = launch_256x<EPI_BF16_DACT, RVLM_ACT_QUICK_GELU, false>(q, tiles_m, tiles_n, m_total, s); break;
        case EPI_BF16_ACT:
            if (q.out_pre) rc = gelu ? launch_256x<EPI_BF16_ACT, RVLM_ACT_GELU, true>(q, tiles_m, tiles_n, m_total, s)
                                     : launch_256x<EPI_BF16_ACT, RVLM_ACT_QUICK_GELU, true>(q, tiles_m, tiles_n, m_total, s);
            else rc = gelu ? launch_256x<EPI_BF16_ACT, RVLM_ACT_GELU, false>(q, tiles_m, tiles_n, m_total, s)
                           : launch_256x<EPI_BF16_ACT, RVLM_ACT_QUICK_GELU, false>(q, tiles_m, tiles_n, m_total, s);
            break;
        default: return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256x: unknown epilogue");
    }
    if (rc) return rc;
    *rows_done = m_total;
    return RVLM_OK;
}

}  // namespace rvlm
