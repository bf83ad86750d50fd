// bf16 MFMA GEMM for gfx950, PERSISTENT 256x256 tiles with the two wave groups of a workgroup PHASE-SHIFTED by half a
// tile ("256x": cross-phased / ping-pong form of gemm_bf16_256p.hip).
//
// Why: the 256p kernel's mainloop is within 4 % of what two waves per SIMD and one
// barrier per K-step give, but its fused epilogues - 4 k (bf16) to 17 k (fp32 + residual) shader cycles beside 38 k of
// mainloop at K = 1024 - run with all 8 waves in lockstep, i.e. with the matrix pipe idle: the kernel's own cube rate is
// 1 440 TFLOP/s, its average over the encoder's launches 1 040.
//
// Structure.  A workgroup is two groups of four waves, one wave of each group per SIMD: group X owns rows [128 X, +128) of
// "its" 256x256 tile (wave tile 128 x 64 as in 256p, 128 accumulators per lane).  Both groups walk the SAME tile list, but
// group 1 runs H = (nk + E) / 2 K-steps behind group 0 (nk = K / 64 steps of MFMAs per tile, E = the epilogue cut into E
// barrier-to-barrier steps).  While one group is in its epilogue the other is in the middle of a tile and has the matrix
// pipe to itself (a lone wave issues an MFMA every ~39 cycles against 2 x 37.5 shared), so the pipe idles only when BOTH
// groups are out of MFMAs: never in steady state (E <= nk).
//   * One operand stream, one barrier per step for all 8 waves (gfx950 has no sub-workgroup barrier): stage g holds the
//     64-deep K slice (g mod nk) of the weight panel in the B ring and, in the A ring, that slice of the rows of whichever
//     tile each group is on.  Both groups therefore always sit at the same K slice and a group starts its tile at
//     whatever slice the stream is at: a tile's K sum is a ROTATION of 0 .. nk-1 (fp32 accumulation order differs from
//     256p's; results agree to fp32 rounding of the sum).  The workgroup keeps its weight panel as long as it can (tile
//     order below); when the panel changes group 0 waits the H - E steps group 1 still needs the old one.
//   * Operand requests are made by whoever is NOT in MFMAs: both groups computing -> the 256p split roles (group 1 all
//     of B right after the barrier, group 0 all of A at the end of its step); one group computing -> the other one
//     requests everything right after the barrier and the computing group issues nothing but MFMAs and fragment reads.
//   * Epilogue staging (wave-private LDS transpose -> full-line stores) lives in the A-ring half of the group that is in
//     its epilogue: that half is not requested for steps in which its group does not compute.
//   * LDS 160 KiB as in 256p: A ring 3 x 32 KiB (two steps ahead), B ring 2 x 32 KiB (one step ahead).
//
// Requirements (else the dispatcher keeps 256p): N % 256 == 0, K % 128 == 0, K >= 512, (M / 256) % (32 / nb) == 0 and
// tiles % 256 == 0 (every workgroup gets the same number of tiles, every XCD whole 32-tile blocks), no batched form.
#include "kernels.h"
#include "gemm_persist.h"
#include "gemm_strip.h"
#include <type_traits>

namespace rvlm {

constexpr int X_HALF = P_OPER_BYTES / 2;   // one group's 128 rows of an A stage

template <int EPI, int ACT, bool HAS_PRE>
__global__ void __launch_bounds__(512)
gemm_bf16_nt_256x_kernel(GemmBf16 p, int nb, int mblocks, int blocks_per_xcd, int m_total) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr bool OUT_F32 = (EPI == EPI_F32_RESID || EPI == EPI_F32);
    // epilogue steps: one 32 x 64 bf16 block (4 full-line stores) or one 32 x 32 fp32 sub-tile per step; the activation
    // pair (two outputs, VALU-bound) per 32 x 32 sub-tile
    constexpr int E = (OUT_F32 || (EPI == EPI_BF16_ACT && HAS_PRE)) ? 8 : 4;

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

    // ---- tile list of this workgroup: XCD x (= blockIdx & 7) owns blocks [x * bpx, (x + 1) * bpx) of 32 tiles (mb m-tiles
    // x nb n-tiles, the L2 working set of 256p's grouped order), numbered m-fastest inside an n-set so that the weight
    // panel of a workgroup changes as rarely as possible; workgroup l of the XCD keeps position (l % mb, l / mb) in
    // every block.
    const int T = blocks_per_xcd;
    const int mb = 32 / nb;
    const int xl = (int)blockIdx.x >> 3, li = xl % mb, lj = xl / mb;
    const int blk0 = ((int)blockIdx.x & 7) * blocks_per_xcd;
    auto tile_m0 = [&](int t) { return (((blk0 + t) % mblocks) * mb + li) * P_M; };
    auto tile_n0 = [&](int t) { return (((blk0 + t) / mblocks) * nb + lj) * P_N; };

    const int nk = p.K / P_K;
    const int P = nk + E, H = P >> 1;      // period of a group's tile, lag of group 1 (nk and E even)
    const int stall_len = H - E;           // extra idle steps of a group behind a tile whose successor has another panel

    // ---- schedule state, identical in every wave (scalar): for each group Z its mode at steps g, g+1, g+2 and the
    // tile rows / panel of the frontier.  cur* is the cursor of group Z at time g + 2.
    // (every field is a separate scalar: a runtime-indexed array of flags would live in vector registers)
    struct Cur { int tile, ph, per, m0, n0; };
    Cur cur0, cur1;
    bool c0_0, c0_1, c1_0, c1_1, c2_0, c2_1;   // cK_Z: group Z computes at step g + K
    int n1_0, n1_1, n2_0, n2_1, m2_0, m2_1;    // panel at g+1 / g+2, tile rows at g+2
    auto cur_load = [&](Cur& c) __attribute__((always_inline)) {
        if (c.tile < T) {
            c.m0 = tile_m0(c.tile);
            c.n0 = tile_n0(c.tile);
            const int nn = c.tile + 1 < T ? tile_n0(c.tile + 1) : c.n0;
            c.per = P + (nn != c.n0 ? stall_len : 0);
        } else {
            c.per = 1 << 30;
        }
    };
    auto cur_computes = [&](const Cur& c) __attribute__((always_inline)) { return c.tile < T && c.ph >= 0 && c.ph < nk; };
    auto cur_step = [&](Cur& c) __attribute__((always_inline)) {
        if (++c.ph == c.per) { ++c.tile; c.ph = 0; cur_load(c); }
    };
    cur0.tile = 0; cur0.ph = 0; cur0.m0 = 0; cur0.n0 = 0; cur_load(cur0);
    cur1.tile = 0; cur1.ph = -H; cur1.m0 = 0; cur1.n0 = 0; cur_load(cur1);
    c0_0 = cur_computes(cur0); c0_1 = cur_computes(cur1);
    cur_step(cur0); cur_step(cur1);
    c1_0 = cur_computes(cur0); c1_1 = cur_computes(cur1); n1_0 = cur0.n0; n1_1 = cur1.n0;
    cur_step(cur0); cur_step(cur1);
    c2_0 = cur_computes(cur0); c2_1 = cur_computes(cur1); n2_0 = cur0.n0; n2_1 = cur1.n0; m2_0 = cur0.m0; m2_1 = cur1.m0;
    int kb1 = 1 % nk, kb2 = 2 % nk;         // K slice of stages g+1, g+2 (g = global step, between barrier g-1 and barrier g)
    int sa0 = 0, sa2 = 2, sb0 = 0, sb1 = 1; // ring slots: A of stage g / g+2, B of stage g / g+1
    auto advance = [&]() __attribute__((always_inline)) {   // g -> g + 1 (called right after a barrier)
        c0_0 = c1_0; c0_1 = c1_1; c1_0 = c2_0; c1_1 = c2_1; n1_0 = n2_0; n1_1 = n2_1;
        cur_step(cur0); cur_step(cur1);
        c2_0 = cur_computes(cur0); c2_1 = cur_computes(cur1); n2_0 = cur0.n0; n2_1 = cur1.n0; m2_0 = cur0.m0; m2_1 = cur1.m0;
        kb1 = kb2; kb2 = (kb2 + 1 == nk) ? 0 : kb2 + 1;
        sa0 = (sa0 == 2) ? 0 : sa0 + 1; sa2 = (sa2 == 2) ? 0 : sa2 + 1;
        sb0 ^= 1; sb1 ^= 1;
    };

    // ---- operand requests.  One DMA instruction moves 8 rows x 128 B; row r of a stage lives at r * 128 B of its slot,
    // its 16-B chunk c holds logical chunk c ^ ((r >> 1) & 7) (first rows below are multiples of 16, so the swizzle of
    // piece j only depends on the parity of j).
    int a_loff[2], b_loff[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int clog = (lane & 7) ^ (((jp * 8 + (lane >> 3)) >> 1) & 7);
        a_loff[jp] = ((lane >> 3) * lda + clog * 8) * 2;
        b_loff[jp] = ((lane >> 3) * ldb + clog * 8) * 2;
    }
    // NP pieces = rows [row0, row0 + 8 NP) of an A stage (row0 in 0..255: the group halves are stacked) from tile rows m0
    auto req_a = [&](int slot, int row0, int m0, int kb, auto np_c) __attribute__((always_inline)) {
        constexpr int NP = decltype(np_c)::value;
        __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)lds + (slot * PA_SLOT + row0 * 128);
        const int so = __builtin_amdgcn_readfirstlane(((m0 + row0) * lda + kb * P_K) * 2);
#pragma unroll
        for (int j = 0; j < NP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(dst + j * 1024), 16, a_loff[j & 1], so + j * 16 * lda, 0, 0);
    };
    auto req_b = [&](int slot, int row0, int n0, int kb, auto np_c) __attribute__((always_inline)) {
        constexpr int NP = decltype(np_c)::value;
        __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)lds + (PB_BASE + slot * PB_SLOT + row0 * 128);
        const int so = __builtin_amdgcn_readfirstlane(((n0 + row0) * ldb + kb * P_K) * 2);
#pragma unroll
        for (int j = 0; j < NP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_ptr_t)(dst + j * 1024), 16, b_loff[j & 1], so + j * 16 * ldb, 0, 0);
    };
    const std::integral_constant<int, 2> np2;
    const std::integral_constant<int, 4> np4;
    const std::integral_constant<int, 8> np8;
    // Right after barrier g-1 (state = step g).  Both groups in MFMAs: group 1 requests B of stage g+1.  Otherwise the
    // group that is not in MFMAs (group 0 when neither is) requests B of stage g+1 and, for every group that computes at
    // g+2, its half of A of stage g+2 (4 pieces per wave and half).
    auto duties_after_barrier = [&]() __attribute__((always_inline)) {
        const bool cc = c0_0 && c0_1;
        const bool need_b = c1_0 || c1_1;
        const int nb1 = c1_0 ? n1_0 : n1_1;
        if (cc) {
            if (X == 1 && need_b) req_b(sb1, wi * 64, nb1, kb1, np8);
        } else if (X == (c0_0 ? 1 : 0)) {
            if (need_b) req_b(sb1, wi * 64, nb1, kb1, np8);
            if (c2_0) req_a(sa2, wi * 32, m2_0, kb2, np4);
            if (c2_1) req_a(sa2, 128 + wi * 32, m2_1, kb2, np4);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // End of step g in a both-compute step: group 0 requests A of stage g+2 (wave wi: rows [64 wi, +64) of the stacked
    // stage, i.e. the half of group wi >> 1).  Returns whether this wave issued (its 8 pieces may stay in flight).
    auto duties_end_of_step = [&]() __attribute__((always_inline)) -> bool {
        if (!(c0_0 && c0_1) || X != 0) return false;
        const bool upper = (wi >> 1) != 0;
        if (!(upper ? c2_1 : c2_0)) return false;
        req_a(sa2, wi * 64, upper ? m2_1 : m2_0, kb2, np8);
        return true;
    };

    // ---- fragments (as in 256p): per-lane part by k-slice + ring slot offsets kept opaque to the optimiser
    const int swz = (l31 >> 1) & 7;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned fa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fa[kk] = lds_base + (X * 128 + l31) * 128 + (((kk * 2 + hi) ^ swz) << 4);
    const int ab_delta = PB_BASE + (wi * 64 - X * 128) * 128;
    f32x16 acc[4][2];
    i32x4 a0[4], b0[2], a1[4], b1[2];
    auto load_frags = [&](int sa, int sb, int kk, i32x4 (&a)[4], i32x4 (&b)[2]) __attribute__((always_inline)) {
        int oa = sa * PA_SLOT, ob = sb * PB_SLOT + ab_delta;
        asm volatile("" : "+s"(oa), "+s"(ob));
        const unsigned aa = fa[kk] + oa, bb = fa[kk] + ob;
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
    // 8 MFMAs of one k-slice with the 6 fragment reads of slice kk of stage (sa, sb) issued between them
    auto mma_lf = [&](const i32x4 (&a)[4], const i32x4 (&b)[2], int sa, int sb, int kk, i32x4 (&na)[4], i32x4 (&nb_)[2])
                      __attribute__((always_inline)) {
        int oa = sa * PA_SLOT, ob = sb * PB_SLOT + ab_delta;
        asm volatile("" : "+s"(oa), "+s"(ob));
        const unsigned aa = fa[kk] + oa, bb = fa[kk] + ob;
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

    // optional timeline (test hook rvlm_k_gemm_set_trace): per wave, s_memtime at kernel start / end and at the start of
    // the MFMAs, the end of the MFMAs and the end of the epilogue of each of its first 8 tiles: [wg][wave][2 + 3 * 8]
    auto stamp = [&](int k) {
        if (p.trace) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.trace[((long)blockIdx.x * 8 + w) * 26 + k] = t;
        }
    };
    stamp(0);

    // ---- prologue: B of stage 0, group 0's half of A of stages 0 and 1 (steps 0 and 1 are group 0's: H >= 6) ----
    req_b(0, w * 32, tile_n0(0), 0, np4);
    req_a(0, w * 16, tile_m0(0), 0, np2);
    req_a(1, w * 16, tile_m0(0), 1 % nk, np2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // lane offsets of the epilogue's coalesced accesses (bytes), as in 256p
    const int st16_loff = ((lane >> 3) * ldo + (lane & 7) * 8) * 2;   // bf16: 8 rows x 128 B (64 columns) per instruction
    const int st32_loff = ((lane >> 3) * ldo + (lane & 7) * 4) * 4;   // fp32: 8 rows x 128 B (32 columns)
    const int h16_loff = ((lane >> 2) * ldo + (lane & 3) * 8) * 2;    // bf16 32-column sub-tile: 16 rows x 64 B
    const int r0 = lane >> 3;

    // a step in which this group neither computes nor stores: operand requests (if it is the requesting group), barrier
    auto idle_step = [&]() __attribute__((always_inline)) {
        duties_after_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
    };

    for (int s = 0; s < (X ? H : 0); ++s) idle_step();      // group 1 starts H steps late

    for (int ti = 0; ti < T; ++ti) {
        const int m0 = tile_m0(ti), n0 = tile_n0(ti);
        const int stall = (ti + 1 < T && tile_n0(ti + 1) != n0) ? stall_len : 0;
        if (ti < 8) stamp(2 + 3 * ti);
        // ---- nk steps of MFMAs.  Entering: barrier g-1 passed, stage g landed, nothing of this step done yet.
        duties_after_barrier();
        load_frags(sa0, sb0, 0, a0, b0);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_lf(a0, b0, sa0, sb0, 1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_lf(a1, b1, sa0, sb0, 2, a0, b0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma_lf(a0, b0, sa0, sb0, 3, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            if (duties_end_of_step()) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            advance();
            if (kt + 1 < nk) {       // the last k-slice's MFMAs with the next stage's first fragments (its barrier is behind us)
                duties_after_barrier();
                mma_lf(a1, b1, sa0, sb0, 0, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- E epilogue steps.  The first one opens with the requests (time-critical: B of the next stage has one step
        // to land), then the tile's last 8 MFMAs.
        duties_after_barrier();
        float4 bv[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)   // (a null bias has a zero-length descriptor: out-of-range loads return 0)
                bv[ni][gq] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                    bias_rs, hi * 16, __builtin_amdgcn_readfirstlane((n0 + wi * 64 + ni * 32 + 8 * gq) * 4), 0));
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        if (ti < 8) stamp(3 + 3 * ti);
        const int m_base = m0 + X * 128, n_base = n0 + wi * 64;
#pragma unroll
        for (int c = 0; c < E; ++c) {
            if (c > 0) duties_after_barrier();
            // staging: this group's half of the A slot of the current stage (not requested: the group is not computing)
            const unsigned ebuf = lds_base + sa0 * PA_SLOT + X * X_HALF + wi * P_EPI_WAVE;
            const unsigned w16_pre = ebuf + l31 * 128 + ((hi ^ (l31 & 15)) << 3);
            const unsigned w32_pre = ebuf + l31 * 128 + ((hi ^ (l31 & 7)) << 4);
            const unsigned r16_a = ebuf + r0 * 128 + (((lane & 7) ^ (r0 >> 1)) << 4);
            const unsigned r16_b = ebuf + r0 * 128 + (((lane & 7) ^ (r0 >> 1) ^ 4) << 4);
            const unsigned r32 = ebuf + r0 * 128 + (((lane & 7) ^ r0) << 4);
            if (E == 4) {
                // one 32 x 64 bf16 block: bias, (x act'(h) of the forward), LDS transpose, 4 stores of 8 full lines
                const int mi = c;
                u32x4 side[4];
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base) * 2);
                if (EPI == EPI_BF16_DACT) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) side[it] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, st16_loff, so + it * 16 * ldo, 0);
                }
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
                    mul8(t0, side[0]); mul8(t1, side[1]); mul8(t2, side[2]); mul8(t3, side[3]);
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
                    // 8-B chunk index within the 128-B row: (which * 8 + 2 gq + hi) ^ (row & 15); which = 0 act', 1 act
                    lds_w64(w16_pre ^ ((2 * gq) << 3), __builtin_bit_cast(u32x2, od));
                    lds_w64(w16_pre ^ ((8 + 2 * gq) << 3), __builtin_bit_cast(u32x2, oa));
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
                        const int c8 = (which * 8 + 2 * q) ^ (row & 15);      // first 8-B chunk; its pair is c8 ^ 1
                        asm volatile("ds_read_b64 %0, %1" : "=v"(rq[(which * 2 + half) * 2]) : "v"(base + (c8 << 3)) : "memory");
                        asm volatile("ds_read_b64 %0, %1" : "=v"(rq[(which * 2 + half) * 2 + 1]) : "v"(base + ((c8 ^ 1) << 3)) : "memory");
                    }
                lds_wait();
                auto join = [](u32x2 lo, u32x2 hi2) { u32x4 t; t.x = lo.x; t.y = lo.y; t.z = hi2.x; t.w = hi2.y; return t; };
                const u32x4 p0 = join(rq[0], rq[1]), p1 = join(rq[2], rq[3]), q0 = join(rq[4], rq[5]), q1 = join(rq[6], rq[7]);
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 2);
                __builtin_amdgcn_sched_barrier(0);
                store16(p0, pre_rs, h16_loff, so);
                store16(p1, pre_rs, h16_loff, so + 32 * ldo);
                store16(q0, o_rs, h16_loff, so);
                store16(q1, o_rs, h16_loff, so + 32 * ldo);
            } else {
                // fp32 output, one 32 x 32 sub-tile: (+ fp32 residual read in the store pattern), 4 stores of 8 full lines
                const int mi = c >> 1, ni = c & 1;
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 4);
                u32x4 side[4];
                if (EPI == EPI_F32_RESID) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) side[it] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, st32_loff, so + it * 32 * ldo, 0);
                }
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
                        const float4 r = __builtin_bit_cast(float4, side[it]);
                        t[it] = __builtin_bit_cast(u32x4, make_float4(x.x + r.x, x.y + r.y, x.z + r.z, x.w + r.w));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 4; ++it) store16(t[it], o_rs, st32_loff, so + it * 32 * ldo);
            }
            __builtin_amdgcn_sched_barrier(0);
            // the step's 4 stores may stay in flight; everything older (the requests of this step, side loads) has landed
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            advance();
        }
        if (ti < 8) stamp(4 + 3 * ti);
        for (int s = 0; s < stall; ++s) idle_step();
    }
    for (int s = 0; s < (X ? 0 : H); ++s) idle_step();      // group 0 keeps requesting operands for group 1's last half tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (m_total > p.M) strip_tail<EPI, ACT>(p, p.M, m_total, lds, w, lane);
    stamp(1);
}

unsigned long long* g_x_trace = nullptr;    // [256 workgroups][8 waves][26 stamps] or null
void gemm_x_set_trace(unsigned long long* ptr) { g_x_trace = ptr; }

template <int EPI, int ACT, bool HAS_PRE>
static int launch_256x(const GemmBf16& p, int nb, int mblocks, int bpx, int m_total, hipStream_t s) {
    static bool attr_set = false;
    const int lds_bytes = 3 * PA_SLOT + 2 * PB_SLOT;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256x_kernel<EPI, ACT, HAS_PRE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_set = true;
    }
    GemmBf16 q = p;
    q.trace = g_x_trace;
    hipLaunchKernelGGL((gemm_bf16_nt_256x_kernel<EPI, ACT, HAS_PRE>), dim3(256), dim3(512), lds_bytes, s, q, nb, mblocks, bpx, m_total);
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
    const int nb = tiles_n % 4 == 0 ? 4 : tiles_n % 2 == 0 ? 2 : 1, mb = 32 / nb;
    if (tiles_m % mb != 0) return RVLM_OK;
    const int mblocks = tiles_m / mb, blocks = mblocks * (tiles_n / nb);
    if (blocks % 8 != 0) return RVLM_OK;
    GemmBf16 q = p;
    q.M = tiles_m * P_M;
    if (q.epi == EPI_F32_RESID && !q.residual) q.epi = EPI_F32;
    static int tail_on = -1;
    if (tail_on < 0) { const char* e = getenv("RVLM_GEMM_TAIL"); tail_on = e ? atoi(e) : 1; }
    const int m_total = (tail_on && p.M > q.M) ? p.M : q.M;
    const int bpx = blocks / 8;
    const bool gelu = p.act != RVLM_ACT_QUICK_GELU;
    int rc;
    switch (q.epi) {
        case EPI_BF16: rc = launch_256x<EPI_BF16, RVLM_ACT_QUICK_GELU, false>(q, nb, mblocks, bpx, m_total, s); break;
        case EPI_F32_RESID: rc = launch_256x<EPI_F32_RESID, RVLM_ACT_QUICK_GELU, false>(q, nb, mblocks, bpx, m_total, s); break;
        case EPI_F32: rc = launch_256x<EPI_F32, RVLM_ACT_QUICK_GELU, false>(q, nb, mblocks, bpx, m_total, s); break;
        case EPI_BF16_DACT: rc = launch_256x<EPI_BF16_DACT, RVLM_ACT_QUICK_GELU, false>(q, nb, mblocks, bpx, m_total, s); break;
        case EPI_BF16_ACT:
            if (q.out_pre) rc = gelu ? launch_256x<EPI_BF16_ACT, RVLM_ACT_GELU, true>(q, nb, mblocks, bpx, m_total, s)
                                     : launch_256x<EPI_BF16_ACT, RVLM_ACT_QUICK_GELU, true>(q, nb, mblocks, bpx, m_total, s);
            else rc = gelu ? launch_256x<EPI_BF16_ACT, RVLM_ACT_GELU, false>(q, nb, mblocks, bpx, m_total, s)
                           : launch_256x<EPI_BF16_ACT, RVLM_ACT_QUICK_GELU, false>(q, nb, mblocks, bpx, m_total, s);
            break;
        default: return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256x: unknown epilogue");
    }
    if (rc) return rc;
    *rows_done = m_total;
    return RVLM_OK;
}

}  // namespace rvlm
