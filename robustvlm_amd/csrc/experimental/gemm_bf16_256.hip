// bf16 MFMA GEMM, large-tile variant for gfx950: 256x256 block tile, BK = 32, 8 waves (2 x 4, each a
// 128x64 sub-tile = 4x2 v_mfma_f32_32x32x16_bf16), 4-stage LDS ring (4 x 32 KiB = 128 KiB, one
// workgroup per CU) filled by global_load_lds_dwordx4 with COUNTED vmcnt: two K-steps of DMA stay
// in flight across the single raw s_barrier per K-step (never drained to 0 in the main loop).
//
// Why 256x256: at 128x128 every CU pulls (128+128)*2 B per 128*128*2 FLOP*k => ~39 TB/s of L2 traffic
// at the MFMA peak, above the measured ~34 TB/s aggregate L2 bandwidth; 256x256 halves that.
// LDS rows are 64 B (BK bf16); the 16-B chunk index is XOR-swizzled with (row>>2)&3 so a 16-lane
// ds_read_b128 group covers all 16 slots of the 256-B bank row.  The swizzle is applied to the
// per-lane GLOBAL source address (the DMA writes LDS lane-linearly).
//
// Requirements: M % 256 == 0 rows handled here (the caller runs the 128x128 kernel on the remainder
// rows), N % 256 == 0, K % 32 == 0.
#include "../kernels.h"
#include "../gemm_epilogue.h"

namespace rvlm {

typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int L_M = 256, L_N = 256, L_K = 32, L_STAGES = 4;
constexpr int L_OPER_BYTES = L_M * L_K * 2;        // 16 KiB per operand per stage
constexpr int L_STAGE_BYTES = 2 * L_OPER_BYTES;    // 32 KiB

__device__ __forceinline__ void glds16b(const void* gptr, void* lds_ptr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_ptr, 16, 0, 0);
}

template <int EPI, bool PRIO, bool SUPER>
__global__ void __launch_bounds__(512)
gemm_bf16_nt_256_kernel(GemmBf16 p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int nwg = gridDim.x, pid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = pid & 7, loc = pid >> 3;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int group_size = 8 * tiles_n;
    const int first_m = (t / group_size) * 8;
    const int gm = min(tiles_m - first_m, 8);
    const int tm = first_m + (t % group_size) % gm;
    const int tn = (t % group_size) / gm;
    const int m0 = tm * L_M, n0 = tn * L_N;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;

    // ---- staging: wave w loads rows [32w, 32w+32) of both operand tiles, 16 rows (1 KiB) per DMA ----
    const bf16_t* a_src[2];
    const bf16_t* b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w * 32 + j * 16 + (lane >> 2);
        const int clog = (lane & 3) ^ ((r >> 2) & 3);
        a_src[j] = p.A + (long)(m0 + r) * p.lda + clog * 8;
        b_src[j] = p.Bw + (long)(n0 + r) * p.ldb + clog * 8;
    }
    const int stage_wave_off = (w * 32) * 64;

    // ---- fragment offsets ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    int koff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) koff[kk] = ((kk * 2 + hi) ^ swz) << 4;
    const int a_row_off = (wm * 128 + l31) * 64;
    const int b_row_off = L_OPER_BYTES + (wn * 64 + l31) * 64;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / L_K;
    auto issue = [&](int kt) {
        char* dst = lds + (kt & (L_STAGES - 1)) * L_STAGE_BYTES + stage_wave_off;
        const int ko = kt * L_K;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16b(a_src[j] + ko, dst + j * 1024);
            glds16b(b_src[j] + ko, dst + L_OPER_BYTES + j * 1024);
        }
    };
    // Fragment reads are inline asm so that the waits can be COUNTED by hand: hipcc's own scoreboard
    // loses the in-order information across the loop back-edge and falls back to lgkmcnt(0), which
    // would drain the reads that were just issued for the next k-slice.
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto load_frags = [&](int kt, int kk, i32x4 (&a)[4], i32x4 (&b)[2]) {
        const unsigned st = lds_base + (kt & (L_STAGES - 1)) * L_STAGE_BYTES + koff[kk];
        const unsigned aa = st + a_row_off, bb = st + b_row_off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(a[3]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b[1]) : "v"(bb));
    };
    auto mma = [&](const i32x4 (&a)[4], const i32x4 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                    __builtin_bit_cast(bf16x8, a[i]), acc[i][j],
                                                                    0, 0, 0);
    };

    // Software pipeline (fragments double-buffered in registers): while the 8 MFMAs of one k-slice
    // issue, the ds_read_b128s of the next k-slice are in flight, so LDS latency never idles the matrix
    // pipe.
    i32x4 a0[4], b0[2], a1[4], b1[2];
    if (!SUPER) {
        // one barrier per BK=32 stage; stage kt+1 is landed and barrier-published one step early, two
        // K-steps of DMA (kt+2, kt+3) stay in flight during step kt.
#pragma unroll
        for (int s = 0; s < L_STAGES - 1; ++s)
            if (s < nk) issue(s);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(0, 0, a0, b0);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // stage kt+1 visible to all; slot of stage kt-1 is free
            __builtin_amdgcn_sched_barrier(0);
            if (kt + L_STAGES - 1 < nk) issue(kt + L_STAGES - 1);
            load_frags(kt, 1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");    // a0/b0 (issued one group earlier) landed
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            mma(a0, b0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) {
                load_frags(kt + 1, 0, a0, b0);
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");   // a1/b1 landed, next a0/b0 in flight
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            mma(a1, b1);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
    } else {
        // "super-stage" schedule: the ring is 2 x (two BK=32 stages); ONE barrier per 64-deep K-step
        // (32 MFMAs per wave between barriers), the other super-stage's 8 DMAs per wave in flight.
        const int ns = (nk + 1) / 2;
        auto issue_super = [&](int sp) { issue(2 * sp); if (2 * sp + 1 < nk) issue(2 * sp + 1); };
        issue_super(0);
        if (ns > 1) issue_super(1);
        for (int sp = 0; sp < ns; ++sp) {
            // retire super-stage sp; only in step 0 is a younger super-stage (1) already in flight
            if (sp == 0 && ns > 1) {
                if (3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();   // super-stage sp visible; every wave is done reading sp-1
            __builtin_amdgcn_sched_barrier(0);
            if (sp >= 1 && sp + 1 < ns) issue_super(sp + 1);   // refills the slots of super-stage sp-1
            const int k0 = 2 * sp;
            const bool two = (k0 + 1 < nk);
            load_frags(k0, 0, a0, b0);
            load_frags(k0, 1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (two) { load_frags(k0 + 1, 0, a0, b0); asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); }
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
                load_frags(k0 + 1, 1, a1, b1);
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
            }
        }
    }

    // ---- epilogue (wave-private LDS transpose -> full-line global accesses) ----
    __builtin_amdgcn_s_barrier();   // all DMA retired (vmcnt(0) above), every wave done with the ring
    gemm_epilogue<EPI, 4, 2>(acc, p, m0 + wm * 128, n0 + wn * 64, lane, lds + w * EPI_LDS_BYTES_PER_WAVE);
}

// =================================================================================================
// Staggered ("role-split") schedule.  Same tile / ring / swizzle as above, but every K-step is cut into
// two barrier-delimited phases,  L(t): issue the DMAs of stage t+3 + the 12 ds_read_b128 of stage t,
// M(t): the 16 MFMAs of stage t,  and wave group B (waves 4-7) runs ONE BARRIER BEHIND group A
// (waves 0-3): B executes one extra s_barrier before its loop, A one extra after.  A workgroup's waves
// go to the SIMDs in cyclic order, so wave w and wave w+4 share a SIMD: while one of them is in its
// MFMA phase the other is in its load phase, i.e. the matrix pipe no longer idles while both waves issue
// DMA / LDS reads in lockstep (measured: 53 % MFMA-busy with the lockstep schedule above).
// Because the two groups are one barrier apart, every producer->consumer edge through LDS keeps one
// barrier of slack: own-DMA waits for stage t+1 sit at the end of L(t) (two barriers before the reads in
// L(t+1)), and the slot of stage t is refilled in L(t+1) (two barriers after its last read).
// =================================================================================================
template <int EPI>
__global__ void __launch_bounds__(512)
gemm_bf16_nt_256s_kernel(GemmBf16 p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int nwg = gridDim.x, pid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = pid & 7, loc = pid >> 3;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int group_size = 8 * tiles_n;
    const int first_m = (t / group_size) * 8;
    const int gm = min(tiles_m - first_m, 8);
    const int tm = first_m + (t % group_size) % gm;
    const int tn = (t % group_size) / gm;
    const int m0 = tm * L_M, n0 = tn * L_N;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const bool groupB = w >= 4;

    const bf16_t* a_src[2];
    const bf16_t* b_src[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = w * 32 + j * 16 + (lane >> 2);
        const int clog = (lane & 3) ^ ((r >> 2) & 3);
        a_src[j] = p.A + (long)(m0 + r) * p.lda + clog * 8;
        b_src[j] = p.Bw + (long)(n0 + r) * p.ldb + clog * 8;
    }
    const int stage_wave_off = (w * 32) * 64;
    const int l31 = lane & 31, hi = lane >> 5;
    const int swz = (l31 >> 2) & 3;
    int koff[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) koff[kk] = ((kk * 2 + hi) ^ swz) << 4;
    const int a_row_off = (wm * 128 + l31) * 64;
    const int b_row_off = L_OPER_BYTES + (wn * 64 + l31) * 64;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / L_K;
    auto issue = [&](int kt) {
        char* dst = lds + (kt & (L_STAGES - 1)) * L_STAGE_BYTES + stage_wave_off;
        const int ko = kt * L_K;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16b(a_src[j] + ko, dst + j * 1024);
            glds16b(b_src[j] + ko, dst + L_OPER_BYTES + j * 1024);
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto load_frags = [&](int kt, int kk, i32x4 (&a)[4], i32x4 (&b)[2]) {
        const unsigned st = lds_base + (kt & (L_STAGES - 1)) * L_STAGE_BYTES + koff[kk];
        const unsigned aa = st + a_row_off, bb = st + b_row_off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(a[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(a[3]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(b[1]) : "v"(bb));
    };
    auto mma = [&](const i32x4 (&a)[4], const i32x4 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                    __builtin_bit_cast(bf16x8, a[i]), acc[i][j],
                                                                    0, 0, 0);
    };

    // ---- prologue: stages 0..2 in flight, stage 0 landed (own part) before barrier X ----
#pragma unroll
    for (int s = 0; s < L_STAGES - 1; ++s)
        if (s < nk) issue(s);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // X
    if (groupB) __builtin_amdgcn_s_barrier();        // E : group B now runs one barrier behind group A

    i32x4 a0[4], b0[2], a1[4], b1[2];
    for (int kt = 0; kt < nk; ++kt) {
        // ---------------- L(kt) ----------------
        __builtin_amdgcn_s_barrier();                // B_alpha(kt)
        __builtin_amdgcn_sched_barrier(0);
        if (kt + L_STAGES - 1 < nk) issue(kt + L_STAGES - 1);   // slot of stage kt-1 (last read in L(kt-1))
        load_frags(kt, 0, a0, b0);
        load_frags(kt, 1, a1, b1);
        // own DMA of stage kt+1 must have landed two barriers before L(kt+1) reads it
        if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- M(kt) ----------------
        __builtin_amdgcn_s_barrier();                // B_beta(kt)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        mma(a0, b0);
        mma(a1, b1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!groupB) __builtin_amdgcn_s_barrier();       // E': pairs with group B's last B_beta

    __builtin_amdgcn_s_barrier();                    // F: every wave is done with the ring
    gemm_epilogue<EPI, 4, 2>(acc, p, m0 + wm * 128, n0 + wn * 64, lane, lds + w * EPI_LDS_BYTES_PER_WAVE);
}


// =================================================================================================
// BK = 64 schedule ("k64").  Same 256x256 tile and wave layout, but LDS rows are 128 B (one full L2
// line per operand row per K-step) and the ring is 2 x 64 KiB.  Motivation (PMC, cube 8192): the BK=32
// kernel above and the 128x128x64 kernel issue the SAME number of TCC requests per second (~136 G/s)
// although the latter moves twice the bytes -- the 64-B rows of BK=32 halve the payload per request.
// One barrier per 64-deep K-step, placed between the last fragment read of the step and its last MFMA
// group: after it (a) stage kt+1 (issued a full K-step earlier) is visible to every wave and (b) the slot
// of stage kt is free, so stage kt+2 is issued and the first fragments of stage kt+1 are fetched while the
// last 8 MFMAs of stage kt run.  Only one stage is ever in flight at the wait, so it is a plain vmcnt(0).
// =================================================================================================
constexpr int K64_OPER_BYTES = L_M * 64 * 2;       // 32 KiB per operand per stage
constexpr int K64_STAGE_BYTES = 2 * K64_OPER_BYTES;

template <int EPI>
__global__ void __launch_bounds__(512)
gemm_bf16_nt_256k_kernel(GemmBf16 p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int nwg = gridDim.x, pid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = pid & 7, loc = pid >> 3;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int group_size = 8 * tiles_n;
    const int first_m = (t / group_size) * 8;
    const int gm = min(tiles_m - first_m, 8);
    const int tm = first_m + (t % group_size) % gm;
    const int tn = (t % group_size) / gm;
    const int m0 = tm * L_M, n0 = tn * L_N;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;

    // ---- staging: wave w loads rows [32w, 32w+32) of both operand tiles, 8 rows x 128 B per DMA ----
    const bf16_t* a_src[4];
    const bf16_t* b_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = w * 32 + j * 8 + (lane >> 3);
        const int clog = (lane & 7) ^ ((r >> 1) & 7);
        a_src[j] = p.A + (long)(m0 + r) * p.lda + clog * 8;
        b_src[j] = p.Bw + (long)(n0 + r) * p.ldb + clog * 8;
    }
    const int stage_wave_off = (w * 32) * 128;

    const int l31 = lane & 31, hi = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + hi) ^ swz) << 4;
    const int a_row_off = (wm * 128 + l31) * 128;
    const int b_row_off = K64_OPER_BYTES + (wn * 64 + l31) * 128;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / 64;
    auto issue = [&](int kt) {
        char* dst = lds + (kt & 1) * K64_STAGE_BYTES + stage_wave_off;
        const int ko = kt * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16b(a_src[j] + ko, dst + j * 1024);
            glds16b(b_src[j] + ko, dst + K64_OPER_BYTES + j * 1024);
        }
    };
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto load_frags = [&](int kt, int kk, i32x4 (&a)[4], i32x4 (&b)[2]) {
        const unsigned st = lds_base + (kt & 1) * K64_STAGE_BYTES + koff[kk];
        const unsigned aa = st + a_row_off, bb = st + b_row_off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[2]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a[3]) : "v"(aa));
        asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(bb));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(b[1]) : "v"(bb));
    };
    auto mma = [&](const i32x4 (&a)[4], const i32x4 (&b)[2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]),
                                                                    __builtin_bit_cast(bf16x8, a[i]), acc[i][j],
                                                                    0, 0, 0);
    };

    i32x4 a0[4], b0[2], a1[4], b1[2];
    issue(0);
    if (nk > 1) { issue(1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    load_frags(0, 0, a0, b0);
    for (int kt = 0; kt < nk; ++kt) {
        load_frags(kt, 1, a1, b1);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(kt, 2, a0, b0);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(kt, 3, a1, b1);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        // last reads of stage kt landed + own DMA of stage kt+1 landed -> publish / free the slot
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) issue(kt + 2);
        if (kt + 1 < nk) load_frags(kt + 1, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }

    __builtin_amdgcn_s_barrier();   // every wave is done with the ring
    gemm_epilogue<EPI, 4, 2>(acc, p, m0 + wm * 128, n0 + wn * 64, lane, lds + w * EPI_LDS_BYTES_PER_WAVE);
}

template <int EPI>
static int launch_256(const GemmBf16& p, int tiles_m, int tiles_n, hipStream_t s) {
    static bool attr_set = false;
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("RVLM_GEMM_SUPER"); prio = e ? atoi(e) : 0; }
    const size_t lds_bytes = (size_t)L_STAGES * L_STAGE_BYTES;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<EPI, false, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256_kernel<EPI, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_set = true;
    }
    if (prio == 3 && p.K % 64 == 0) {
        static bool attr3 = false;
        if (!attr3) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256k_kernel<EPI>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * K64_STAGE_BYTES);
            if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
            attr3 = true;
        }
        hipLaunchKernelGGL((gemm_bf16_nt_256k_kernel<EPI>), dim3(tiles_m * tiles_n), dim3(512), 2 * K64_STAGE_BYTES, s, p,
                           tiles_m, tiles_n);
    } else if (prio == 2) {
        static bool attr2 = false;
        if (!attr2) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256s_kernel<EPI>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
            attr2 = true;
        }
        hipLaunchKernelGGL((gemm_bf16_nt_256s_kernel<EPI>), dim3(tiles_m * tiles_n), dim3(512), lds_bytes, s, p,
                           tiles_m, tiles_n);
    } else if (prio)
        hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<EPI, false, true>), dim3(tiles_m * tiles_n), dim3(512), lds_bytes, s, p,
                           tiles_m, tiles_n);
    else
        hipLaunchKernelGGL((gemm_bf16_nt_256_kernel<EPI, false, false>), dim3(tiles_m * tiles_n), dim3(512), lds_bytes, s, p,
                           tiles_m, tiles_n);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// rows [0, 256*floor(M/256)) of the problem; returns the number of rows it covered in *rows_done
int gemm_bf16_nt_256(const GemmBf16& p, int* rows_done, hipStream_t s) {
    *rows_done = 0;
    if (p.M < L_M || p.N % L_N != 0 || p.K % L_K != 0 || p.K < L_K) return RVLM_OK;
    GemmBf16 q = p;
    const int tiles_m = p.M / L_M, tiles_n = p.N / L_N;
    q.M = tiles_m * L_M;
    int rc;
    switch (p.epi) {
        case EPI_BF16: rc = launch_256<EPI_BF16>(q, tiles_m, tiles_n, s); break;
        case EPI_F32_RESID: rc = launch_256<EPI_F32_RESID>(q, tiles_m, tiles_n, s); break;
        case EPI_BF16_ACT: rc = launch_256<EPI_BF16_ACT>(q, tiles_m, tiles_n, s); break;
        case EPI_BF16_DACT: rc = launch_256<EPI_BF16_DACT>(q, tiles_m, tiles_n, s); break;
        case EPI_F32: rc = launch_256<EPI_F32>(q, tiles_m, tiles_n, s); break;
        default: return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256: unknown epilogue");
    }
    if (rc) return rc;
    *rows_done = q.M;
    return RVLM_OK;
}

}  // namespace rvlm
