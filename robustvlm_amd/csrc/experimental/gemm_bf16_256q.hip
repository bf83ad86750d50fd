// bf16 MFMA GEMM for gfx950, persistent 256x256 variant with FOUR waves ("256q"): one wave per SIMD, each a
// 128x128 wave tile (4x4 v_mfma_f32_32x32x16_bf16, 256 accumulator registers out of the 512 a lone wave may use).
//
// Why a second persistent kernel: with 8 waves x 128x64 every K-step moves 8 x (128 + 64) rows x 128 B = 192 KiB
// out of LDS for 64 KiB that came in; 4 waves x 128x128 move 128 KiB (a third fewer ds_read_b128 per MFMA), the
// same ratio the vendor library's 256x256 kernel uses.  A lone wave has no partner to hide its latencies, so every
// non-MFMA instruction of the K-step is placed by hand BETWEEN two MFMAs (one filler per gap; the matrix pipe takes
// 32 cycles per 32x32x16 MFMA, the fillers issue in its shadow): the 8 fragment reads of the next k-slice ride on the
// first 8 MFMAs of a slice, the 16 DMA pieces of a K-step on the 16 MFMAs of its last slice.
//
// Everything else is gemm_bf16_256p.hip: tile order, A ring 3 x 32 KiB + B ring 2 x 32 KiB (A three K-steps ahead, B
// two), one barrier per K-step, counted vmcnt, epilogue through the freed B slot.
#include "../kernels.h"
#include "../gemm_persist.h"

namespace rvlm {

constexpr int Q_EPI_WAVE = 8192;   // staging bytes per wave (4 waves x 8 KiB = one B slot)

template <int EPI, int ACT, int ABL>
__global__ void __launch_bounds__(256)
gemm_bf16_nt_256q_kernel(GemmBf16 p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int ntiles = tiles_m * tiles_n;
    constexpr bool OUT_F32 = (EPI == EPI_F32_RESID || EPI == EPI_F32);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const int lda = (int)p.lda, ldb = (int)p.ldb, ldo = (int)p.ldo;   // byte offsets fit 31 bits (host check)

    const auto a_rs = make_rsrc(p.A, (unsigned)((p.M - 1) * lda + p.K) * 2u);
    const auto b_rs = make_rsrc(p.Bw, (unsigned)((p.N - 1) * ldb + p.K) * 2u);
    const unsigned out_elems = (unsigned)((p.M - 1) * ldo + p.N);
    const auto o_rs = make_rsrc(p.out, out_elems * (OUT_F32 ? 4u : 2u));
    const auto pre_rs = make_rsrc(EPI == EPI_BF16_ACT ? (const void*)p.out_pre : (const void*)p.out, out_elems * 2u);
    const auto h_rs = make_rsrc(EPI == EPI_BF16_DACT ? (const void*)p.h_pre : (const void*)p.out, out_elems * 2u);
    const auto bias_rs = make_rsrc(p.bias ? (const void*)p.bias : (const void*)p.out, p.bias ? (unsigned)p.N * 4u : 0u);
    const auto r_rs = make_rsrc(EPI == EPI_F32_RESID ? (const void*)p.residual : (const void*)p.out, out_elems * 4u);

    auto tile_origin = [&](int tile, int& m0, int& n0) {
        const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = tile & 7, loc = tile >> 3;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
        const int group_size = 8 * tiles_n;
        const int first_m = (t / group_size) * 8;
        const int gm = min(tiles_m - first_m, 8);
        m0 = __builtin_amdgcn_readfirstlane((first_m + (t % group_size) % gm) * P_M);
        n0 = __builtin_amdgcn_readfirstlane(((t % group_size) / gm) * P_N);
    };
    const int nk = p.K / P_K;
    const int ntw = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup

    // ---- operand streams: wave w loads rows [64w, 64w+64) of each half, 8 pieces of 8 rows x 128 B.  Row of piece j:
    // 64w + 8j + (lane>>3); its 16-B chunk (lane&7) holds logical chunk (lane&7) ^ ((row>>1)&7) -> depends on j & 1.
    int a_loff[2], b_loff[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const int clog = (lane & 7) ^ (((jp * 8 + (lane >> 3)) >> 1) & 7);
        a_loff[jp] = ((lane >> 3) * lda + clog * 8) * 2;
        b_loff[jp] = ((lane >> 3) * ldb + clog * 8) * 2;
    }
    const int stage_wave_off = (w * 64) * 128;
    int a_ti = 0, a_kt = 0, a_soff = 0, a_slot = 0;   // cursors: next half-stage to request
    int b_ti = 0, b_kt = 0, b_soff = 0, b_slot = 0;
    {
        int m0, n0;
        tile_origin(blockIdx.x, m0, n0);
        a_soff = (m0 + w * 64) * lda * 2;
        b_soff = (n0 + w * 64) * ldb * 2;
    }
    // one DMA piece (j = 0..7) of the A / B half-stage under the cursor; *_advance() moves the cursor on
    auto a_piece = [&](int j) {
        if ((ABL & 1) || a_ti >= ntw) return;
        __attribute__((address_space(3))) char* dst =
            (__attribute__((address_space(3))) char*)lds + (a_slot * PA_SLOT + stage_wave_off + j * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)dst, 16, a_loff[j & 1],
                                                 __builtin_amdgcn_readfirstlane(a_soff + a_kt * (P_K * 2) + j * 16 * lda), 0, 0);
    };
    auto a_advance = [&]() -> bool {
        if (a_ti >= ntw) return false;
        a_slot = (a_slot == 2) ? 0 : a_slot + 1;
        if (++a_kt == nk) {
            a_kt = 0;
            if (++a_ti < ntw) {
                int m0, n0;
                tile_origin(blockIdx.x + a_ti * gridDim.x, m0, n0);
                a_soff = (m0 + w * 64) * lda * 2;
            }
        }
        return true;
    };
    auto b_piece = [&](int j) {
        if ((ABL & 1) || b_ti >= ntw) return;
        __attribute__((address_space(3))) char* dst =
            (__attribute__((address_space(3))) char*)lds + (PB_BASE + b_slot * PB_SLOT + stage_wave_off + j * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_ptr_t)dst, 16, b_loff[j & 1],
                                                 __builtin_amdgcn_readfirstlane(b_soff + b_kt * (P_K * 2) + j * 16 * ldb), 0, 0);
    };
    auto b_advance = [&]() {
        if (b_ti >= ntw) return;
        b_slot ^= 1;
        if (++b_kt == nk) {
            b_kt = 0;
            if (++b_ti < ntw) {
                int m0, n0;
                tile_origin(blockIdx.x + b_ti * gridDim.x, m0, n0);
                b_soff = (n0 + w * 64) * ldb * 2;
            }
        }
    };

    // ---- fragments: one ds_read_b128 per 32-row block and k-slice; address = per-lane part + (opaque) slot offset
    const int swz = (l31 >> 1) & 7;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned fa[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fa[kk] = lds_base + (wm * 128 + l31) * 128 + (((kk * 2 + hi) ^ swz) << 4);
    const int ab_delta = PB_BASE + (wn * 128 - wm * 128) * 128;
    struct Frag { i32x4 a[4], b[4]; };
    // read number r (0..7) of k-slice kk of the stage in slots (sa, sb): r < 4 -> A block r, else B block r - 4
    auto frag_read = [&](Frag& f, unsigned aa, unsigned bb, int r) {
        if (ABL & 4) return;
        switch (r) {
            case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(f.a[0]) : "v"(aa)); break;
            case 1: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.a[1]) : "v"(aa)); break;
            case 2: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(f.a[2]) : "v"(aa)); break;
            case 3: asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(f.a[3]) : "v"(aa)); break;
            case 4: asm volatile("ds_read_b128 %0, %1" : "=v"(f.b[0]) : "v"(bb)); break;
            case 5: asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(f.b[1]) : "v"(bb)); break;
            case 6: asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(f.b[2]) : "v"(bb)); break;
            default: asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(f.b[3]) : "v"(bb)); break;
        }
    };
    auto frag_addr = [&](int sa, int sb, int kk, unsigned& aa, unsigned& bb) {
        int oa = sa * PA_SLOT, ob = sb * PB_SLOT + ab_delta;
        asm volatile("" : "+s"(oa), "+s"(ob));
        aa = fa[kk] + oa;
        bb = fa[kk] + ob;
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    auto mfma1 = [&](const Frag& f, int t) {   // MFMA number t (0..15) of a k-slice: block (t >> 2, t & 3)
        if (ABL & 2) return;
        const int i = t >> 2, j = t & 3;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[j]),
                                                            __builtin_bit_cast(bf16x8, f.a[i]), acc[i][j], 0, 0, 0);
    };

    // ---- prologue: stages 0 and 1 complete, A of stage 2; then the fragments of (stage 0, k-slice 0) ----
    bool a_ahead;   // was the A half of stage g+2 requested at the previous barrier?
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) a_piece(j);
        a_advance();
#pragma unroll
        for (int j = 0; j < 8; ++j) b_piece(j);
        b_advance();
#pragma unroll
        for (int j = 0; j < 8; ++j) a_piece(j);
        a_advance();
#pragma unroll
        for (int j = 0; j < 8; ++j) b_piece(j);
        b_advance();
#pragma unroll
        for (int j = 0; j < 8; ++j) a_piece(j);
        a_ahead = a_advance();
        if (a_ahead) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // (a single tile with two K-steps)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    Frag F0, F1;
    int ca_slot = 0, cb_slot = 0;   // slots of the stage being consumed
    {
        unsigned aa, bb;
        frag_addr(0, 0, 0, aa, bb);
#pragma unroll
        for (int r = 0; r < 8; ++r) frag_read(F0, aa, bb, r);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

    const int st16_loff = ((lane >> 3) * ldo + (lane & 7) * 8) * 2;   // bf16: 128 B (64 columns) per row
    const int st32_loff = ((lane >> 3) * ldo + (lane & 7) * 4) * 4;   // fp32: 128 B (32 columns) per row
    const int h16_loff = ((lane >> 2) * ldo + (lane & 3) * 8) * 2;    // bf16 next to a 32-column sub-tile: 64 B per row
    const int r0 = lane >> 3;

    auto stamp = [&](int ti, int k) {
        if (p.trace && w == 0 && ti < 7) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.trace[((long)blockIdx.x * 8 + ti) * 4 + k] = t;
        }
    };
    if (p.trace && w == 0 && lane == 0) {
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 0] = __builtin_amdgcn_s_memtime();
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    }

    for (int ti = 0; ti < ntw; ++ti) {
        int m0, n0;
        tile_origin(blockIdx.x + ti * gridDim.x, m0, n0);
        const bool last_tile = (ti + 1 == ntw);
        stamp(ti, 0);

        // One K-step = four k-slices of 16 MFMAs.  Slices 0-2: the 8 fragment reads of the next slice sit behind the
        // first 8 MFMAs.  Slice 3 opens with the stage hand-over (counted vmcnt + barrier: the next stage has landed
        // for every wave, this stage's slots are free), then carries the 16 DMA pieces that refill the two slots (B of
        // stage g+2 first, then A of stage g+3) and the reads of (next stage, slice 0).
        auto k_step = [&](bool first_of_tile, bool last_of_tile) {
            const int na_slot = (ca_slot == 2) ? 0 : ca_slot + 1, nb_slot = cb_slot ^ 1;
            unsigned aa, bb;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                Frag& cur = (kk & 1) ? F1 : F0;
                Frag& nxt = (kk & 1) ? F0 : F1;
                frag_addr(ca_slot, cb_slot, kk + 1, aa, bb);
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    mfma1(cur, t);
                    if (t < 8) frag_read(nxt, aa, bb, t);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            // first step of a later tile: the B half of the next stage was requested AFTER the epilogue's stores
            if (a_ahead && !(first_of_tile && ti > 0)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (first_of_tile) stamp(ti, 1);
            const bool more = !(last_of_tile && last_tile);   // is there a next stage to prefetch fragments from?
            frag_addr(na_slot, nb_slot, 0, aa, bb);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                mfma1(F1, t);
                if (t < 8) {
                    if (more) frag_read(F0, aa, bb, t);
                    if (!last_of_tile) b_piece(t);    // (B is deferred past the epilogue in a tile's last step)
                } else {
                    a_piece(t - 8);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!last_of_tile) b_advance();
            a_ahead = a_advance();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            ca_slot = na_slot;
            cb_slot = nb_slot;
        };
        k_step(true, false);
        for (int kt = 1; kt < nk - 1; ++kt) k_step(false, false);
        const int stage_slot = cb_slot;   // B slot of the tile's last stage = epilogue staging area after its barrier
        k_step(false, true);
        stamp(ti, 2);

        // ---- epilogue of (m0, n0) ----
        const int m_base = m0 + wm * 128, n_base = n0 + wn * 128;
        {   // bias joins the accumulators first (its registers are not live during the staging)
            float4 bv[4][4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g)   // (a null bias has a zero-length descriptor: out-of-range loads return 0)
                    bv[ni][g] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        bias_rs, hi * 16, __builtin_amdgcn_readfirstlane((n_base + ni * 32 + 8 * g) * 4), 0));
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        acc[mi][ni][g * 4 + 0] += bv[ni][g].x; acc[mi][ni][g * 4 + 1] += bv[ni][g].y;
                        acc[mi][ni][g * 4 + 2] += bv[ni][g].z; acc[mi][ni][g * 4 + 3] += bv[ni][g].w;
                    }
        }
        // side input (fp32 residual / bf16 h_pre) in the coalesced store pattern, prefetched SIDE_DEPTH 32x32 sub-tiles
        // ahead; sub-tile s = 4*mi + ni lives in side[s % SIDE_DEPTH]
        constexpr int SIDE_DEPTH = 4;
        u32x4 side[SIDE_DEPTH][4];
        auto load_side = [&](int sub) {
            const int mi = sub >> 2, ni = sub & 3;
            if (EPI == EPI_F32_RESID) {
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 4);
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    side[sub % SIDE_DEPTH][it] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, st32_loff, so + it * 32 * ldo, 0);
            } else {
                const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 2);
                side[sub % SIDE_DEPTH][0] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, h16_loff, so, 0);
                side[sub % SIDE_DEPTH][1] = __builtin_amdgcn_raw_buffer_load_b128(h_rs, h16_loff, so + 32 * ldo, 0);
            }
        };
        if (EPI == EPI_F32_RESID || EPI == EPI_BF16_DACT) {
#pragma unroll
            for (int sub = 0; sub < SIDE_DEPTH; ++sub) load_side(sub);
        }

        const unsigned ebuf = lds_base + PB_BASE + stage_slot * PB_SLOT + w * Q_EPI_WAVE;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if (!OUT_F32 && EPI != EPI_BF16_DACT) {
                // bf16 output(s): two 64-column halves, each staged as 32 rows x 128 B (8-B chunk index XOR (row & 15)) in
                // its own 4 KiB of the wave's staging area, then 4 stores of 8 full 128-B rows
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const unsigned eb = ebuf + nh * 4096;
                    const unsigned w16_pre = eb + l31 * 128 + ((hi ^ (l31 & 15)) << 3);
                    const unsigned r16_a = eb + r0 * 128 + (((lane & 7) ^ (r0 >> 1)) << 4);
                    const unsigned r16_b = eb + r0 * 128 + (((lane & 7) ^ (r0 >> 1) ^ 4) << 4);
                    const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + nh * 64) * 2);
                    auto stage_flush = [&](__amdgpu_buffer_rsrc_t rs, int what) {   // what: 0 value, 1 act(value), 2 act'(value)
#pragma unroll
                        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int ni = nh * 2 + nj;
                                float v[4] = {acc[mi][ni][g * 4 + 0], acc[mi][ni][g * 4 + 1], acc[mi][ni][g * 4 + 2],
                                              acc[mi][ni][g * 4 + 3]};
                                bf16x4 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                float av, dv;
                                actp_pair<ACT>(v[e], av, dv);
                                o[e] = (bf16_t)(what == 0 ? v[e] : what == 1 ? av : dv);
                            }
                                lds_w64(w16_pre ^ ((nj * 8 + 2 * g) << 3), __builtin_bit_cast(u32x2, o));
                            }
                        u32x4 t0 = lds_r128<0>(r16_a), t1 = lds_r128<8 * 128>(r16_b), t2 = lds_r128<16 * 128>(r16_a),
                              t3 = lds_r128<24 * 128>(r16_b);
                        lds_wait();
                        if (r0 & 1) {   // a lane's 16 B cover two 8-B chunks, swapped when its row is odd
                            t0 = __builtin_shufflevector(t0, t0, 2, 3, 0, 1); t1 = __builtin_shufflevector(t1, t1, 2, 3, 0, 1);
                            t2 = __builtin_shufflevector(t2, t2, 2, 3, 0, 1); t3 = __builtin_shufflevector(t3, t3, 2, 3, 0, 1);
                        }
                        store16(t0, rs, st16_loff, so);
                        store16(t1, rs, st16_loff, so + 16 * ldo);
                        store16(t2, rs, st16_loff, so + 32 * ldo);
                        store16(t3, rs, st16_loff, so + 48 * ldo);
                    };
                    stage_flush(EPI == EPI_BF16_ACT ? pre_rs : o_rs, EPI == EPI_BF16_ACT ? 2 : 0);
                    if (EPI == EPI_BF16_ACT) stage_flush(o_rs, 1);
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[mi][nh * 2 + nj][e] = 0.0f;
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    // fp32 staging of one 32x32 sub-tile (128-B rows, 16-B chunk index XOR (row & 7)); consecutive
                    // sub-tiles alternate between the two 4 KiB halves of the wave's staging area
                    const int sub = mi * 4 + ni;
                    const unsigned eb = ebuf + (sub & 1) * 4096;
                    const unsigned w32_pre = eb + l31 * 128 + ((hi ^ (l31 & 7)) << 4);
                    const unsigned r32 = eb + r0 * 128 + (((lane & 7) ^ r0) << 4);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 v = make_float4(acc[mi][ni][g * 4 + 0], acc[mi][ni][g * 4 + 1],
                                                     acc[mi][ni][g * 4 + 2], acc[mi][ni][g * 4 + 3]);
                        lds_w128(w32_pre ^ (g << 5), __builtin_bit_cast(u32x4, v));
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
                    if (EPI == EPI_BF16_DACT) {
                        const int so = __builtin_amdgcn_readfirstlane(((m_base + mi * 32) * ldo + n_base + ni * 32) * 2);
                        const int drow = lane >> 2, dq = (lane & 3) * 2;
                        const unsigned dA = eb + drow * 128 + ((dq ^ (drow & 7)) << 4);
                        const unsigned dB = eb + drow * 128 + (((dq + 1) ^ (drow & 7)) << 4);
                        const u32x4 fa0 = lds_r128<0>(dA), fb0 = lds_r128<0>(dB), fa1 = lds_r128<16 * 128>(dA),
                                    fb1 = lds_r128<16 * 128>(dB);
                        lds_wait();
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const float4 a = __builtin_bit_cast(float4, half ? fa1 : fa0);
                            const float4 b = __builtin_bit_cast(float4, half ? fb1 : fb0);
                            const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                            const bf16x8 h8 = __builtin_bit_cast(bf16x8, side[sub % SIDE_DEPTH][half]);
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(f[e] * (float)h8[e]);   // h8 = act'(h) stored by the forward
                            store16(__builtin_bit_cast(u32x4, o), o_rs, h16_loff, so + half * 32 * ldo);
                        }
                    } else {
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
                            store16(t[it], o_rs, st32_loff, so + it * 32 * ldo);
                        }
                    }
                    if ((EPI == EPI_F32_RESID || EPI == EPI_BF16_DACT) && sub + SIDE_DEPTH < 16) load_side(sub + SIDE_DEPTH);
                }
            }
        }
        stamp(ti, 3);
        if (!last_tile) {
            // every wave is done with the staging slot: request the B half that was held back
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) b_piece(j);
            b_advance();
        }
    }
    if (p.trace && w == 0 && lane == 0) {
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 2] = __builtin_amdgcn_s_memtime();
        p.trace[((long)blockIdx.x * 8 + 7) * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

extern int g_persist_ablate;
extern unsigned long long* g_persist_trace;

template <int EPI, int ACT, int ABL>
static int launch_256q_abl(const GemmBf16& p, int tiles_m, int tiles_n, hipStream_t s) {
    static bool attr_set = false;
    const int lds_bytes = 3 * PA_SLOT + 2 * PB_SLOT;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt_256q_kernel<EPI, ACT, ABL>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return fail(RVLM_ERR_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(e));
        attr_set = true;
    }
    const int grid = std::min(tiles_m * tiles_n, 256);
    GemmBf16 q = p;
    q.trace = g_persist_trace;
    hipLaunchKernelGGL((gemm_bf16_nt_256q_kernel<EPI, ACT, ABL>), dim3(grid), dim3(256), lds_bytes, s, q, tiles_m, tiles_n);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}
template <int EPI, int ACT>
static int launch_256q_act(const GemmBf16& p, int tiles_m, int tiles_n, hipStream_t s) {
    if constexpr (EPI == EPI_BF16) {   // the timing experiments exist for the plain epilogue only
        switch (g_persist_ablate) {
            case 1: return launch_256q_abl<EPI, ACT, 1>(p, tiles_m, tiles_n, s);
            case 2: return launch_256q_abl<EPI, ACT, 2>(p, tiles_m, tiles_n, s);
            case 4: return launch_256q_abl<EPI, ACT, 4>(p, tiles_m, tiles_n, s);
            case 5: return launch_256q_abl<EPI, ACT, 5>(p, tiles_m, tiles_n, s);
            default: break;
        }
    }
    return launch_256q_abl<EPI, ACT, 0>(p, tiles_m, tiles_n, s);
}
template <int EPI>
static int launch_256q(const GemmBf16& p, int tiles_m, int tiles_n, hipStream_t s) {
    if constexpr (EPI == EPI_BF16_ACT || EPI == EPI_BF16_DACT) {
        if (p.act != RVLM_ACT_QUICK_GELU) return launch_256q_act<EPI, RVLM_ACT_GELU>(p, tiles_m, tiles_n, s);
    }
    return launch_256q_act<EPI, RVLM_ACT_QUICK_GELU>(p, tiles_m, tiles_n, s);
}

// rows [0, 256*floor(M/256)) of the problem; *rows_done = 0 when the shape does not qualify
int gemm_bf16_nt_256q(const GemmBf16& p, int* rows_done, hipStream_t s) {
    *rows_done = 0;
    if (p.M < P_M || p.N % P_N != 0 || p.K % (2 * P_K) != 0 || p.K < 2 * P_K) return RVLM_OK;
    const long lim = 1L << 31;
    if ((long)p.M * p.lda * 2 >= lim || (long)p.N * p.ldb * 2 >= lim || (long)p.M * p.ldo * 4 >= lim) return RVLM_OK;
    GemmBf16 q = p;
    const int tiles_m = p.M / P_M, tiles_n = p.N / P_N;
    q.M = tiles_m * P_M;
    if (q.epi == EPI_F32_RESID && !q.residual) q.epi = EPI_F32;
    int rc;
    switch (q.epi) {
        case EPI_BF16: rc = launch_256q<EPI_BF16>(q, tiles_m, tiles_n, s); break;
        case EPI_F32_RESID: rc = launch_256q<EPI_F32_RESID>(q, tiles_m, tiles_n, s); break;
        case EPI_BF16_ACT: rc = launch_256q<EPI_BF16_ACT>(q, tiles_m, tiles_n, s); break;
        case EPI_BF16_DACT: rc = launch_256q<EPI_BF16_DACT>(q, tiles_m, tiles_n, s); break;
        case EPI_F32: rc = launch_256q<EPI_F32>(q, tiles_m, tiles_n, s); break;
        default: return fail(RVLM_ERR_ARG, "gemm_bf16_nt_256q: unknown epilogue");
    }
    if (rc) return rc;
    *rows_done = q.M;
    return RVLM_OK;
}

}  // namespace rvlm
