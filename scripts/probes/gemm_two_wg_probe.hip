// Probe for VERDICT r5 item 1: a bf16 GEMM with TWO workgroups per CU, so that the hardware scheduler runs one workgroup's
// epilogue (and pipeline fill) under the other's mainloop - instead of the shipped ONE 8-wave 256 x 256 workgroup per CU whose fused
// epilogues stand in the way of the next tile's MFMAs (robustvlm_amd/csrc/gemm_bf16_256p.hip).
//
// The form: 4-wave workgroups (one wave per SIMD and workgroup, the SIMD's second wave belongs to the OTHER workgroup), tile
// 128 x 256, the shipped wave tile 128 x 64 = 8 x 4 tiles of v_mfma_f32_16x16x32_bf16, 256 VGPRs.  LDS budget of HALF a CU, 80 KiB:
// a BK = 64 stage of this tile is A 16 KiB + B 32 KiB = 48 KiB - two stages do not fit, so the ring is 3 stages of BK = 32
// (A 8 KiB + B 16 KiB, 64-byte LDS rows, 16-B chunk slot XOR S[(row >> 2) & 3], S = {0, 3, 2, 1}: conflict-free under
// ds_read_b128's four 16-lane groups) = 72 KiB, one barrier per stage, the stage two ahead requested right behind it.
// Operands arrive by `buffer_load ... lds` (16 rows x 64 B per instruction), 6 requests per wave and stage = 1.5x the shipped
// kernel's DMA bytes per FLOP (its A rows are shared by two wave rows, these are not).  One tile per workgroup, grid = tiles:
// the dispatcher refills a CU's half as soon as a workgroup leaves, which desynchronises the two halves by itself.
// The epilogue is the plain one (bf16 store, 8 bytes per lane and accumulator tile): NO fused epilogue is built - the question
// here is what the mainloop of this form holds next to the shipped kernel's PLAIN rate (1 230-1 340 TFLOP/s in the pipeline) on the
// same box, same shapes, same process (librvlm.so's test hook is called from this binary).  Results are checked against a host
// dot product on sampled elements.
//
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/gemm_two_wg_probe.hip -o scripts/probes/gemm_two_wg_probe.bin -ldl
// run:   scripts/probes/gemm_two_wg_probe.bin [path/to/librvlm.so]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int T_M = 128, T_N = 256, BK = 32, STAGES = 3;
constexpr int A_BYTES = T_M * BK * 2, B_BYTES = T_N * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;    // 8 + 16 KiB

__device__ __forceinline__ int swz4(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// GROUP_M = 8 tile order inside each XCD's share (blockIdx % 8 = XCD): consecutive workgroups of an XCD share B panels in its L2
__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int ntiles = tiles_m * tiles_n, q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    constexpr int GM = 8;
    const int gsz = GM * tiles_n, first = (t / gsz) * GM, gm = min(tiles_m - first, GM);
    tm = first + (t % gsz) % gm;
    tn = (t % gsz) / gm;
}

__global__ void __launch_bounds__(256, 2)
gemm_two_wg_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ Bw, __bf16* __restrict__ C, int M, int N, int K,
                   int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tm, tn;
    tile_of(blockIdx.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * T_M, n0 = tn * T_N, nk = K / BK;
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A), 0, (unsigned)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Bw), 0, (unsigned)((size_t)N * K * 2), 0x00020000);
    // one DMA instruction = 16 rows x 64 B: lane -> row (lane >> 2), LDS slot (lane & 3) holds logical chunk slot ^ swz4(row)
    // (a piece starts at a multiple of 16 rows, so the key only depends on the lane)
    const int prow = lane >> 2, pchunk = (lane & 3) ^ swz4(prow);
    const int a_loff = (prow * K + pchunk * 8) * 2, b_loff = a_loff;
    // wave w requests A pieces 2 w, 2 w + 1 (rows 32 w ..) and B pieces 4 w .. 4 w + 3 (rows 64 w ..)
    const int a_soff = ((m0 + 32 * w) * K) * 2, b_soff = ((n0 + 64 * w) * K) * 2;
    auto issue = [&](int kt) {
        const int slot = kt % STAGES;
        __attribute__((address_space(3))) char* base = (__attribute__((address_space(3))) char*)lds + slot * STAGE_BYTES;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(base + (2 * w + p) * 1024), 16, a_loff,
                                                     __builtin_amdgcn_readfirstlane(a_soff + p * 16 * K * 2 + kt * BK * 2), 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_ptr_t)(base + A_BYTES + (4 * w + p) * 1024), 16, b_loff,
                                                     __builtin_amdgcn_readfirstlane(b_soff + p * 16 * K * 2 + kt * BK * 2), 0, 0);
    };
    // fragment reads: lane (i16, G) reads row 16 t + i16, logical chunk G (k = 8 G .. 8 G + 7 of the 32-deep stage)
    const int i16 = lane & 15, G = lane >> 4;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned fa = lds_base + i16 * 64 + ((G ^ swz4(i16)) << 4);                       // A rows 16 t + i16: + t * 1024
    const unsigned fb = fa + A_BYTES + w * 4096;                                            // B rows 64 w + 16 t + i16
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    issue(0);
    if (nk > 1) issue(1);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed for this wave when at most the 6 requests of stage kt + 1 are still in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // ... for every wave; and every wave is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) issue(kt + 2);                     // into the slot of stage kt - 1
        const unsigned so = (unsigned)((kt % STAGES) * STAGE_BYTES);
        i32x4 x[8], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[t]) : "v"(fb + so + t * 1024));
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(x[t]) : "v"(fa + so + t * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, x[q]),
                                                                   acc[q][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // the second half's X fragments go out behind the first MFMAs
                if (q == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(x[4 + nt]) : "v"(fa + so + (4 + nt) * 1024));
                __builtin_amdgcn_sched_barrier(0);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 4; q < 8; ++q)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[nt]), __builtin_bit_cast(bf16x8, x[q]),
                                                                   acc[q][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // plain epilogue: tile (mt, nt), lane (i16, G) holds row m = 16 mt + i16, columns n = 16 nt + 4 G + 0..3
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[mt][nt][e];
            *(bf16x4*)(C + (size_t)(m0 + 16 * mt + i16) * N + n0 + 64 * w + 16 * nt + 4 * G) = o;
        }
}

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

typedef int (*gemm_hook_t)(const void*, long, const void*, long, int, int, int, int, int, const void*, void*, long, void*, const void*,
                           const void*, int, void*);
typedef const char* (*err_t)();

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "robustvlm_amd/librvlm.so";
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    gemm_hook_t hook = h ? (gemm_hook_t)dlsym(h, "rvlm_k_gemm_bf16_nt") : nullptr;
    err_t last_error = h ? (err_t)dlsym(h, "rvlm_last_error") : nullptr;
    if (!hook) fprintf(stderr, "(librvlm.so not loaded from %s: %s - the shipped kernel's arm is skipped)\n", libpath, dlerror());
    CHECK(hipFuncSetAttribute((const void*)gemm_two_wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * STAGE_BYTES));
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_two_wg_kernel, 256, STAGES * STAGE_BYTES));
    printf("two-workgroup form: 4 waves, tile %d x %d, BK %d, %d stages = %d KiB of LDS, occupancy %d workgroups per CU\n", T_M, T_N, BK, STAGES,
           STAGES * STAGE_BYTES / 1024, occ);
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"qkv fwd", 32768, 3072, 1024}, {"out / fc2-like K=1024 N=1024", 32768, 1024, 1024}, {"fc1 fwd", 32768, 4096, 1024},
                            {"fc2 fwd", 32768, 1024, 4096}, {"cube 8192", 8192, 8192, 8192}};
    hipStream_t s; CHECK(hipStreamCreate(&s));
    for (const Shape& sh : shapes) {
        const size_t na = (size_t)sh.M * sh.K, nb = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
        std::vector<uint16_t> ha(na), hb(nb);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : ha) v = f2bf(rnd());
        for (auto& v : hb) v = f2bf(rnd() * 0.125f);
        void *dA, *dB, *dC, *dC2;
        CHECK(hipMalloc(&dA, na * 2)); CHECK(hipMalloc(&dB, nb * 2)); CHECK(hipMalloc(&dC, nc * 2)); CHECK(hipMalloc(&dC2, nc * 2));
        CHECK(hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
        CHECK(hipMemset(dC, 0, nc * 2)); CHECK(hipMemset(dC2, 0, nc * 2));
        const int tiles_m = sh.M / T_M, tiles_n = sh.N / T_N;
        auto run2 = [&]() {
            hipLaunchKernelGGL(gemm_two_wg_kernel, dim3(tiles_m * tiles_n), dim3(256), STAGES * STAGE_BYTES, s, (const __bf16*)dA, (const __bf16*)dB,
                               (__bf16*)dC, sh.M, sh.N, sh.K, tiles_m, tiles_n);
        };
        auto run1 = [&]() {
            int rc = hook(dA, sh.K, dB, sh.K, sh.M, sh.N, sh.K, sh.M, 0, nullptr, dC2, sh.N, nullptr, nullptr, nullptr, 0, (void*)s);
            if (rc) { fprintf(stderr, "rvlm_k_gemm_bf16_nt: %s\n", last_error ? last_error() : "?"); exit(1); }
        };
        run2(); CHECK(hipStreamSynchronize(s));
        // check sampled elements against the host (fp64 dot products of the bf16 operands; the result is bf16-rounded)
        std::vector<uint16_t> hc(nc);
        CHECK(hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost));
        double worst = 0.0;
        for (int i = 0; i < 256; ++i) {
            const size_t m = ((size_t)i * 7919u * 131u) % sh.M, n = ((size_t)i * 104729u + 17u) % sh.N;
            double ref = 0.0;
            for (int k = 0; k < sh.K; ++k) ref += (double)bf2f(ha[m * sh.K + k]) * (double)bf2f(hb[n * sh.K + k]);
            worst = fmax(worst, fabs(bf2f(hc[m * sh.N + n]) - ref) / (fabs(ref) + 0.05 * sqrt((double)sh.K) * 0.036));
        }
        if (hook) {
            run1(); CHECK(hipStreamSynchronize(s));
            std::vector<uint16_t> hc2(nc);
            CHECK(hipMemcpy(hc2.data(), dC2, nc * 2, hipMemcpyDeviceToHost));
            size_t diff = 0;
            for (size_t i = 0; i < nc; i += 97) diff += hc[i] != hc2[i];
            printf("%-30s M %d N %d K %d: sampled error vs host %.2e (bf16 rounding ~4e-3); elements differing from the shipped kernel's: %zu of %zu sampled\n",
                   sh.name, sh.M, sh.N, sh.K, worst, diff, nc / 97);
        } else
            printf("%-30s M %d N %d K %d: sampled error vs host %.2e\n", sh.name, sh.M, sh.N, sh.K, worst);
        // timing: clock warm-up, then three alternations of 30 launches each
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 60; ++i) run2();
        CHECK(hipStreamSynchronize(s));
        const double flop = 2.0 * sh.M * sh.N * sh.K;
        for (int rep = 0; rep < 3; ++rep) {
            float ms2 = 0.f, ms1 = 0.f;
            CHECK(hipEventRecord(e0, s)); for (int i = 0; i < 30; ++i) run2(); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms2, e0, e1));
            if (hook) {
                CHECK(hipEventRecord(e0, s)); for (int i = 0; i < 30; ++i) run1(); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms1, e0, e1));
            }
            printf("   rep %d: two-workgroup form %7.1f us = %6.0f TFLOP/s", rep, ms2 / 30 * 1e3, flop / (ms2 / 30 * 1e-3) / 1e12);
            if (hook) printf("   | shipped 256p (plain epilogue) %7.1f us = %6.0f TFLOP/s", ms1 / 30 * 1e3, flop / (ms1 / 30 * 1e-3) / 1e12);
            printf("\n");
        }
        fflush(stdout);
        CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dC2));
    }
    return 0;
}
