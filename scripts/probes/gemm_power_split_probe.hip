// Where the persistent GEMM's energy goes: the 16x16x32 MFMA stream of its K-step (8 waves per CU, wave tile 128 x 64, 64 MFMAs per
// wave and K-step) alone and with the kernel's other two per-K-step activities added one at a time, under the socket power cap:
//   kind 0: MFMAs only (operands held in registers)                                   [= kind 1 of mfma_power_probe.hip]
//   kind 1: + the fragment reads: 24 ds_read_b128 per wave and K-step (192 KiB of LDS reads per CU and K-step), consumed by the MFMAs
//   kind 2: + the operand stream: 8 x 1 KiB `global_load ... lds` per wave and K-step (64 KiB per CU) from an L2-resident buffer,
//             one s_waitcnt vmcnt(0) + barrier per K-step
//   kind 3: MFMAs + operand stream, no fragment reads
//   kinds 4-7 (round 5, VERDICT r4 item 6: fewer LDS bytes per FLOP?): FOUR waves per CU with 128 x 128 wave tiles (one wave per
//             SIMD, 256 accumulator registers; 32 ds_read_b128 per wave and K-step = 128 KiB per CU, a third fewer than the
//             8-wave form): 4 = v_mfma_f32_32x32x16_bf16 only, 5 = + its fragment reads, 6 = v_mfma_f32_16x16x32_bf16 only, 7 = + reads
// One workgroup per CU, 160 KiB of LDS like the kernel.  python scripts/gemm_power_split.py samples rocm-smi beside it.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/gemm_power_split_probe.hip -o scripts/probes/gemm_power_split_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int KIND>
__global__ void __launch_bounds__(512)
probe(const bf16x8* __restrict__ seed, const char* __restrict__ stream, unsigned stream_mask, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr bool READS = KIND == 1 || KIND == 2, DMA = KIND == 2 || KIND == 3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // fill the LDS once with operand-like data (random bf16 in +-[0.5, 1))
    for (int i = threadIdx.x; i < 160 * 1024 / 16; i += 512) ((bf16x8*)lds)[i] = seed[i & 4095];
    __syncthreads();
    i32x4 a[2][8], b[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[s][i] = __builtin_bit_cast(i32x4, seed[(s * 16 + i) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[s][j] = __builtin_bit_cast(i32x4, seed[(s * 16 + 8 + j) * 64 + lane]);
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // fragment read addresses: the kernel's pattern (row = lane & 15, 16-B chunk = lane >> 4, XOR swizzle), tiles 2 KiB apart
    const unsigned fa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (w >> 2) * 16384 + (lane & 15) * 128 +
                        ((((lane >> 4)) ^ (((lane & 15) >> 1) & 7)) << 4);
    const unsigned fb = fa + 96 * 1024 + (w & 3) * 8192 - (w >> 2) * 16384;
    unsigned goff = (blockIdx.x * 8 + w) * 8192 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = (it % 3) * 32768, bslot = (it & 1) * 32768;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the 32 MFMAs of slice s; with READS, the 12 fragment reads of the other slice go out behind the first MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[s][j]), __builtin_bit_cast(bf16x8, a[s][i]),
                                                                      acc[i][j], 0, 0, 0);
                    if (READS) {
                        const int r = i * 4 + j;
                        const unsigned aa = fa + slot + (s ^ 1) * 64, bb = fb + bslot + (s ^ 1) * 64;
                        if (r < 8) asm volatile("ds_read_b128 %0, %1" : "=v"(a[s ^ 1][r]) : "v"(aa + r * 2048));
                        else if (r < 12) asm volatile("ds_read_b128 %0, %1" : "=v"(b[s ^ 1][r - 8]) : "v"(bb + (r - 8) * 2048));
                    }
                }
            if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (DMA) {
            // this wave's eighth of a 64-KiB stage: 8 pieces of 1 KiB into the ring slot that is not being read
            __attribute__((address_space(3))) char* dst = (__attribute__((address_space(3))) char*)lds + ((it + 1) % 3) * 32768 + w * 8192;
#pragma unroll
            for (int p = 0; p < 8; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + ((goff + p * 1024) & stream_mask)),
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
            goff += 256 * 8 * 8192;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    float sink = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sink == 123456.789f) out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}


typedef __attribute__((ext_vector_type(16))) float f32x16;
// four waves, wave tile 128 x 128.  M32: 4 x 4 tiles of 32x32x16 (4 slices of 16 per K-step, 4 + 4 fragments per slice);
// else 8 x 8 tiles of 16x16x32 (2 slices of 32, 8 + 8 fragments per slice).  The next slice's fragments go out behind the first MFMAs.
template <bool M32, bool READS>
__global__ void __launch_bounds__(256)
probe4(const bf16x8* __restrict__ seed, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 160 * 1024 / 16; i += 256) ((bf16x8*)lds)[i] = seed[i & 4095];
    __syncthreads();
    constexpr int NF = M32 ? 4 : 8;            // fragments per operand and slice
    constexpr int NS = M32 ? 4 : 2;            // slices per K-step
    i32x4 a[2][NF], b[2][NF];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            a[s][i] = __builtin_bit_cast(i32x4, seed[(s * 16 + i) * 64 + lane]);
            b[s][i] = __builtin_bit_cast(i32x4, seed[(s * 16 + 8 + i) * 64 + lane]);
        }
    f32x16 acc32[M32 ? 4 : 1][M32 ? 4 : 1];
    f32x4 acc16[M32 ? 1 : 8][M32 ? 1 : 8];
#pragma unroll
    for (int i = 0; i < (M32 ? 4 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M32 ? 4 : 1); ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.0f;
#pragma unroll
    for (int i = 0; i < (M32 ? 1 : 8); ++i)
#pragma unroll
        for (int j = 0; j < (M32 ? 1 : 8); ++j) acc16[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const unsigned fa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (w >> 1) * 16384 + (lane & 31) * 128 +
                        ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) << 4);
    const unsigned fb = fa + 96 * 1024 + (w & 1) * 16384 - (w >> 1) * 16384;
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = (it % 3) * 32768, bslot = (it & 1) * 32768;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if (M32)
                        acc32[M32 ? i : 0][M32 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, b[cur][j]), __builtin_bit_cast(bf16x8, a[cur][i]), acc32[M32 ? i : 0][M32 ? j : 0], 0, 0, 0);
                    else
                        acc16[M32 ? 0 : i][M32 ? 0 : j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, b[cur][j]), __builtin_bit_cast(bf16x8, a[cur][i]), acc16[M32 ? 0 : i][M32 ? 0 : j], 0, 0, 0);
                    if (READS) {
                        const int r = i * NF + j;
                        const unsigned aa = fa + slot + ((s + 1) % NS) * 32, bb = fb + bslot + ((s + 1) % NS) * 32;
                        if (r < NF) asm volatile("ds_read_b128 %0, %1" : "=v"(a[nxt][r]) : "v"(aa + r * (M32 ? 4096 : 2048)));
                        else if (r < 2 * NF) asm volatile("ds_read_b128 %0, %1" : "=v"(b[nxt][r - NF]) : "v"(bb + (r - NF) * (M32 ? 4096 : 2048)));
                    }
                }
            if (READS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float sink = 0.0f;
#pragma unroll
    for (int i = 0; i < (M32 ? 4 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M32 ? 4 : 1); ++j) sink += acc32[i][j][0] + acc32[i][j][7] + acc32[i][j][15];
#pragma unroll
    for (int i = 0; i < (M32 ? 1 : 8); ++i)
#pragma unroll
        for (int j = 0; j < (M32 ? 1 : 8); ++j) sink += acc16[i][j][0] + acc16[i][j][3];
    if (sink == 123456.789f) out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    const size_t stream_bytes = (size_t)(argc > 3 ? atoi(argv[3]) : 16) << 20;      // footprint of the operand stream (MiB, power of two)
    std::vector<unsigned short> h(4096 * 8);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x8000u) | 0x3f00u | ((x >> 8) & 0x00ffu)); }
    bf16x8* seed; float* out; char* stream;
    CK(hipMalloc(&seed, h.size() * 2)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&stream, stream_bytes));
    CK(hipMemcpy(seed, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (size_t o = 0; o < stream_bytes; o += h.size() * 2) CK(hipMemcpy(stream + o, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int iters = 20000;
    const int lds_bytes = 160 * 1024;
    auto launch = [&]() {
        switch (kind) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), lds_bytes, 0, seed, stream, (unsigned)(stream_bytes - 1), out, iters); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), lds_bytes, 0, seed, stream, (unsigned)(stream_bytes - 1), out, iters); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), lds_bytes, 0, seed, stream, (unsigned)(stream_bytes - 1), out, iters); break;
            case 4: hipLaunchKernelGGL((probe4<true, false>), dim3(256), dim3(256), lds_bytes, 0, seed, out, iters); break;
            case 5: hipLaunchKernelGGL((probe4<true, true>), dim3(256), dim3(256), lds_bytes, 0, seed, out, iters); break;
            case 6: hipLaunchKernelGGL((probe4<false, false>), dim3(256), dim3(256), lds_bytes, 0, seed, out, iters); break;
            case 7: hipLaunchKernelGGL((probe4<false, true>), dim3(256), dim3(256), lds_bytes, 0, seed, out, iters); break;
            default: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), lds_bytes, 0, seed, stream, (unsigned)(stream_bytes - 1), out, iters); break;
        }
    };
    CK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe4<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe4<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe4<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CK(hipFuncSetAttribute((const void*)probe4<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    launch(); CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    do {
        launch(); CK(hipDeviceSynchronize()); ++n;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    const double flops = 2.0 * 128 * 64 * 64 * (double)iters * 8 * 256 * n;     // (4 waves x 128 x 128 = 8 waves x 128 x 64)
    const char* names[8] = {"MFMA only", "MFMA + fragment reads", "MFMA + fragment reads + operand DMA", "MFMA + operand DMA",
                            "4 waves x 128x128, 32x32x16 MFMA only", "4 waves x 128x128, 32x32x16 MFMA + fragment reads",
                            "4 waves x 128x128, 16x16x32 MFMA only", "4 waves x 128x128, 16x16x32 MFMA + fragment reads"};
    printf("kind %d (%s; stream footprint %zu MiB): %.1f TFLOP/s, %.0f ns per K-step\n", kind, names[kind & 7], stream_bytes >> 20,
           flops / el / 1e12, el / n / iters * 1e9);
    return 0;
}
