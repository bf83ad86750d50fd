// Register-only MFMA streams for a power / clock comparison of the two bf16 MFMA shapes on gfx950:
//   kind 0: v_mfma_f32_32x32x16_bf16, 8 waves per CU, wave tile 128 x 64  (4 x 2 accumulator tiles = 128 registers) - the shipped GEMM
//   kind 1: v_mfma_f32_16x16x32_bf16, 8 waves per CU, wave tile 128 x 64  (8 x 4 accumulator tiles = 128 registers)
//   kind 2: v_mfma_f32_32x32x16_bf16, 4 waves per CU, wave tile 128 x 128 (4 x 4 tiles = 256 registers)
//   kind 3: v_mfma_f32_16x16x32_bf16, 4 waves per CU, wave tile 128 x 128 (8 x 8 tiles = 256 registers) - hipBLASLt's shape
// Every variant issues the MFMAs of one 64-deep K-step per loop iteration (equal FLOPs per wave-tile element), operands are
// random bf16 fragments held in registers (no LDS, no memory in the loop), one workgroup per CU.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_power_probe.hip -o scripts/probes/mfma_power_probe.bin
// run:   mfma_power_probe.bin <kind> <seconds>   -> prints TFLOP/s (python scripts/mfma_power.py samples rocm-smi beside it)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int KIND>
__global__ void __launch_bounds__((KIND < 2 ? 8 : 4) * 64)
probe(const bf16x8* __restrict__ seed, float* __restrict__ out, int iters) {
    constexpr int NT = KIND < 2 ? 2 : 4;              // 32-column (kind 0, 2) / 16-column groups handled below
    const int lane = threadIdx.x & 63;
    float sink = 0.0f;
    if (KIND == 0 || KIND == 2) {
        // per 16-deep slice: 4 A fragments (128 rows), NT B fragments (NT x 32 columns); 4 slices per K-step
        bf16x8 a[4][4], b[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[s][i] = seed[(s * 8 + i) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[s][j] = seed[(s * 8 + 4 + j) * 64 + lane];
        }
        f32x16 acc[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sink += acc[i][j][e];
    } else {
        // per 32-deep slice: 8 A fragments (128 rows), 2 NT B fragments (2 NT x 16 columns); 2 slices per K-step
        constexpr int NB = 2 * NT;
        bf16x8 a[2][8], b[2][NB];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[s][i] = seed[(s * 16 + i) * 64 + lane];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[s][j] = seed[(s * 16 + 8 + j) * 64 + lane];
        }
        f32x4 acc[8][NB];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    if (sink == 123456.789f) out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    const int waves = kind < 2 ? 8 : 4;
    const int ncol = kind < 2 ? 64 : 128;
    std::vector<unsigned short> h(64 * 64 * 8);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 31) << 15)); }   // random sign / mantissa, |v| in [1, 2)... bf16 bits 0x3f80 +-: close enough to random data
    for (auto& v : h) v = (unsigned short)((v & 0x8000u) | 0x3f00u | (v & 0x00ffu));     // bf16 in +-[0.5, 1)
    bf16x8* seed; float* out;
    CK(hipMalloc(&seed, h.size() * 2)); CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMemcpy(seed, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int iters = 20000;      // K-steps per launch: 20000 x 64 deep
    auto launch = [&]() {
        switch (kind) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, seed, out, iters); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, seed, out, iters); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, seed, out, iters); break;
            default: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, seed, out, iters); break;
        }
    };
    launch(); CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    do {
        launch(); CK(hipDeviceSynchronize()); ++n;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < secs);
    const double flops = 2.0 * 128 * ncol * 64 * (double)iters * waves * 256 * n;
    printf("kind %d (%s, %d waves per CU, wave tile 128 x %d): %.1f TFLOP/s, %.0f cycles-equivalent per K-step at 2.4 GHz\n", kind,
           (kind & 1) ? "16x16x32" : "32x32x16", waves, ncol, flops / el / 1e12, el / n / iters * 2.4e9);
    return 0;
}
