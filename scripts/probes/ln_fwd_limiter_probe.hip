// LayerNorm forward: what limits it?  (0.65 of 8 TB/s against the backward's 0.79 on "the same access pattern".)
//
// The shipped layernorm_fwd8 (csrc/vit_kernels.hip) reads the fp32 residual stream (4 B per element) and writes the bf16 operand of
// the next GEMM (2 B): 6 B per element, one wave per row, two row reductions between the loads and the stores.  This probe times,
// on the encoder's shape (M = 32 896 rows, W = 1024) over 24 distinct buffer sets (nothing re-read from a cache):
//   kind 0: the shipped kernel's code (row statistics, gamma / beta, bf16 store)
//   kind 1: the same loads and stores with NO reductions and no gamma / beta (a converting copy: the memory system's answer for a
//           4 B read + 2 B write stream in this geometry)
//   kind 2: kind 0 with gamma / beta requested next to x (before the reductions instead of behind them)
//   kind 3: kind 0, two rows per wave (the second row's loads in flight under the first row's reductions)
//   kind 4: a float4 -> float4 copy of the same number of BYTES per row (the guide's 6.29 TB/s reference pattern, 1 : 1 read : write)
// If kind 1 runs at kind 0's rate the kernel is at what HBM gives a 2 : 1 stream and only fewer bytes help.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probes/ln_fwd_limiter_probe.hip -o scripts/probes/ln_fwd_limiter_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __bf16 bf16_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int KIND>
__global__ void __launch_bounds__(256)
ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y,
       float* __restrict__ mean, float* __restrict__ rstd, int M) {
    constexpr int NV = 2, W = NV * 512, ROWS = KIND == 3 ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (row0 >= M) return;
    if (KIND == 4) {          // 6 KiB per row moved as 3 KiB in + 3 KiB out: 3 float4 per lane
        const float4* s = (const float4*)(x + (long)row0 * W);
        float4* d = (float4*)(y + (long)row0 * W);       // (the bf16 buffer, 2 KiB per row) + the upper part of x's row as overflow
        float4 a = s[lane], b = s[64 + lane], c = s[128 + lane];
        float4* d2 = (float4*)(const_cast<float*>(x) + (long)row0 * W + 768);
        d[lane] = a; d[64 + lane] = b; d2[lane] = c;
        return;
    }
    float v[ROWS][NV][8], g[NV][8], b[NV][8];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            const float* xr = x + (long)(row0 + r) * W + lane * 8 + it * 512;
            *(float4*)&v[r][it][0] = *(const float4*)xr;
            *(float4*)&v[r][it][4] = *(const float4*)(xr + 4);
        }
    auto load_gb = [&]() {
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            *(float4*)&g[it][0] = *(const float4*)(gamma + it * 512 + lane * 8);
            *(float4*)&g[it][4] = *(const float4*)(gamma + it * 512 + lane * 8 + 4);
            *(float4*)&b[it][0] = *(const float4*)(beta + it * 512 + lane * 8);
            *(float4*)&b[it][4] = *(const float4*)(beta + it * 512 + lane * 8 + 4);
        }
    };
    if (KIND == 2) load_gb();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = row0 + r;
        if (row >= M) break;
        bf16_t* yr = y + (long)row * W + lane * 8;
        if (KIND == 1) {
#pragma unroll
            for (int it = 0; it < NV; ++it) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[r][it][e];
                *(bf16x8*)(yr + it * 512) = o;
            }
            continue;
        }
        float s = 0.0f;
#pragma unroll
        for (int it = 0; it < NV; ++it)
            s += ((v[r][it][0] + v[r][it][1]) + (v[r][it][2] + v[r][it][3])) + ((v[r][it][4] + v[r][it][5]) + (v[r][it][6] + v[r][it][7]));
        const float mu = wave_sum(s) / (float)W;
        float q = 0.0f;
#pragma unroll
        for (int it = 0; it < NV; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[r][it][e] - mu; q = fmaf(d, d, q); }
        const float rs = rsqrtf(wave_sum(q) / (float)W + 1e-5f);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        if (KIND != 2 && r == 0) load_gb();
#pragma unroll
        for (int it = 0; it < NV; ++it) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((v[r][it][e] - mu) * rs * g[it][e] + b[it][e]);
            *(bf16x8*)(yr + it * 512) = o;
        }
    }
}

int main(int argc, char** argv) {
    const int M = 32896, W = 1024, SETS = 24, reps = argc > 1 ? atoi(argv[1]) : 20;
    std::vector<float*> x(SETS), mean(SETS), rstd(SETS);
    std::vector<bf16_t*> y(SETS);
    float *gamma, *beta;
    CHECK(hipMalloc(&gamma, W * 4)); CHECK(hipMalloc(&beta, W * 4));
    std::vector<float> h((size_t)M * W);
    unsigned st = 1u;
    for (auto& f : h) { st = st * 1664525u + 1013904223u; f = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    CHECK(hipMemcpy(gamma, h.data(), W * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(beta, h.data() + W, W * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < SETS; ++i) {
        CHECK(hipMalloc(&x[i], (size_t)M * W * 4)); CHECK(hipMalloc(&y[i], (size_t)M * W * 2));
        CHECK(hipMalloc(&mean[i], M * 4)); CHECK(hipMalloc(&rstd[i], M * 4));
        CHECK(hipMemcpy(x[i], h.data(), (size_t)M * W * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[5] = {"shipped code (stats + affine)", "converting copy, no reductions", "gamma / beta requested up front", "two rows per wave",
                            "float4 copy, same bytes (1:1)"};
    auto launch = [&](int kind, int i) {
        const dim3 b(256);
        switch (kind) {
            case 0: hipLaunchKernelGGL(ln_fwd<0>, dim3((M + 3) / 4), b, 0, 0, x[i], gamma, beta, y[i], mean[i], rstd[i], M); break;
            case 1: hipLaunchKernelGGL(ln_fwd<1>, dim3((M + 3) / 4), b, 0, 0, x[i], gamma, beta, y[i], mean[i], rstd[i], M); break;
            case 2: hipLaunchKernelGGL(ln_fwd<2>, dim3((M + 3) / 4), b, 0, 0, x[i], gamma, beta, y[i], mean[i], rstd[i], M); break;
            case 3: hipLaunchKernelGGL(ln_fwd<3>, dim3((M + 7) / 8), b, 0, 0, x[i], gamma, beta, y[i], mean[i], rstd[i], M); break;
            default: hipLaunchKernelGGL(ln_fwd<4>, dim3((M + 3) / 4), b, 0, 0, x[i], gamma, beta, y[i], mean[i], rstd[i], M); break;
        }
    };
    for (int round = 0; round < 3; ++round)
        for (int kind = 0; kind < 5; ++kind) {
            for (int i = 0; i < SETS; ++i) launch(kind, i);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r)
                for (int i = 0; i < SETS; ++i) launch(kind, i);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / (reps * SETS), bytes = (double)M * W * 6;
            printf("round %d kind %d %-34s: %6.1f us per launch = %5.2f TB/s of 6 B per element (%.3f of 8)\n", round, kind, names[kind], us,
                   bytes / us / 1e6, bytes / us / 1e6 / 8.0);
        }
    return 0;
}
