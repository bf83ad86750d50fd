// Shared device/host helpers for librvlm (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

#include "../../include/rvlm.h"

namespace rvlm {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing (no exceptions across the ABI) --------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define RVLM_HIP(expr)                                                                       \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rvlm::fail(RVLM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define RVLM_CHECK_LAUNCH()                                                                  \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess)                                                                \
            return ::rvlm::fail(RVLM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(_e)); \
    } while (0)

#define RVLM_REQUIRE(cond, msg)                                            \
    do {                                                                   \
        if (!(cond)) return ::rvlm::fail(RVLM_ERR_ARG, std::string(msg)); \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// hipFuncSetAttribute opt-ins (dynamic LDS beyond 64 KiB) are per DEVICE: a launcher remembers them in a per-kernel atomic
// mask, bit d = done on device d - not in a process-wide bool (another device of the process would launch without the
// opt-in; two host threads would race)
#define RVLM_ONCE_PER_DEVICE(mask_var, stmt)                                                       \
    do {                                                                                            \
        int _dev = 0;                                                                               \
        (void)hipGetDevice(&_dev);                                                                  \
        const unsigned long long _bit = 1ull << (_dev & 63);                                        \
        if (!(__atomic_load_n(&(mask_var), __ATOMIC_ACQUIRE) & _bit)) {                             \
            stmt;                                                                                   \
            __atomic_fetch_or(&(mask_var), _bit, __ATOMIC_RELEASE);                                 \
        }                                                                                           \
    } while (0)
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// activation and derivative (open_clip QuickGELU x*sigmoid(1.702x), or exact-erf GELU)
__device__ __forceinline__ float act_fwd(float h, int act) {
    if (act == RVLM_ACT_QUICK_GELU) {
        return h / (1.0f + __expf(-1.702f * h));
    } else {
        return 0.5f * h * (1.0f + erff(h * 0.70710678118654752f));
    }
}
__device__ __forceinline__ float act_bwd(float h, int act) {
    if (act == RVLM_ACT_QUICK_GELU) {
        float s = 1.0f / (1.0f + __expf(-1.702f * h));
        return s * (1.0f + 1.702f * h * (1.0f - s));
    } else {
        float cdf = 0.5f * (1.0f + erff(h * 0.70710678118654752f));
        float pdf = 0.3989422804014327f * __expf(-0.5f * h * h);
        return cdf + h * pdf;
    }
}
// act(h) and act'(h) together (they share the sigmoid / erf): the bf16 fc1 epilogue stores both, so that the fc2 dgrad
// epilogue is one multiply instead of an exp + rcp per element
__device__ __forceinline__ void act_pair(float h, int act, float& a, float& d) {
    if (act == RVLM_ACT_QUICK_GELU) {
        // the SAME instruction sequence as actp_pair<QUICK_GELU> of the persistent kernel (gemm_persist.h): which GEMM
        // kernel a shape is routed to must not change the bits of act(h) / act'(h) (batch-split invariance)
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h * (-1.702f * 1.4426950408889634f)));
        a = h * s;
        d = __builtin_fmaf(1.702f, __builtin_fmaf(-a, s, a), s);
    } else {
        const float cdf = 0.5f * (1.0f + erff(h * 0.70710678118654752f));
        a = h * cdf;
        d = cdf + h * (0.3989422804014327f * __expf(-0.5f * h * h));
    }
}
// precise variants for the fp32 (parity) path
__device__ __forceinline__ float act_fwd_precise(float h, int act) {
    if (act == RVLM_ACT_QUICK_GELU) {
        return h * (1.0f / (1.0f + expf(-1.702f * h)));
    } else {
        return 0.5f * h * (1.0f + erff(h * 0.70710678118654752f));
    }
}
__device__ __forceinline__ float act_bwd_precise(float h, int act) {
    if (act == RVLM_ACT_QUICK_GELU) {
        float s = 1.0f / (1.0f + expf(-1.702f * h));
        return s * (1.0f + 1.702f * h * (1.0f - s));
    } else {
        float cdf = 0.5f * (1.0f + erff(h * 0.70710678118654752f));
        float pdf = 0.3989422804014327f * expf(-0.5f * h * h);
        return cdf + h * pdf;
    }
}

}  // namespace rvlm
