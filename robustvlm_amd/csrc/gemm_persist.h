// Shared pieces of the persistent 256x256 bf16 GEMM kernels (gemm_bf16_256p.hip: 8 waves, 128x64 wave tiles;
// gemm_bf16_256q.hip: 4 waves, 128x128 wave tiles): ring geometry, activation helpers, inline-asm LDS staging
// accessors, buffer-descriptor helpers.
#pragma once
#include "kernels.h"

namespace rvlm {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int P_M = 256, P_N = 256, P_K = 64;
constexpr int P_OPER_BYTES = P_M * P_K * 2;      // 32 KiB per operand per stage
constexpr int P_STAGE_BYTES = 2 * P_OPER_BYTES;  // 64 KiB
constexpr int P_EPI_WAVE = 4096;                 // epilogue staging bytes per wave
constexpr int PA_SLOT = P_OPER_BYTES, PB_SLOT = P_OPER_BYTES;   // ring slots: one operand half of one stage
constexpr int PB_BASE = 3 * PA_SLOT;             // A ring (3 slots) | B ring (2 slots) = 160 KiB

__device__ __forceinline__ void glds16p(const void* gptr, void* lds_ptr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                     (__attribute__((address_space(3))) void*)lds_ptr, 16, 0, 0);
}

// act(h) and act'(h) together (shared sigmoid / erf); fast reciprocal for the QuickGELU of the CLIP towers
template <int ACT>
__device__ __forceinline__ void actp_pair(float h, float& a, float& d) {
    if (ACT == RVLM_ACT_QUICK_GELU) {
        // s = sigmoid(1.702 h); act = h s; act' = s + 1.702 h s (1 - s).  5 full-rate + 2 quarter-rate instructions
        // (the epilogue of the fc1 GEMM is VALU-bound on this).
        const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(h * (-1.702f * 1.4426950408889634f)));
        a = h * s;
        d = __builtin_fmaf(1.702f, __builtin_fmaf(-a, s, a), s);
    } else {
        const float cdf = 0.5f * (1.0f + erff(h * 0.70710678118654752f));
        a = h * cdf;
        d = cdf + h * (0.3989422804014327f * __expf(-0.5f * h * h));
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Epilogue staging goes through inline-asm DS instructions: hipcc cannot prove that its own ds_write does not
// alias the LDS-DMA destinations and would drain the operand stream (vmcnt(0)) in front of the first staging write.
// DS operations of one wave execute in order, so a read needs no wait after the write it depends on; the values
// read are waited for with lds_wait() before their first use.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_w64(unsigned addr, u32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w128(unsigned addr, u32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int OFF>
__device__ __forceinline__ u32x4 lds_r128(unsigned addr) {
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Swizzle key of the activation-pair staging ([32 rows][act' 64 B | act 64 B], 8-B chunks): a permutation of the row's low
// four bits - row bits (1, 0, 3, 2) -> key bits (0, 1, 2, 3).  Any bijection keeps the 16-lane ds_write_b64 groups (16
// consecutive rows, one chunk) conflict-free; the read-back (ds_read_b64, 32 lanes = 8 rows x four 16-B slots of one output)
// needs the four same-parity rows of a half wave to differ in key bits 0 and 3, the bits the slot index 2 q does not
// touch (with key = row & 15 the read-back was a 4-way conflict: 18 % of the kernel's LDS-active cycles, PMC).
__device__ __forceinline__ int pair_key(int row) {
    return ((row >> 1) & 1) | ((row & 1) << 1) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3);
}

// 16-byte buffer store with the whole offset in the VGPR operand and soffset = 0.  With an SGPR soffset hipcc's hazard
// recogniser assumes that a >8-byte store's data registers may be overwritten by the next VALU instruction; on gfx950
// that corrupted the upper dwords of stores that were followed by dense VALU code (measured: EPI_BF16_ACT / _DACT).
// AUX = cache-policy bits of the store: 2 = nt (the output streams past the L2 and leaves it to the operand panels).  Used
// where the consumer is far away in time (act'(h): read by the backward) or the launch's operand set is large (fp32 outputs
// at K >= 2048); an output the NEXT kernel reads (qkv -> attention, out-proj -> LayerNorm) is slower to re-read when it was
// stored nt (profiles/r03_ab_attn_swizzle_nt_stores.log: all-nt build qkv forward +1.3 ms, attention forward +1.0 ms per step)
#ifndef RVLM_STORE_AUX          // A/B builds: default cache policy of the epilogue stores (16 = sc1 write-through, 17 = sc0 sc1)
#define RVLM_STORE_AUX 0
#endif
template <int AUX = RVLM_STORE_AUX>
__device__ __forceinline__ void store16(u32x4 v, __amdgpu_buffer_rsrc_t rs, int lane_off, int scalar_off) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane_off + scalar_off, 0, AUX);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, bytes, 0x00020000);
}

}  // namespace rvlm
