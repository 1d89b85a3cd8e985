// common.h — shared host/device helpers for libforge_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#include "../../include/forge_hip.h"

namespace forge {

void set_error(const char* fmt, ...);

#define FORGE_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            ::forge::set_error(__VA_ARGS__);      \
            return (code);                        \
        }                                         \
    } while (0)

#define FORGE_LAUNCH_CHECK(what)                                                        \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ::forge::set_error("%s: launch failed: %s", what, hipGetErrorString(e_));   \
            return (int)e_;                                                             \
        }                                                                               \
    } while (0)

// Raise a kernel's dynamic-LDS limit once PER DEVICE (the attribute is per device; a process may drive several GPUs). hipGetDevice
// and the attribute call touch no stream, so this is safe under stream capture.
#define FORGE_SET_MAX_LDS_ONCE(kernel_ptr, bytes)                                                                      \
    do {                                                                                                               \
        static bool done_[64] = {};                                                                                    \
        int dev_ = 0;                                                                                                  \
        (void)hipGetDevice(&dev_);                                                                                     \
        if (dev_ >= 0 && dev_ < 64 && !done_[dev_]) {                                                                  \
            (void)hipFuncSetAttribute((const void*)(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            done_[dev_] = true;                                                                                        \
        }                                                                                                              \
    } while (0)

constexpr int NUM_XCD = 8;   // MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8

// Bijective remap of a linear workgroup id so that each XCD (private 4 MiB L2) gets one
// CONTIGUOUS chunk of the logical grid instead of every 8th workgroup. Speed only — results
// never depend on placement.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg / NUM_XCD, r = nwg % NUM_XCD;
    const unsigned xcd = bid % NUM_XCD, k = bid / NUM_XCD;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float4 f4_fma(float w, float4 v, float4 a) {
    a.x = fmaf(w, v.x, a.x);
    a.y = fmaf(w, v.y, a.y);
    a.z = fmaf(w, v.z, a.z);
    a.w = fmaf(w, v.w, a.w);
    return a;
}

// hardware fp32 atomic add (global_atomic_add_f32), no CAS loop
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- LDS-DMA (gfx950): global memory -> LDS without staging registers -------------------------------------------------------------
// lds_dma16(rsrc, voff, lds): `buffer_load_dwordx4 v, s[rsrc], 0 offen lds` with M0 = lds. Every lane of the wave loads the 16 bytes at
// buffer byte offset voff[lane] and the hardware writes them to LDS byte address lds + 16 lane (lds must be wave-uniform); lanes whose
// offset lies outside the buffer's num_records write zeros (the halo / ragged-edge padding of every GEMM here). It is inline assembly
// on purpose: the compiler's waitcnt insertion does not see these loads, so the K loops place their own `s_waitcnt vmcnt(0)` directly
// in front of the barrier that publishes the stage, AFTER the step's MFMAs (the builtin form is drained before them). M0 is declared
// clobbered: the compiler must not keep a live M0 value (LDS / movrel / readlane uses) across the statement.
typedef int forge_v4i32 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ forge_v4i32 make_rsrc_words(const void* base, long long bytes) {     // raw dword buffer, stride 0
    const unsigned long long p = (unsigned long long)base;
    forge_v4i32 r;
    // readfirstlane: the words must live in SGPRs (an "s" operand of the inline assembly) even where the compiler cannot prove the base uniform
    r.x = __builtin_amdgcn_readfirstlane((int)(p & 0xffffffffull)); r.y = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffull));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes); r.w = 0x00020000;
    return r;
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"       // "clobber list contains reserved registers: m0" - the clobber is the point (see above)
__device__ __forceinline__ void lds_dma16(const forge_v4i32& rsrc, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc) : "memory", "m0");
}
#pragma clang diagnostic pop

__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned lds_addr(const void* smem_ptr) {                             // LDS byte address of a __shared__ pointer
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)smem_ptr;
}

}  // namespace forge
