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

}  // namespace forge
