// Shared helpers for the idiff HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/idiff.h"

extern thread_local char g_idiff_err[512];
extern std::atomic<unsigned long long> g_idiff_launches;  // kernel launches enqueued by this library, any host thread (idiff_launch_count)

#define IDIFF_FAIL(code, ...)                                   \
    do {                                                        \
        snprintf(g_idiff_err, sizeof(g_idiff_err), __VA_ARGS__); \
        return (code);                                          \
    } while (0)

#define IDIFF_CHECK_ARG(cond, ...) \
    do {                           \
        if (!(cond)) IDIFF_FAIL(IDIFF_E_BADARG, __VA_ARGS__); \
    } while (0)

#define IDIFF_CHECK_LAUNCH(name)                                                               \
    do {                                                                                       \
        g_idiff_launches.fetch_add(1, std::memory_order_relaxed);                              \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess) IDIFF_FAIL(IDIFF_E_HIP, "%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: cache what has been set per (call site, device), safe for
// concurrent host threads (one cache object per kernel instantiation: a function-local static of the launcher).
struct idiff_dyn_lds_cache {
    std::atomic<size_t> set[64];
};
inline hipError_t idiff_ensure_dyn_lds(idiff_dyn_lds_cache& c, const void* fn, size_t bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool cached = dev >= 0 && dev < 64;
    if (cached && c.set[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (cached) {
        size_t prev = c.set[dev].load(std::memory_order_relaxed);
        while (prev < bytes && !c.set[dev].compare_exchange_weak(prev, bytes, std::memory_order_release)) {}
    }
    return hipSuccess;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// v_exp_f32 + v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE divide: used on the conv gather/epilogue
__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// XCD-aware bijective remap of a linear workgroup id: blocks b and b+8 share an XCD (observed
// round-robin dispatch), so give each XCD a contiguous chunk of the logical grid.  Speed only.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, slot = bid >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

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
// sum over each 32-lane half independently
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
