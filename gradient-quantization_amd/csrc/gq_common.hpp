// Shared host/device helpers for libgq_hsq.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gq_hsq.h"

#define GQ_API extern "C" __attribute__((visibility("default")))

namespace gq {

// Thread-local text of the last failure; returned by gq_last_error().
char *last_error_buf();
int fail(int code, const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Number of compute units of the current device (cached).
int cu_count();

#define GQ_CHECK_LAUNCH(what)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) return gq::fail(GQ_ERR_HIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

// ---- device helpers -------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based uniform [0,1) generator for GQ_RANDOM_DEVICE: a 64-bit mix of
// (seed, index) (splitmix64 finaliser), top 24 bits -> k * 2^-24, the same grid of
// values torch.rand produces for float32.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(uint32_t)(z >> 40) * 5.9604644775390625e-08f;  // 2^-24
}

}  // namespace gq
