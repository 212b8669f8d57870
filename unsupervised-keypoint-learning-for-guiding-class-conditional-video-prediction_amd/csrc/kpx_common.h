// Shared helpers for the kpx HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/kpx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KPX_WAVE 64

static inline int kpx_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

static inline hipStream_t kpx_stream(void* s) { return (hipStream_t)s; }

// Bijective XCD-aware remap of a 1-D block id: consecutive logical ids land on the same XCD
// (blocks b and b+8 share an XCD under the round-robin dispatch; speed only, never correctness).
__device__ __forceinline__ int kpx_xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float kpx_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double kpx_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float kpx_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// d(act)/d(pre-activation) from the ACTIVATED value y (relu / leaky relu keep the sign of their argument)
__device__ __forceinline__ float kpx_act_grad_from_y(float y, int act) {
    return act == KPX_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : act == KPX_ACT_LRELU ? (y > 0.f ? 1.f : 0.01f) : 1.f;
}

// Large-LDS kernels need hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device; `mask` is a per-kernel-family bitmask.
// Thread-safe: the bit is claimed with an atomic fetch_or (two racing first callers both set the attribute, which is idempotent).
static inline bool kpx_first_use_on_device(std::atomic<unsigned long long>* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    if ((mask->load(std::memory_order_acquire) >> dev) & 1ULL) return false;
    mask->fetch_or(1ULL << dev, std::memory_order_acq_rel);
    return true;
}
