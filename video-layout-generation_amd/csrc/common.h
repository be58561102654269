// Shared device helpers for the gfx950 kernels (64-lane wavefronts throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vlg_hip.h"

#define VLG_WAVE 64

// launch-status helper: kernels never synchronise, so only configuration errors surface here
static inline int vlg_last_error() { return (int)hipGetLastError(); }

static inline bool vlg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// reduce over aligned groups of W consecutive lanes (W power of two <= 64)
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// exact (erf) GELU and its derivative, matching F.gelu's default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// block-wide sum of one float per thread; result valid in thread 0 (smem >= blockDim/64 floats)
__device__ __forceinline__ float block_sum(float v, float* smem) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    if (l == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += smem[i];
    __syncthreads();
    return r;
}
