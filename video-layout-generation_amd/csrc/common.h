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
// bf16 storage (activations of the bf16 mode): 4 consecutive elements = 8 bytes.  Widening is a 16-bit shift,
// narrowing is round-to-nearest-even (v_cvt_pk_bf16_f32).  ld4 / st4 are overloaded on the element type so the
// row kernels are written once.
typedef __bf16 bf16_t;
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                       __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    const bf16x4_t t = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<bf16x4_t*>(p) = t;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }
static inline bool vlg_aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// GELU (erf form, F.gelu's default) and its derivative.  erf comes from Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, i.e. below fp32 resolution of the result) instead of libm's erff: one
// v_exp_f32 + one v_rcp_f32 + 6 FMAs, and the SAME exponential exp(-x^2/2) serves the normal pdf
// needed by the derivative.  These run in GEMM epilogues, so their VALU cost is on the critical path
// of the store tail.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x);
    const float e = __expf(-0.5f * x * x);                     // exp(-(x/sqrt2)^2)
    // v_rcp_f32 (1 ulp) - NOT __frcp_rn, which hipcc expands to the 10-instruction correctly rounded division
    // (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas, v_div_fixup): a third of this function, in a GEMM epilogue
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * 0.70710678118654752f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float tail = 0.5f * poly * e;                        // 1 - Phi(|x|)
    cdf = x >= 0.f ? 1.0f - tail : tail;
    pdf = 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float c, p;
    gelu_parts(x, c, p);
    return x * c;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float c, p;
    gelu_parts(x, c, p);
    return c + x * p;
}

// Two elements at a time, on packed fp32 instructions (v_pk_mul / v_pk_fma / v_pk_add_f32: one issue slot for two lanes'
// worth of work).  For the fp32 GEMM epilogues: v_mfma_f32_32x32x2_f32 and the vector ALU of a SIMD execute SERIALLY
// (tools/micro/mfma_f32_valu_share.hip: +8 cycles of matrix time per vector instruction of ANY wave on the SIMD, +12
// per transcendental, packed or not), so an epilogue's vector instruction count is paid in matrix throughput.
// Same formulas as gelu_parts above; branch-free:  gelu(x) = max(x, 0) - |x * tail|,  Phi(x) = 0.5 + copysign(0.5 - tail, x).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2f v2(float a) { return v2f{a, a}; }
// N pairs in LOCKSTEP: every step is a burst of N independent instructions.  A wave in its epilogue shares the SIMD with
// waves that issue 64-cycle MFMAs; a lone ready vector instruction waits for the MFMA in flight, so a dependent chain
// advances one link per MFMA (a 17-link chain per pair = ~17 us per 64x64 tile), while N independent instructions issue
// back to back behind ONE wait.  The steps are fenced so the scheduler keeps them whole.
#define VLG_STEP() __builtin_amdgcn_sched_barrier(0)
template <int N>
__device__ __forceinline__ void gelu_parts2n(const v2f (&x)[N], v2f (&tail)[N], v2f (&e)[N]) {
    const float k = 0.3275911f * 0.70710678118654752f;
    v2f t[N], p[N];
#pragma unroll
    for (int n = 0; n < N; ++n) e[n] = x[n] * x[n];
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) e[n] = e[n] * v2(-0.5f * 1.4426950408889634f);
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = v2f{fmaf(fabsf(x[n].x), k, 1.0f), fmaf(fabsf(x[n].y), k, 1.0f)};
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) e[n] = v2f{__builtin_amdgcn_exp2f(e[n].x), __builtin_amdgcn_exp2f(e[n].y)};      // exp(-x^2 / 2)
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) t[n] = v2f{__builtin_amdgcn_rcpf(t[n].x), __builtin_amdgcn_rcpf(t[n].y)};
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(t[n], v2(0.5f * 1.061405429f), v2(0.5f * -1.453152027f));
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(t[n], p[n], v2(0.5f * 1.421413741f));
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(t[n], p[n], v2(0.5f * -0.284496736f));
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(t[n], p[n], v2(0.5f * 0.254829592f));
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) p[n] = p[n] * t[n];
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) tail[n] = p[n] * e[n];                                                            // 1 - Phi(|x|)
    VLG_STEP();
}
// v_max_f32 against 0 in ONE instruction (fmaxf() adds a canonicalising v_max_f32 x, x, x in front)
__device__ __forceinline__ float relu_f(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
template <int N>
__device__ __forceinline__ void gelu2n(v2f (&x)[N]) {            // in place
    v2f tail[N], e[N];
    gelu_parts2n<N>(x, tail, e);
#pragma unroll
    for (int n = 0; n < N; ++n) tail[n] = x[n] * tail[n];
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) x[n] = v2f{relu_f(x[n].x), relu_f(x[n].y)};
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) x[n] = v2f{x[n].x - fabsf(tail[n].x), x[n].y - fabsf(tail[n].y)};
    VLG_STEP();
}
// x -> gelu(x) in place, d = gelu'(x): the derivative costs five more steps on top of the shared exp / reciprocal / polynomial
template <int N>
__device__ __forceinline__ void gelu_both2n(v2f (&x)[N], v2f (&d)[N]) {
    v2f tail[N], e[N];
    gelu_parts2n<N>(x, tail, e);
#pragma unroll
    for (int n = 0; n < N; ++n) { d[n] = v2(0.5f) - tail[n]; e[n] = x[n] * e[n]; tail[n] = x[n] * tail[n]; }
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = v2f{__builtin_copysignf(d[n].x, x[n].x), __builtin_copysignf(d[n].y, x[n].y)};
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) { d[n] = d[n] + v2(0.5f); x[n] = v2f{relu_f(x[n].x), relu_f(x[n].y)}; }
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        d[n] = __builtin_elementwise_fma(e[n], v2(0.39894228040143268f), d[n]);
        x[n] = v2f{x[n].x - fabsf(tail[n].x), x[n].y - fabsf(tail[n].y)};
    }
    VLG_STEP();
}
template <int N>
__device__ __forceinline__ void dgelu2n(const v2f (&x)[N], v2f (&d)[N]) {
    v2f tail[N], e[N];
    gelu_parts2n<N>(x, tail, e);
#pragma unroll
    for (int n = 0; n < N; ++n) tail[n] = v2(0.5f) - tail[n];                                                     // >= 0
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) tail[n] = v2f{__builtin_copysignf(tail[n].x, x[n].x), __builtin_copysignf(tail[n].y, x[n].y)};
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) { tail[n] = tail[n] + v2(0.5f); e[n] = x[n] * e[n]; }
    VLG_STEP();
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = __builtin_elementwise_fma(e[n], v2(0.39894228040143268f), tail[n]);
    VLG_STEP();
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
