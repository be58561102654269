// Slab reduction and fused Adam on the flat parameter buffer (both HBM-bound, float4 streams).
//
// vlg_reduce_slabs : dst[e] = sum_s slabs[s*stride + e].  Every partial-sum producer in this
//                    library (weight gradients split over tokens, layer-norm gain/bias sums,
//                    embedding-table sums) writes slabs and leaves the final sum to this kernel,
//                    so gradients are bitwise reproducible run to run (no float atomics).
//                    Algorithmic bytes: 4*len*(n_slabs + 1).
// vlg_adam_step    : torch.optim.Adam arithmetic on ONE flat fp32 buffer holding every
//                    parameter (reference src/trainer.py:83,258 - Adam(lr, betas=(beta1,0.999)),
//                    no weight decay, no amsgrad).  One launch per step instead of one per tensor.
//                    Algorithmic bytes: 16 B read + 12 B written per parameter.
#include "common.h"
#include "reduce_body.h"
#include <math.h>

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slabs, int64_t stride,
                                                           int n_slabs, float* __restrict__ dst, int64_t len4) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 acc = f4_zero();
        const float* p = slabs + e * 4;
        int s = 0;
        for (; s + 4 <= n_slabs; s += 4) {       // 4 independent loads in flight per lane
            const float4 a = ld4(p), b = ld4(p + stride), c = ld4(p + 2 * stride), d = ld4(p + 3 * stride);
            acc = f4_add(acc, f4_add(f4_add(a, b), f4_add(c, d)));
            p += 4 * stride;
        }
        for (; s < n_slabs; ++s) { acc = f4_add(acc, ld4(p)); p += stride; }
        st4(dst + e * 4, acc);
    }
}

// Many slabs, short vectors (layer-norm gain/bias: 512 slabs x 2d floats; embedding tables): the flat
// kernel above would leave one block walking hundreds of slabs serially.  Here a block owns 16 float4
// columns and its 256 threads split the slabs 16 ways, then combine through LDS in a fixed order
// (still bitwise reproducible).
__global__ __launch_bounds__(256) void reduce_slabs_tall_kernel(const float* __restrict__ slabs, int64_t stride,
                                                                int n_slabs, float* __restrict__ dst, int64_t len4) {
    __shared__ float4 part[16][16];
    const int c = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int64_t col = (int64_t)blockIdx.x * 16 + c;
    float4 acc = f4_zero();
    if (col < len4) {
        const float* p = slabs + col * 4;
        int s = sg;
        for (; s + 48 < n_slabs; s += 64) {       // 4 independent loads in flight per lane
            const float4 a = ld4(p + s * stride), b = ld4(p + (s + 16) * stride);
            const float4 c2 = ld4(p + (s + 32) * stride), d = ld4(p + (s + 48) * stride);
            acc = f4_add(acc, f4_add(f4_add(a, b), f4_add(c2, d)));
        }
        for (; s < n_slabs; s += 16) acc = f4_add(acc, ld4(p + s * stride));
    }
    part[sg][c] = acc;
    __syncthreads();
    if (sg == 0 && col < len4) {
        float4 t = part[0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) t = f4_add(t, part[k][c]);
        st4(dst + col * 4, t);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   float beta1, float beta2, float eps, float step_size,
                                                   float sqrt_bc2, float grad_scale, bf16_t* __restrict__ shadow,
                                                   const float* __restrict__ coef_dev) {
    // captured-graph mode: the bias-correction factors of THIS step live in device memory (adam_state_kernel), because a
    // replayed launch cannot take new by-value arguments
    if (coef_dev != nullptr) { step_size = coef_dev[0]; sqrt_bc2 = coef_dev[1]; }
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
    (void)beta1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 p = ld4(param + e * 4), g = ld4(grad + e * 4), mm = ld4(m + e * 4), vv = ld4(v + e * 4);
        float* pp = &p.x; float* gp = &g.x; float* mp = &mm.x; float* vp = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gp[k] * grad_scale;
            mp[k] = mp[k] + omb1 * (gk - mp[k]);            // exp_avg.lerp_(grad, 1-beta1)
            vp[k] = vp[k] * beta2 + omb2 * gk * gk;
            const float denom = sqrtf(vp[k]) / sqrt_bc2 + eps;
            pp[k] -= step_size * (mp[k] / denom);
        }
        st4(param + e * 4, p); st4(m + e * 4, mm); st4(v + e * 4, vv);
        if (shadow != nullptr) st4(shadow + e * 4, p);         // bf16 copy of the updated weights for the bf16-MFMA projections
    }
}

static unsigned stream_blocks(int64_t n4) {
    int64_t b = (n4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// Many independent reductions in ONE launch (blockIdx.y = table row): the backward of the reference's GridNet leaves one
// slab set per convolution (61) - each a few microseconds of work behind its own launch.  table[i] = {slabs pointer,
// slab stride, slab count, destination pointer, length}, int64 each, in DEVICE memory (built once: the tape is static).
// Every row is reduced exactly as vlg_reduce_slabs would reduce it (same summation order).
__global__ __launch_bounds__(256) void reduce_slabs_table_kernel(const int64_t* __restrict__ table) {
    __shared__ float4 part[16][16];
    reduce_table_row(table, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, part);          // csrc/reduce_body.h
}

extern "C" int vlg_reduce_slabs_table(const int64_t* table, int n_rows, int blocks_per_row, void* stream) {
    if (!table || n_rows < 1 || n_rows > 65535 || blocks_per_row < 1 || blocks_per_row > 4096) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(reduce_slabs_table_kernel, dim3((unsigned)blocks_per_row, (unsigned)n_rows), dim3(256), 0,
                       (hipStream_t)stream, table);
    return vlg_last_error();
}

// table[i] = {partials pointer, count, destination pointer}: dst[0] = sum of the partials (as vlg_sum_partials, accumulate = 0)
__global__ __launch_bounds__(256) void sum_partials_table_kernel(const int64_t* __restrict__ table) {
    __shared__ float red[4];
    const int64_t* t = table + 3 * (int64_t)blockIdx.x;
    const float* __restrict__ part = reinterpret_cast<const float*>(t[0]);
    const int n = (int)t[1];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) reinterpret_cast<float*>(t[2])[0] = s;
}

extern "C" int vlg_sum_partials_table(const int64_t* table, int n_rows, void* stream) {
    if (!table || n_rows < 1) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL(sum_partials_table_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, table);
    return vlg_last_error();
}

extern "C" int vlg_reduce_slabs(const float* slabs, int64_t slab_stride, int n_slabs, float* dst, int64_t len,
                                void* stream) {
    if (n_slabs < 1 || len < 4 || (len & 3) || (slab_stride & 3) || slab_stride < len) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(slabs) || !vlg_aligned16(dst)) return VLG_ERR_ALIGN;
    const int64_t len4 = len / 4;
    if (len4 < 32768 && n_slabs >= 16)         // fewer than 128 flat blocks: split the slabs across threads instead
        hipLaunchKernelGGL(reduce_slabs_tall_kernel, dim3((unsigned)((len4 + 15) / 16)), dim3(256), 0,
                           (hipStream_t)stream, slabs, slab_stride, n_slabs, dst, len4);
    else
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(stream_blocks(len4)), dim3(256), 0, (hipStream_t)stream, slabs,
                           slab_stride, n_slabs, dst, len4);
    return vlg_last_error();
}

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, bf16_t* shadow, int64_t n,
                       int step, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (n < 4 || (n & 3) || step < 1) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(param) || !vlg_aligned16(grad) || !vlg_aligned16(exp_avg) || !vlg_aligned16(exp_avg_sq) ||
        (shadow && !vlg_aligned8(shadow))) return VLG_ERR_ALIGN;
    // bias corrections in double, as torch.optim.Adam computes them on the host
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float sqrt_bc2 = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n / 4, beta1, beta2, eps, step_size, sqrt_bc2, grad_scale, shadow, (const float*)nullptr);
    return vlg_last_error();
}

extern "C" int vlg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             int step, float lr, float beta1, float beta2, float eps, float grad_scale,
                             void* stream) {
    return adam_launch(param, grad, exp_avg, exp_avg_sq, nullptr, n, step, lr, beta1, beta2, eps, grad_scale, stream);
}

extern "C" int vlg_adam_step_bf16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                  vlg_bf16* shadow, int64_t n, int step, float lr, float beta1, float beta2,
                                  float eps, float grad_scale, void* stream) {
    if (!shadow) return VLG_ERR_SHAPE;
    return adam_launch(param, grad, exp_avg, exp_avg_sq, reinterpret_cast<bf16_t*>(shadow), n, step, lr, beta1, beta2, eps,
                       grad_scale, stream);
}

// ---- Adam with its step counter on the device (hipGraph replay: vlg/engine.py capture_train_step)
// state = {float step_size, float sqrt_bc2, int step, int pad}: one thread advances the counter and recomputes the two
// factors in double, exactly as adam_launch does on the host.
__global__ void adam_state_kernel(float* state, float lr, float beta1, float beta2, int advance) {
    int* istate = reinterpret_cast<int*>(state);
    const int step = istate[2] + (advance ? 1 : 0);
    istate[2] = step;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    state[0] = (float)((double)lr / bc1);
    state[1] = (float)sqrt(bc2);
}

extern "C" int vlg_adam_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, vlg_bf16* shadow,
                                   int64_t n, float* state, int advance, float lr, float beta1, float beta2, float eps,
                                   float grad_scale, void* stream) {
    if (n < 4 || (n & 3) || !state) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(param) || !vlg_aligned16(grad) || !vlg_aligned16(exp_avg) || !vlg_aligned16(exp_avg_sq) ||
        !vlg_aligned16(state) || (shadow && !vlg_aligned8(shadow))) return VLG_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_state_kernel, dim3(1), dim3(1), 0, s, state, lr, beta1, beta2, advance);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n / 4,
                       beta1, beta2, eps, 0.f, 1.f, grad_scale, reinterpret_cast<bf16_t*>(shadow), (const float*)state);
    return vlg_last_error();
}
