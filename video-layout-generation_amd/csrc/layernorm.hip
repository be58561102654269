// Layer-norm forward / backward, one 64-lane wavefront per token row (HBM-bound).
//
// A row of d floats lives in registers (d/64 per lane, float4 loads when d % 256 == 0),
// mean/variance/gradient sums are wavefront-shuffle reductions - no LDS in the row path.
// fwd algorithmic bytes: 8*d per row (+8 for the saved mean / rstd).
// bwd algorithmic bytes: 16*d per row (dy, x, dres in; dx out); per-channel dgamma/dbeta
// sums stay in registers across the grid-stride loop, are combined across the block's 4
// waves through LDS and leave as one [dgamma | dbeta] slab per block (reduced by
// vlg_reduce_slabs: reproducible, no atomics).
#include "common.h"

#define LN_BLOCK 256
#define LN_WAVES (LN_BLOCK / 64)
#define LN_BWD_MAX_BLOCKS 512

// E = floats per lane; V4 => lane holds E/4 float4 at columns 4*lane + 256*i, else scalars at lane + 64*i
template <int E, bool V4>
struct RowIO {
    template <typename EL>
    __device__ static __forceinline__ void load(const EL* __restrict__ row, int lane, float (&v)[E]) {
        if constexpr (V4) {
#pragma unroll
            for (int i = 0; i < E / 4; ++i) {
                const float4 t = ld4(row + 4 * lane + 256 * i);
                v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) v[i] = ld1(row + lane + 64 * i);
        }
    }
    template <typename EL>
    __device__ static __forceinline__ void store(EL* __restrict__ row, int lane, const float (&v)[E]) {
        if constexpr (V4) {
#pragma unroll
            for (int i = 0; i < E / 4; ++i)
                st4(row + 4 * lane + 256 * i, make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]));
        } else {
#pragma unroll
            for (int i = 0; i < E; ++i) st1(row + lane + 64 * i, v[i]);
        }
    }
    // column index of register j
    __device__ static __forceinline__ int col(int lane, int j) {
        if constexpr (V4) return 4 * lane + 256 * (j >> 2) + (j & 3);
        else return lane + 64 * j;
    }
};

// rows a wave holds in flight per loop trip: one 1-KB row per wave leaves too few bytes in flight to cover the HBM latency
// (8 192 resident waves x 1 KB = 8 MB against ~16 MB for 8 TB/s x 2 us), so short rows are processed LN_ROWS(E) at a time
#define LN_ROWS(E) ((E) <= 4 ? 4 : (E) <= 8 ? 2 : 1)

template <int E, bool V4, typename EY = float>
__global__ __launch_bounds__(LN_BLOCK) void ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    EY* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, float eps) {
    constexpr int d = E * 64, R = LN_ROWS(E);
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * LN_WAVES;
    float g[E], b[E];
    RowIO<E, V4>::load(gamma, lane, g);
    RowIO<E, V4>::load(beta, lane, b);
    for (int64_t r0 = wave * R; r0 < rows; r0 += nwaves * R) {
        float v[R][E];
#pragma unroll
        for (int i = 0; i < R; ++i) {                          // a short last group re-reads the last row, stores are guarded
            const int64_t r = r0 + i < rows ? r0 + i : rows - 1;
            RowIO<E, V4>::load(x + r * d, lane, v[i]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < E; ++j) s += v[i][j];
            const float mu = wave_sum(s) * (1.0f / d);
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < E; ++j) { const float t = v[i][j] - mu; q += t * t; }
            const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / d) + eps);
#pragma unroll
            for (int j = 0; j < E; ++j) v[i][j] = (v[i][j] - mu) * rs * g[j] + b[j];
            if (r0 + i < rows) {
                RowIO<E, V4>::store(y + (r0 + i) * d, lane, v[i]);
                if (lane == 0) { mean[r0 + i] = mu; rstd[r0 + i] = rs; }
            }
        }
    }
}

template <int E, bool V4, typename EY = float>
__global__ __launch_bounds__(LN_BLOCK) void ln_bwd_kernel(
    const EY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ dres,
    float* __restrict__ dx_out, float* __restrict__ slabs, int64_t slab_stride, int64_t rows) {
    constexpr int d = E * 64, R = LN_ROWS(E);
    __shared__ float red[LN_WAVES][2 * d];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * LN_WAVES + w;
    const int64_t nwaves = (int64_t)gridDim.x * LN_WAVES;
    float g[E], dg[E], db[E];
    RowIO<E, V4>::load(gamma, lane, g);
#pragma unroll
    for (int j = 0; j < E; ++j) { dg[j] = 0.f; db[j] = 0.f; }
    for (int64_t r0 = wave * R; r0 < rows; r0 += nwaves * R) {
        float gy[R][E], xv[R][E], o[R][E], mu[R], rs[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int64_t r = r0 + i < rows ? r0 + i : rows - 1;
            RowIO<E, V4>::load(dy + r * d, lane, gy[i]);
            RowIO<E, V4>::load(x + r * d, lane, xv[i]);
            if (dres != nullptr) RowIO<E, V4>::load(dres + r * d, lane, o[i]);
            else {
#pragma unroll
                for (int j = 0; j < E; ++j) o[i][j] = 0.f;
            }
            mu[i] = mean[r];
            rs[i] = rstd[r];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float live = r0 + i < rows ? 1.f : 0.f;      // rows past the end add nothing to dgamma / dbeta
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const float xh = (xv[i][j] - mu[i]) * rs[i];
                const float gl = gy[i][j] * live;
                const float a = gy[i][j] * g[j];
                dg[j] += gl * xh;
                db[j] += gl;
                s1 += a;
                s2 += a * xh;
                xv[i][j] = xh;
                gy[i][j] = a;
            }
            s1 = wave_sum(s1) * (1.0f / d);
            s2 = wave_sum(s2) * (1.0f / d);
#pragma unroll
            for (int j = 0; j < E; ++j) o[i][j] += rs[i] * (gy[i][j] - s1 - xv[i][j] * s2);
            if (r0 + i < rows) RowIO<E, V4>::store(dx_out + (r0 + i) * d, lane, o[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < E; ++j) {
        const int c = RowIO<E, V4>::col(lane, j);
        red[w][c] = dg[j];
        red[w][d + c] = db[j];
    }
    __syncthreads();
    float* slab = slabs + (int64_t)blockIdx.x * slab_stride;
    for (int c = threadIdx.x; c < 2 * d; c += LN_BLOCK) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_WAVES; ++k) s += red[k][c];
        slab[c] = s;
    }
}

static int ln_fwd_blocks(int64_t rows) {
    int64_t b = (rows + LN_WAVES - 1) / LN_WAVES;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
static int ln_bwd_blocks(int64_t rows) {
    int64_t b = (rows + LN_WAVES - 1) / LN_WAVES;
    return (int)(b > LN_BWD_MAX_BLOCKS ? LN_BWD_MAX_BLOCKS : (b < 1 ? 1 : b));
}

#define LN_DISPATCH(d, CALL)                                 \
    switch (d) {                                             \
        case 64:   { CALL(1, false) } break;                 \
        case 128:  { CALL(2, false) } break;                 \
        case 192:  { CALL(3, false) } break;                 \
        case 256:  { CALL(4, true) } break;                  \
        case 512:  { CALL(8, true) } break;                  \
        case 768:  { CALL(12, true) } break;                 \
        case 1024: { CALL(16, true) } break;                 \
        default: return VLG_ERR_SHAPE;                       \
    }

extern "C" int vlg_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                 float* mean, float* rstd, int64_t rows, int d, float eps, void* stream) {
    if (rows < 1) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(x) || !vlg_aligned16(y) || !vlg_aligned16(gamma) || !vlg_aligned16(beta)) return VLG_ERR_ALIGN;
    const dim3 grid(ln_fwd_blocks(rows)), block(LN_BLOCK);
    hipStream_t s = (hipStream_t)stream;
#define CALL(E, V4) hipLaunchKernelGGL((ln_fwd_kernel<E, V4>), grid, block, 0, s, x, gamma, beta, y, mean, rstd, rows, eps);
    LN_DISPATCH(d, CALL)
#undef CALL
    return vlg_last_error();
}

extern "C" int vlg_layernorm_bwd_slabs(int64_t rows) { return ln_bwd_blocks(rows); }

extern "C" int vlg_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* dres, float* dx_out, float* slabs,
                                 int64_t slab_stride, int64_t slab_capacity, int64_t rows, int d, void* stream) {
    if (rows < 1 || slab_stride < 2 * (int64_t)d) return VLG_ERR_SHAPE;
    if (slab_capacity < (int64_t)ln_bwd_blocks(rows) * slab_stride) return VLG_ERR_SHAPE;   // every block writes one slab
    if (!vlg_aligned16(dy) || !vlg_aligned16(x) || !vlg_aligned16(gamma) || !vlg_aligned16(dx_out) ||
        (dres && !vlg_aligned16(dres))) return VLG_ERR_ALIGN;
    const dim3 grid(ln_bwd_blocks(rows)), block(LN_BLOCK);
    hipStream_t s = (hipStream_t)stream;
#define CALL(E, V4) hipLaunchKernelGGL((ln_bwd_kernel<E, V4>), grid, block, 0, s, dy, x, mean, rstd, gamma, dres, dx_out, slabs, slab_stride, rows);
    LN_DISPATCH(d, CALL)
#undef CALL
    return vlg_last_error();
}

extern "C" int vlg_layernorm_fwd_bf16(const float* x, const float* gamma, const float* beta, vlg_bf16* y,
                                      float* mean, float* rstd, int64_t rows, int d, float eps, void* stream) {
    if (rows < 1) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(x) || !vlg_aligned8(y) || !vlg_aligned16(gamma) || !vlg_aligned16(beta)) return VLG_ERR_ALIGN;
    const dim3 grid(ln_fwd_blocks(rows)), block(LN_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    bf16_t* yb = reinterpret_cast<bf16_t*>(y);
#define CALL(E, V4) hipLaunchKernelGGL((ln_fwd_kernel<E, V4, bf16_t>), grid, block, 0, s, x, gamma, beta, yb, mean, rstd, rows, eps);
    LN_DISPATCH(d, CALL)
#undef CALL
    return vlg_last_error();
}

extern "C" int vlg_layernorm_bwd_bf16(const vlg_bf16* dy, const float* x, const float* mean, const float* rstd,
                                      const float* gamma, const float* dres, float* dx_out, float* slabs,
                                      int64_t slab_stride, int64_t slab_capacity, int64_t rows, int d, void* stream) {
    if (rows < 1 || slab_stride < 2 * (int64_t)d) return VLG_ERR_SHAPE;
    if (slab_capacity < (int64_t)ln_bwd_blocks(rows) * slab_stride) return VLG_ERR_SHAPE;   // every block writes one slab
    if (!vlg_aligned8(dy) || !vlg_aligned16(x) || !vlg_aligned16(gamma) || !vlg_aligned16(dx_out) ||
        (dres && !vlg_aligned16(dres))) return VLG_ERR_ALIGN;
    const dim3 grid(ln_bwd_blocks(rows)), block(LN_BLOCK);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* dyb = reinterpret_cast<const bf16_t*>(dy);
#define CALL(E, V4) hipLaunchKernelGGL((ln_bwd_kernel<E, V4, bf16_t>), grid, block, 0, s, dyb, x, mean, rstd, gamma, dres, dx_out, slabs, slab_stride, rows);
    LN_DISPATCH(d, CALL)
#undef CALL
    return vlg_last_error();
}
