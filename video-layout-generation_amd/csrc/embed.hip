// Object-slot embedding, forward and backward (HBM-bound: one pass over (M,d) fp32).
//
// fwd  : one float4 per thread, grid-stride; writes rows in the internal order
//        m = (b*N + n)*T + t so that each (clip, slot) owns one contiguous T x d tile.
//        Algorithmic bytes: 4*M*d written + 28*M read (ids + boxes); tables are L2-resident.
// bwd  : thread c owns channel c; a block walks whole (clip, slot) sequences and keeps
//        class-table sums in LDS ([vocab][d], conflict-free: a thread only touches its own
//        column), frame-table / box sums in registers; each block writes one partial slab
//        [cls_emb | box_w | box_b | time_emb] which vlg_reduce_slabs sums (bitwise
//        reproducible - no float atomics).  Algorithmic bytes: 4*M*d read.
#include "common.h"

#define EMBED_BWD_SLABS 256

__global__ __launch_bounds__(256) void embed_fwd_kernel(
    const int64_t* __restrict__ slot_class, const float* __restrict__ slot_box,
    const float* __restrict__ cls_emb, const float* __restrict__ box_w,
    const float* __restrict__ box_b, const float* __restrict__ time_emb,
    float* __restrict__ x, int B, int T, int N, int d, int vocab) {
    const int d4 = d >> 2;
    const int64_t total = (int64_t)B * T * N * d4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / d4;
        const int c = (int)(e - m * d4) << 2;
        const int t = (int)(m % T);
        const int64_t bn = m / T;
        const int n = (int)(bn % N);
        const int64_t b = bn / N;
        const int64_t src = (b * T + t) * N + n;
        int64_t cls = slot_class[src];
        cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);   // host validates; never fault
        const float4 bx = ld4(slot_box + src * 4);
        float4 acc = f4_add(ld4(cls_emb + cls * d + c), f4_add(ld4(time_emb + (int64_t)t * d + c), ld4(box_b + c)));
        const float4 w0 = ld4(box_w + (int64_t)(c + 0) * 4), w1 = ld4(box_w + (int64_t)(c + 1) * 4);
        const float4 w2 = ld4(box_w + (int64_t)(c + 2) * 4), w3 = ld4(box_w + (int64_t)(c + 3) * 4);
        acc.x += bx.x * w0.x + bx.y * w0.y + bx.z * w0.z + bx.w * w0.w;
        acc.y += bx.x * w1.x + bx.y * w1.y + bx.z * w1.z + bx.w * w1.w;
        acc.z += bx.x * w2.x + bx.y * w2.y + bx.z * w2.z + bx.w * w2.w;
        acc.w += bx.x * w3.x + bx.y * w3.y + bx.z * w3.z + bx.w * w3.w;
        st4(x + m * d + c, acc);
    }
}

// TT = frames per clip (compile time so the frame-table sums stay in registers).
// blockDim = NG * d: thread (grp, c) owns channel c; the NG groups of a block walk different
// (clip, slot) sequences so NG times more loads are in flight per CU, and are combined through LDS
// in a fixed order at the end (reproducible).
template <int TT>
__global__ void embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ slot_class,
                                 const float* __restrict__ slot_box, float* __restrict__ slabs,
                                 int64_t slab_stride, int B, int N, int d, int vocab, int rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [NG][rows][d], rows = max(vocab, TT+5)
    const int c = threadIdx.x % d, grp = threadIdx.x / d, NG = blockDim.x / d;
    float* cls_acc = lds + (int64_t)grp * rows * d;
    for (int v = 0; v < vocab; ++v) cls_acc[v * d + c] = 0.f;
    float t_acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) t_acc[t] = 0.f;
    float b_acc = 0.f, w_acc0 = 0.f, w_acc1 = 0.f, w_acc2 = 0.f, w_acc3 = 0.f;

    const int64_t n_seq = (int64_t)B * N;
    for (int64_t seq = (int64_t)blockIdx.x * NG + grp; seq < n_seq; seq += (int64_t)gridDim.x * NG) {
        const int64_t b = seq / N;
        const int n = (int)(seq - b * N);
        float gv[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) gv[t] = dx[(seq * TT + t) * d + c];     // TT independent loads in flight
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const float g = gv[t];
            const int64_t src = (b * TT + t) * N + n;
            int64_t cls = slot_class[src];
            cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);
            const float4 bx = ld4(slot_box + src * 4);
            cls_acc[cls * d + c] += g;
            t_acc[t] += g;
            b_acc += g;
            w_acc0 += g * bx.x; w_acc1 += g * bx.y; w_acc2 += g * bx.z; w_acc3 += g * bx.w;
        }
    }
    float* slab = slabs + (int64_t)blockIdx.x * slab_stride;
    __syncthreads();
    if (grp == 0) {
        for (int v = 0; v < vocab; ++v) {
            float s = 0.f;
            for (int k = 0; k < NG; ++k) s += lds[((int64_t)k * rows + v) * d + c];
            slab[v * d + c] = s;
        }
    }
    __syncthreads();
    // second round through the same LDS: the register accumulators (TT frame rows, bias, 4 box columns)
    float* mine = lds + (int64_t)grp * rows * d;
#pragma unroll
    for (int t = 0; t < TT; ++t) mine[t * d + c] = t_acc[t];
    mine[(TT + 0) * d + c] = b_acc;
    mine[(TT + 1) * d + c] = w_acc0; mine[(TT + 2) * d + c] = w_acc1;
    mine[(TT + 3) * d + c] = w_acc2; mine[(TT + 4) * d + c] = w_acc3;
    __syncthreads();
    if (grp == 0) {
        float r[TT + 5];
#pragma unroll
        for (int j = 0; j < TT + 5; ++j) {
            float s = 0.f;
            for (int k = 0; k < NG; ++k) s += lds[((int64_t)k * rows + j) * d + c];
            r[j] = s;
        }
        int64_t off = (int64_t)vocab * d;
        st4(slab + off + (int64_t)c * 4, make_float4(r[TT + 1], r[TT + 2], r[TT + 3], r[TT + 4]));
        off += (int64_t)d * 4;
        slab[off + c] = r[TT];
        off += d;
#pragma unroll
        for (int t = 0; t < TT; ++t) slab[off + (int64_t)t * d + c] = r[t];
    }
}

extern "C" int vlg_embed_fwd(const int64_t* slot_class, const float* slot_box, const float* cls_emb,
                             const float* box_w, const float* box_b, const float* time_emb, float* x,
                             int B, int T, int N, int d, int vocab, void* stream) {
    if (B < 1 || T < 1 || N < 1 || d < 4 || (d & 3) || vocab < 1) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(slot_box) || !vlg_aligned16(cls_emb) || !vlg_aligned16(box_w) ||
        !vlg_aligned16(box_b) || !vlg_aligned16(time_emb) || !vlg_aligned16(x)) return VLG_ERR_ALIGN;
    const int64_t total = (int64_t)B * T * N * (d / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       slot_class, slot_box, cls_emb, box_w, box_b, time_emb, x, B, T, N, d, vocab);
    return vlg_last_error();
}

extern "C" int vlg_embed_bwd_slabs(void) { return EMBED_BWD_SLABS; }

extern "C" int vlg_embed_bwd(const float* dx, const int64_t* slot_class, const float* slot_box,
                             float* slabs, int64_t slab_stride, int64_t slab_capacity, int B, int T, int N, int d,
                             int vocab, void* stream) {
    if (B < 1 || N < 1 || d < 64 || d > 1024 || (d & 63) || vocab < 1) return VLG_ERR_SHAPE;
    if (slab_capacity < (int64_t)EMBED_BWD_SLABS * slab_stride) return VLG_ERR_SHAPE;      // the caller's buffer must hold every slab
    const int64_t need = (int64_t)vocab * d + (int64_t)d * 4 + d + (int64_t)T * d;
    if (slab_stride < need || (slab_stride & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(slot_box) || !vlg_aligned16(slabs)) return VLG_ERR_ALIGN;
    const int rows = vocab > T + 5 ? vocab : T + 5;
    int ng = 1024 / d;                                    // groups per block
    while (ng > 1 && (size_t)ng * rows * d * sizeof(float) > 96 * 1024) ng >>= 1;
    const size_t lds = (size_t)ng * rows * d * sizeof(float);
    if (lds > 160 * 1024) return VLG_ERR_SHAPE;
    const dim3 grid(EMBED_BWD_SLABS), block(ng * d);
    hipStream_t s = (hipStream_t)stream;
#define EMBED_BWD_LAUNCH(TT)                                                                                    \
    {                                                                                                           \
        (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                            (int)lds);                                                                          \
        hipLaunchKernelGGL(embed_bwd_kernel<TT>, grid, block, lds, s, dx, slot_class, slot_box, slabs,          \
                           slab_stride, B, N, d, vocab, rows);                                                  \
    }
    switch (T) {
        case 4:  EMBED_BWD_LAUNCH(4) break;
        case 8:  EMBED_BWD_LAUNCH(8) break;
        case 16: EMBED_BWD_LAUNCH(16) break;
        case 32: EMBED_BWD_LAUNCH(32) break;
        default: return VLG_ERR_SHAPE;
    }
#undef EMBED_BWD_LAUNCH
    return vlg_last_error();
}
