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

// Thread (slot, q) of a block owns channels 4q .. 4q+3 of every row the block's slot walks (d/4 threads per row,
// 1024/d rows in flight per block): the box projection rows and the bias of those channels are loaded ONCE, the
// row's class id and box are one broadcast load each, and per row the thread does two coalesced 16-B table reads
// (class row, frame row) and one coalesced 16-B store.  Round 1 re-read 4 box_w float4 + ids + box per float4
// written (22 % of the HBM roofline).
__global__ __launch_bounds__(256) void embed_fwd_kernel(
    const int64_t* __restrict__ slot_class, const float* __restrict__ slot_box,
    const float* __restrict__ cls_emb, const float* __restrict__ box_w,
    const float* __restrict__ box_b, const float* __restrict__ time_emb,
    float* __restrict__ x, int B, int T, int N, int d, int vocab) {
    const int lpr = d >> 2;                                   // threads per row
    const int q = threadIdx.x % lpr, slot = threadIdx.x / lpr, rpb = blockDim.x / lpr;
    if (slot >= rpb) return;
    const int c = q << 2;
    const float4 w0 = ld4(box_w + (int64_t)(c + 0) * 4), w1 = ld4(box_w + (int64_t)(c + 1) * 4);
    const float4 w2 = ld4(box_w + (int64_t)(c + 2) * 4), w3 = ld4(box_w + (int64_t)(c + 3) * 4);
    const float4 bb = ld4(box_b + c);
    const int64_t M = (int64_t)B * T * N;
#pragma unroll 4
    for (int64_t m = (int64_t)blockIdx.x * rpb + slot; m < M; m += (int64_t)gridDim.x * rpb) {
        const int t = (int)(m % T);
        const int64_t bn = m / T;
        const int n = (int)(bn % N);
        const int64_t b = bn / N;
        const int64_t src = (b * T + t) * N + n;
        int64_t cls = slot_class[src];
        cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);   // host validates; never fault
        const float4 bx = ld4(slot_box + src * 4);
        float4 acc = f4_add(ld4(cls_emb + cls * d + c), f4_add(ld4(time_emb + (int64_t)t * d + c), bb));
        acc.x += bx.x * w0.x + bx.y * w0.y + bx.z * w0.z + bx.w * w0.w;
        acc.y += bx.x * w1.x + bx.y * w1.y + bx.z * w1.z + bx.w * w1.w;
        acc.z += bx.x * w2.x + bx.y * w2.y + bx.z * w2.z + bx.w * w2.w;
        acc.w += bx.x * w3.x + bx.y * w3.y + bx.z * w3.z + bx.w * w3.w;
        st4(x + m * d + c, acc);
    }
}

// TT = frames per clip (compile time so the frame-table sums stay in registers).
// Thread (grp, q) owns channels 4q .. 4q+3 (one 16-B load per row instead of four 4-B ones); the G = 1024/d groups of a
// block walk different (clip, slot) sequences with TT row loads in flight each, keep class-table sums in a private
// LDS table ([vocab][d] per group: 16-B read-modify-write, a thread only touches its own columns) and frame / bias /
// box sums in registers, and are combined through LDS in a fixed order at the end (reproducible, no atomics).
// 256 blocks (one per CU: the class tables take 86 KB of LDS), 16 KB of row loads in flight per wave.
__device__ __forceinline__ float4 f4_fma(float s, float4 v, float4 a) {
    return make_float4(fmaf(s, v.x, a.x), fmaf(s, v.y, a.y), fmaf(s, v.z, a.z), fmaf(s, v.w, a.w));
}
__device__ __forceinline__ float4 f4_scale(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

template <int TT>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ slot_class,
                                                       const float* __restrict__ slot_box, float* __restrict__ slabs,
                                                       int64_t slab_stride, int B, int N, int d, int vocab, int rows, int G) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [G][rows][d], rows = max(vocab, TT+5)
    const int lpr = d >> 2;
    const int q = threadIdx.x % lpr, grp = threadIdx.x / lpr, c = q << 2;
    const bool live = grp < G;
    float* cls_acc = lds + (int64_t)(live ? grp : 0) * rows * d;
    if (live)
        for (int v = 0; v < vocab; ++v) st4(cls_acc + v * d + c, f4_zero());
    float4 t_acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) t_acc[t] = f4_zero();
    float4 b_acc = f4_zero(), w_acc0 = f4_zero(), w_acc1 = f4_zero(), w_acc2 = f4_zero(), w_acc3 = f4_zero();

    const int64_t n_seq = (int64_t)B * N;
    // The class id and box of frame t are the same for every lane of the row.  Lane tl of the wave (of the group, when
    // several groups share a wave) loads frame tl's id and box together with the row loads and the t-loop takes them by
    // lane shuffle, so a sequence costs ONE memory latency, not one per frame (round 1 / the first round-2 version chained
    // sixteen dependent id loads per sequence: 48 us for a 33.5 MB read).
    const int wl = threadIdx.x & 63;
    const int tl = lpr >= 64 ? wl : q;                    // this lane's frame slot
    const int lane0 = wl - tl;                            // first lane of the id holders
    // shuffles only when a group's lanes never straddle a wave (lpr divides 64, or is a multiple of it): at d = 192 / 320 /
    // 384 / 448 a wave mixes two groups (and their trip counts), so those widths - and d = 64 with T = 32 - load directly
    const bool shuffled = lpr >= TT && (lpr < 64 ? (64 % lpr) == 0 : (lpr & 63) == 0);
    if (live) {
        for (int64_t seq = (int64_t)blockIdx.x * G + grp; seq < n_seq; seq += (int64_t)gridDim.x * G) {
            const int64_t b = seq / N;
            const int n = (int)(seq - b * N);
            float4 gv[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) gv[t] = ld4(dx + (seq * TT + t) * d + c);     // TT independent 16-B loads in flight
            int my_cls = 0;
            float4 my_box = f4_zero();
            if (shuffled && tl < TT) {
                const int64_t src = (b * TT + tl) * N + n;
                const int64_t cl = slot_class[src];
                my_cls = (int)(cl < 0 ? 0 : (cl >= vocab ? vocab - 1 : cl));
                my_box = ld4(slot_box + src * 4);
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const float4 g = gv[t];
                int cls;
                float4 bx;
                if (shuffled) {
                    cls = __shfl(my_cls, lane0 + t, 64);
                    bx = make_float4(__shfl(my_box.x, lane0 + t, 64), __shfl(my_box.y, lane0 + t, 64),
                                     __shfl(my_box.z, lane0 + t, 64), __shfl(my_box.w, lane0 + t, 64));
                } else {
                    const int64_t src = (b * TT + t) * N + n;
                    const int64_t cl = slot_class[src];
                    cls = (int)(cl < 0 ? 0 : (cl >= vocab ? vocab - 1 : cl));
                    bx = ld4(slot_box + src * 4);
                }
                st4(cls_acc + cls * d + c, f4_add(ld4(cls_acc + cls * d + c), g));
                t_acc[t] = f4_add(t_acc[t], g);
                b_acc = f4_add(b_acc, g);
                w_acc0 = f4_fma(bx.x, g, w_acc0); w_acc1 = f4_fma(bx.y, g, w_acc1);
                w_acc2 = f4_fma(bx.z, g, w_acc2); w_acc3 = f4_fma(bx.w, g, w_acc3);
            }
        }
    }
    float* slab = slabs + (int64_t)blockIdx.x * slab_stride;
    __syncthreads();
    if (grp == 0) {
        for (int v = 0; v < vocab; ++v) {
            float4 s = f4_zero();
            for (int k = 0; k < G; ++k) s = f4_add(s, ld4(lds + ((int64_t)k * rows + v) * d + c));
            st4(slab + v * d + c, s);
        }
    }
    __syncthreads();
    // second round through the same LDS: the register accumulators (TT frame rows, bias, 4 box columns)
    if (live) {
        float* mine = lds + (int64_t)grp * rows * d;
#pragma unroll
        for (int t = 0; t < TT; ++t) st4(mine + t * d + c, t_acc[t]);
        st4(mine + (TT + 0) * d + c, b_acc);
        st4(mine + (TT + 1) * d + c, w_acc0); st4(mine + (TT + 2) * d + c, w_acc1);
        st4(mine + (TT + 3) * d + c, w_acc2); st4(mine + (TT + 4) * d + c, w_acc3);
    }
    __syncthreads();
    if (grp == 0) {
        float4 r[TT + 5];
#pragma unroll
        for (int j = 0; j < TT + 5; ++j) {
            float4 s = f4_zero();
            for (int k = 0; k < G; ++k) s = f4_add(s, ld4(lds + ((int64_t)k * rows + j) * d + c));
            r[j] = s;
        }
        int64_t off = (int64_t)vocab * d;
        // box_w is [d][4]: channel c+j's row = (sum g*bx.x, sum g*bx.y, sum g*bx.z, sum g*bx.w) of that channel
        st4(slab + off + (int64_t)(c + 0) * 4, make_float4(r[TT + 1].x, r[TT + 2].x, r[TT + 3].x, r[TT + 4].x));
        st4(slab + off + (int64_t)(c + 1) * 4, make_float4(r[TT + 1].y, r[TT + 2].y, r[TT + 3].y, r[TT + 4].y));
        st4(slab + off + (int64_t)(c + 2) * 4, make_float4(r[TT + 1].z, r[TT + 2].z, r[TT + 3].z, r[TT + 4].z));
        st4(slab + off + (int64_t)(c + 3) * 4, make_float4(r[TT + 1].w, r[TT + 2].w, r[TT + 3].w, r[TT + 4].w));
        off += (int64_t)d * 4;
        st4(slab + off + c, r[TT]);
        off += d;
#pragma unroll
        for (int t = 0; t < TT; ++t) st4(slab + off + (int64_t)t * d + c, r[t]);
    }
}

extern "C" int vlg_embed_fwd(const int64_t* slot_class, const float* slot_box, const float* cls_emb,
                             const float* box_w, const float* box_b, const float* time_emb, float* x,
                             int B, int T, int N, int d, int vocab, void* stream) {
    if (B < 1 || T < 1 || N < 1 || d < 4 || (d & 3) || vocab < 1) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(slot_box) || !vlg_aligned16(cls_emb) || !vlg_aligned16(box_w) ||
        !vlg_aligned16(box_b) || !vlg_aligned16(time_emb) || !vlg_aligned16(x)) return VLG_ERR_ALIGN;
    if (d > 1024) return VLG_ERR_SHAPE;                       // d/4 threads of a 256-thread block own one row
    const int rpb = 1024 / d;                                 // rows in flight per block (d <= 1024)
    const int64_t rows = (int64_t)B * T * N;
    int64_t blocks = (rows + rpb - 1) / rpb;
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
    int G = 1024 / d;                                     // groups per block: d/4 threads each
    while (G > 1 && (size_t)G * rows * d * sizeof(float) > 96 * 1024) G >>= 1;
    const size_t lds = (size_t)G * rows * d * sizeof(float);
    if (lds > 160 * 1024 || (slab_stride & 3) || (((int64_t)vocab * d) & 3)) return VLG_ERR_SHAPE;
    const dim3 grid(EMBED_BWD_SLABS), block(256);
    hipStream_t s = (hipStream_t)stream;
#define EMBED_BWD_LAUNCH(TT)                                                                                    \
    {                                                                                                           \
        static size_t granted = 0;        /* the attribute call costs tens of microseconds of host time: once per size */ \
        if (lds > granted) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<TT>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds);                                                                      \
            granted = lds;                                                                                      \
        }                                                                                                       \
        hipLaunchKernelGGL(embed_bwd_kernel<TT>, grid, block, lds, s, dx, slot_class, slot_box, slabs,          \
                           slab_stride, B, N, d, vocab, rows, G);                                               \
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
