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
#include <stdlib.h>

#define EMBED_BWD_SLABS 256

// A block owns SPB whole (clip, slot) sequences (one at the metric shape: T rows of d floats = a contiguous 16 KB tile of x).
// Thread (slot, q) owns channels 4q .. 4q+3 of the rows slot, slot + rpb, ... of that tile (d/4 threads per row, 1024/d rows
// per pass): the box projection rows and the bias of those channels are loaded ONCE, the row's class id and box are one
// broadcast load each, and per row the thread does two coalesced 16-B table reads (class row, frame row) and one coalesced
// 16-B store.  Row -> (clip, frame, slot) costs one 32-bit division per ROW PASS now: round 2 walked a flat token index
// with three 64-bit divisions per row (hundreds of vector instructions each) - the kernel was ALU-bound at 21 % of the
// HBM roofline, not latency-bound as believed; round 1 re-read 4 box_w float4 + ids + box per float4 written.
__global__ __launch_bounds__(256) void embed_fwd_kernel(
    const int64_t* __restrict__ slot_class, const float* __restrict__ slot_box,
    const float* __restrict__ cls_emb, const float* __restrict__ box_w,
    const float* __restrict__ box_b, const float* __restrict__ time_emb,
    float* __restrict__ x, int n_seq, int T, int N, int d, int vocab, int spb) {
    const int lpr = d >> 2;                                   // threads per row
    const int q = threadIdx.x % lpr, slot = threadIdx.x / lpr, rpb = blockDim.x / lpr;
    if (slot >= rpb) return;
    const int c = q << 2;
    const float4 w0 = ld4(box_w + (int64_t)(c + 0) * 4), w1 = ld4(box_w + (int64_t)(c + 1) * 4);
    const float4 w2 = ld4(box_w + (int64_t)(c + 2) * 4), w3 = ld4(box_w + (int64_t)(c + 3) * 4);
    const float4 bb = ld4(box_b + c);
    const unsigned rows = (unsigned)spb * (unsigned)T, uT = (unsigned)T, uN = (unsigned)N;
#pragma unroll 4
    for (unsigned r = (unsigned)slot; r < rows; r += (unsigned)rpb) {
        const unsigned sl = r / uT, t = r - sl * uT;
        const unsigned seq = blockIdx.x * (unsigned)spb + sl;
        if (seq >= (unsigned)n_seq) break;
        const unsigned b = seq / uN, n = seq - b * uN;
        const int64_t src = ((int64_t)b * T + t) * N + n;
        int64_t cls = slot_class[src];
        cls = cls < 0 ? 0 : (cls >= vocab ? vocab - 1 : cls);   // host validates; never fault
        const float4 bx = ld4(slot_box + src * 4);
        float4 acc = f4_add(ld4(cls_emb + cls * d + c), f4_add(ld4(time_emb + (int64_t)t * d + c), bb));
        acc.x += bx.x * w0.x + bx.y * w0.y + bx.z * w0.z + bx.w * w0.w;
        acc.y += bx.x * w1.x + bx.y * w1.y + bx.z * w1.z + bx.w * w1.w;
        acc.z += bx.x * w2.x + bx.y * w2.y + bx.z * w2.z + bx.w * w2.w;
        acc.w += bx.x * w3.x + bx.y * w3.y + bx.z * w3.z + bx.w * w3.w;
        st4(x + ((int64_t)seq * T + t) * d + c, acc);
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

template <int TT, bool PF>
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
        // the rows, ids and boxes of the NEXT sequence are requested before the current one is accumulated (two sequences
        // per group at the metric shape: the second one's HBM latency hides behind the first one's LDS read-modify-writes)
        const int64_t stride = (int64_t)gridDim.x * G;
        int64_t seq = (int64_t)blockIdx.x * G + grp;
        float4 gv[TT], gn[TT];
        int my_cls = 0, nx_cls = 0;
        float4 my_box = f4_zero(), nx_box = f4_zero();
        auto fetch = [&](int64_t sq, float4 (&rows_)[TT], int& cls_, float4& box_) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < TT; ++t) rows_[t] = ld4(dx + (sq * TT + t) * d + c);       // TT independent 16-B loads in flight
            if (shuffled && tl < TT) {
                const int64_t b = sq / N;
                const int64_t src = (b * TT + tl) * N + (sq - b * N);
                const int64_t cl = slot_class[src];
                cls_ = (int)(cl < 0 ? 0 : (cl >= vocab ? vocab - 1 : cl));
                box_ = ld4(slot_box + src * 4);
            }
        };
        constexpr bool PREFETCH = PF && TT <= 16;          // (TT = 32: two row sets would not fit the register file)
        if (PREFETCH && seq < n_seq) fetch(seq, gv, my_cls, my_box);
        for (; seq < n_seq; seq += stride) {
            const int64_t b = seq / N;
            const int n = (int)(seq - b * N);
            if constexpr (!PREFETCH) fetch(seq, gv, my_cls, my_box);
            const bool more = PREFETCH && seq + stride < n_seq;
            if (more) fetch(seq + stride, gn, nx_cls, nx_box);
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
            if (more) {
#pragma unroll
                for (int t = 0; t < TT; ++t) gv[t] = gn[t];
                my_cls = nx_cls;
                my_box = nx_box;
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
    const int64_t n_seq = (int64_t)B * N;
    if (n_seq * T >= (1ll << 31)) return VLG_ERR_SHAPE;       // 32-bit row arithmetic in the kernel
    const int spb = rpb > T ? rpb / T : 1;                    // whole sequences per block (short clips: several)
    const int64_t blocks = (n_seq + spb - 1) / spb;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       slot_class, slot_box, cls_emb, box_w, box_b, time_emb, x, (int)n_seq, T, N, d, vocab, spb);
    return vlg_last_error();
}

extern "C" int vlg_embed_bwd_slabs(void) { return EMBED_BWD_SLABS; }          // the most any launch writes

static int embed_bwd_groups(int T, int d, int vocab) {
    const int rows = vocab > T + 5 ? vocab : T + 5;
    int G = 1024 / d;                                     // groups per block: d/4 threads each
    while (G > 1 && (size_t)G * rows * d * sizeof(float) > 96 * 1024) G >>= 1;
    return G;
}
// slabs (= blocks) of a launch: one group per sequence until the chip is full (few clips: fewer blocks, fewer slabs to write
// and to reduce - the 4-clip shard has 256 sequences = 64 blocks, and wrote 256 slabs = 11 MB before)
extern "C" int vlg_embed_bwd_slabs_for(int B, int T, int N, int d, int vocab) {
    if (B < 1 || N < 1 || d < 64 || d > 1024 || vocab < 1) return EMBED_BWD_SLABS;
    const int G = embed_bwd_groups(T, d, vocab);
    const int64_t want = ((int64_t)B * N + G - 1) / G;
    return (int)(want < EMBED_BWD_SLABS ? want : EMBED_BWD_SLABS);
}

extern "C" int vlg_embed_bwd(const float* dx, const int64_t* slot_class, const float* slot_box,
                             float* slabs, int64_t slab_stride, int64_t slab_capacity, int B, int T, int N, int d,
                             int vocab, void* stream) {
    if (B < 1 || N < 1 || d < 64 || d > 1024 || (d & 63) || vocab < 1) return VLG_ERR_SHAPE;
    const int n_blocks = vlg_embed_bwd_slabs_for(B, T, N, d, vocab);
    if (slab_capacity < (int64_t)n_blocks * slab_stride) return VLG_ERR_SHAPE;             // the caller's buffer must hold every slab
    const int64_t need = (int64_t)vocab * d + (int64_t)d * 4 + d + (int64_t)T * d;
    if (slab_stride < need || (slab_stride & 3)) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(slot_box) || !vlg_aligned16(slabs)) return VLG_ERR_ALIGN;
    const int rows = vocab > T + 5 ? vocab : T + 5;
    const int G = embed_bwd_groups(T, d, vocab);
    const size_t lds = (size_t)G * rows * d * sizeof(float);
    if (lds > 160 * 1024 || (slab_stride & 3) || (((int64_t)vocab * d) & 3)) return VLG_ERR_SHAPE;
    const dim3 grid((unsigned)n_blocks), block(256);
    // VLG_EMBED_PREFETCH=1 (read once): request the next sequence's rows before accumulating the current one.  Measured
    // (tools/kernel_bench.py, one box): 20.3 us against 19.3 us without at the metric shape - the second row set takes the
    // kernel to 256 registers and the accumulation is not what the loads wait for.  OFF.
    static int prefetch = -1;
    if (prefetch < 0) { const char* e = getenv("VLG_EMBED_PREFETCH"); prefetch = e ? atoi(e) : 0; }
    hipStream_t s = (hipStream_t)stream;
#define EMBED_BWD_LAUNCH(TT)                                                                                    \
    {                                                                                                           \
        static size_t granted = 0;        /* the attribute call costs tens of microseconds of host time: once per size */ \
        if (lds > granted) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<TT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds);                                                                      \
            (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<TT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds);                                                                      \
            granted = lds;                                                                                      \
        }                                                                                                       \
        if (prefetch) hipLaunchKernelGGL((embed_bwd_kernel<TT, true>), grid, block, lds, s, dx, slot_class, slot_box, slabs, \
                                         slab_stride, B, N, d, vocab, rows, G);                                 \
        else hipLaunchKernelGGL((embed_bwd_kernel<TT, false>), grid, block, lds, s, dx, slot_class, slot_box, slabs,        \
                                slab_stride, B, N, d, vocab, rows, G);                                          \
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
