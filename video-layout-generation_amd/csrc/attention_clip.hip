// Per-clip temporal encoder core (option attention = "clip"): block-causal softmax attention over ALL T*N tokens of a
// clip, per (clip, head) - token (t, n) attends to every slot of frames <= t; padded slots (valid == 0) are never keys,
// except for themselves.  SELF-ORACLE (the CPU specification clip_attention under oracle/): the reference has no attention at all; this is
// the "LDS-staged per-clip tiles" reading of BASELINE.json's temporal encoder, next to the per-slot default (attention.hip).
//
// Flash-style, exact fp32 on v_mfma_f32_32x32x2_f32, nothing quadratic ever leaves the chip.
//   token order   inside a clip the kernels walk tokens FRAME-major, s = t*N + n (memory row of s: (b*N + n)*T + t), so the
//                 causal structure is block-lower-triangular in s and whole 32-key tiles are either visible, hidden or (on the
//                 frame boundary / with padded slots) masked element-wise
//   forward       a block = 4 waves x 32 queries; 32-key tiles of K (row-major) and V (transposed) are staged through LDS once
//                 per block (register-prefetched one tile ahead); per tile and wave:
//                     S^T = K . Q^T          32 MFMAs   A = K rows from LDS, B = the wave's Q fragment (registers, pre-scaled by
//                                                       log2(e) / sqrt(64)); C layout: lane = query, registers = keys, so the
//                                                       online-softmax row statistics are IN-LANE reductions (+ one lane-xor-32)
//                     O  += P . V            32 MFMAs   the probability tile in C layout IS the A operand (lane = query row,
//                                                       k-slot = the register's key), B = V^T from LDS
//                 the running maximum's rescale factor lives per query LANE but scales O's query ROWS (registers): it is turned
//                 through a 128-B wave-private LDS line (1 write + 4 broadcast reads)
//   backward      two kernels, so that every gradient element has ONE owner and a fixed summation order (no atomics):
//                     dQ  (query owner)   S^T, dP^T = V . dO^T, dS^T = P (dP - delta) / 8, dQ += dS . K        96 MFMAs / tile
//                     dKV (key owner)     S, dP = dO . V^T, dV += P^T . dO, dK += dS^T . Q                      128 MFMAs / tile
//                 P is recomputed from the saved log-sum-exp (one float per query and head); delta = <dO, O> per query.
// Algorithmic work per (clip, head): N^2 T (T + 1) / 2 visible (query, key) pairs x 4 * 64 flop forward, x 2.5 that backward
// (5 products; the two-kernel backward executes 7).  Measured at (32, 16, 64), d = 256 (tools/clip_attn_bench.py, box to box
// +-4 %): forward 195-204 us = 90-94 TFLOP/s = 0.57-0.59 of the fp32 MFMA peak, backward 600-620 us = 74-76 TFLOP/s algorithmic
// (0.47-0.48; 0.66 counting the executed products); (8, 32, 64), d = 512: 0.63 / 0.52.
// Dealing the longest blocks first took the forward from 314 to 210 us and the backward from 1 076 to 647 us (a 32-tile block
// dealt last runs alone for 80 us); occupancy did the rest: the key-owner kernel held to 256 registers (two waves per SIMD
// instead of one: 647 -> 599 us), the forward to 168 (three waves per SIMD, 8 registers spilled: 210 -> 195 us).  A lazily
// updated reference maximum (rescale O only when the maximum moves by > 2^6) was measured: 213 us - the rescale is not what
// the matrix pipe waits for; not kept.
#include "common.h"
#include "gemm_tile.h"

#define CHD 64          /* head dim */
#define CKT 32          /* tokens per LDS tile */
#define CQB 128         /* owner tokens per block: 4 waves x 32 */
#define CLDR 68         /* row stride of a row-major [32][64] tile (floats) */
#define CLDT 36         /* row stride of a transposed [64][32] tile */
#define CINVALID 0x10000
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int64_t clip_row(int64_t b, int s, int T, int N) {
    const int t = s / N, n = s - t * N;
    return (b * N + n) * (int64_t)T + t;
}
// register r of a 32x32 C/D tile, lane half h  ->  row (r & 3) + 8 (r >> 2) + 4 h: the four registers 4g .. 4g+3 are rows
// 8g + 4h .. + 3, i.e. ONE float4 at offset 8g + 4h of a 32-entry per-row table
__device__ __forceinline__ void rows16(float (&v)[16], const float* tab, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = ld4(tab + 8 * g + 4 * h);
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
}
__device__ __forceinline__ void rows16i(int (&v)[16], const int* tab, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int4 t = *reinterpret_cast<const int4*>(tab + 8 * g + 4 * h);
        v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
    }
}
// one thread's share of a staged [32 tokens][64 dims] tile: token tid >> 3, float4 columns (tid & 7) and (tid & 7) + 8
struct ClipStage { float4 a, b; };
__device__ __forceinline__ ClipStage stage_load(const float* __restrict__ p, int c4) {
    ClipStage v;
    v.a = ld4(p + 4 * c4);
    v.b = ld4(p + 4 * (c4 + 8));
    return v;
}
__device__ __forceinline__ void stage_rows(float* Xs, int tok, int c4, const ClipStage& v) {
    st4(Xs + tok * CLDR + 4 * c4, v.a);
    st4(Xs + tok * CLDR + 4 * (c4 + 8), v.b);
}
__device__ __forceinline__ void stage_transposed(float* Xt, int tok, int c4, const ClipStage& v) {
    float* p = Xt + (4 * c4) * CLDT + tok;
    p[0] = v.a.x; p[CLDT] = v.a.y; p[2 * CLDT] = v.a.z; p[3 * CLDT] = v.a.w;
    p += 32 * CLDT;
    p[0] = v.b.x; p[CLDT] = v.b.y; p[2 * CLDT] = v.b.z; p[3 * CLDT] = v.b.w;
}
// D[row][col] = sum over the 64 dims of A[row][dim] B[col][dim], both operands as "X[l31][32 h + i]", i = 0 .. 31 (MFMA step i of
// lane (l31, h)): B = a register fragment, A = read from the row-major LDS tile as it is consumed (four values per ds_read_b128)
__device__ __forceinline__ f32x16 dot64_rows(const float* Xs, int l31, int h, const float (&b)[32]) {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    const float* p = Xs + l31 * CLDR + 32 * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = ld4(p + 4 * q);
        c = MFMA32(t.x, b[4 * q], c);
        c = MFMA32(t.y, b[4 * q + 1], c);
        c = MFMA32(t.z, b[4 * q + 2], c);
        c = MFMA32(t.w, b[4 * q + 3], c);
    }
    return c;
}
// acc[dt] (rows = the A operand's rows, cols = dims 32 dt + l31) += sum over the tile's 32 tokens of a[token] Xt[dim][token]
__device__ __forceinline__ void contract32(f32x16 (&acc)[2], const float (&a)[16], const float* Xt, int l31, int h) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = ld4(Xt + (l31 + 32 * dt) * CLDT + 8 * g + 4 * h);
            acc[dt] = MFMA32(a[4 * g], t.x, acc[dt]);
            acc[dt] = MFMA32(a[4 * g + 1], t.y, acc[dt]);
            acc[dt] = MFMA32(a[4 * g + 2], t.z, acc[dt]);
            acc[dt] = MFMA32(a[4 * g + 3], t.w, acc[dt]);
        }
}
// store a C-layout tile pair (rows = 32 tokens of this wave, cols = 64 dims) to rows rowtab[0..31] of a [rows, ld] matrix
__device__ __forceinline__ void store_rows(float* __restrict__ dst, int64_t ld, const int* rowtab, int h, int l31,
                                           const f32x16 (&acc)[2]) {
    int rows[16];
    rows16i(rows, rowtab, h);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float* p = dst + rows[r] * ld + l31;
        p[0] = acc[0][r];
        p[32] = acc[1][r];
    }
}

#define CLIP_SCALE_LOG2 0.18033688011112042f    /* log2(e) / sqrt(64) */
#define CLIP_SCALE 0.125f

// ---------------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(256, 3) void attn_clip_fwd_kernel(const float* __restrict__ qkv, const float* __restrict__ valid,
                                                            float* __restrict__ out, float* __restrict__ lse, int T, int N, int d) {
    __shared__ __attribute__((aligned(16))) float Ks[CKT * CLDR];
    __shared__ __attribute__((aligned(16))) float Vt[CHD * CLDT];
    __shared__ __attribute__((aligned(16))) int kfr[CKT];
    __shared__ __attribute__((aligned(16))) float tr[4][32];
    __shared__ __attribute__((aligned(16))) int qrow[CQB];
    const int S = T * N, heads = d / CHD;
    // blocks are dealt in order of blockIdx.x, then .y: (clip, head) runs fastest and the LONGEST query blocks (the last
    // frames: they see every key) come first, so the launch ends on short blocks instead of one 32-tile straggler
    const int hh = blockIdx.x % heads;
    const int64_t b = blockIdx.x / heads;
    const int q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * CQB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int nq = S - q0 < CQB ? S - q0 : CQB;
    const int64_t ld = 3 * (int64_t)d;
    if (tid < CQB) qrow[tid] = tid < nq ? (int)clip_row(b, q0 + tid, T, N) : 0;
    const bool wact = 32 * w < nq;                                   // wave-uniform (S is a multiple of 32)
    const int sq = q0 + 32 * w + l31;
    const int tq = wact ? sq / N : 0;
    const int tq_min = (q0 + 32 * w) / N, tq_max = wact ? (q0 + 32 * w + 31) / N : -1;
    float qf[32];
    if (wact) {
        const float* qp = qkv + clip_row(b, sq, T, N) * ld + hh * CHD + 32 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t = ld4(qp + 4 * c);
            qf[4 * c] = t.x * CLIP_SCALE_LOG2; qf[4 * c + 1] = t.y * CLIP_SCALE_LOG2;
            qf[4 * c + 2] = t.z * CLIP_SCALE_LOG2; qf[4 * c + 3] = t.w * CLIP_SCALE_LOG2;
        }
    }
    const int kend = ((q0 + nq - 1) / N + 1) * N;                    // keys of frames <= the block's last query frame
    const int ntiles = ((kend < S ? kend : S) + CKT - 1) / CKT;
    const int tok = tid >> 3, c4 = tid & 7;
    auto tile_load = [&](int j, ClipStage& kn, ClipStage& vn, int& fr) __attribute__((always_inline)) {
        const int sk = j * CKT + tok;
        const int tk = sk / N, nk = sk - tk * N;
        const float* p = qkv + ((b * N + nk) * (int64_t)T + tk) * ld + hh * CHD;
        kn = stage_load(p + d, c4);
        vn = stage_load(p + 2 * d, c4);
        fr = tk + ((valid != nullptr && valid[(b * T + tk) * N + nk] <= 0.f) ? CINVALID : 0);
    };
    ClipStage kn, vn;
    int frn;
    tile_load(0, kn, vn, frn);
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();                                             // the previous tile is consumed
        stage_rows(Ks, tok, c4, kn);
        stage_transposed(Vt, tok, c4, vn);
        if (c4 == 0) kfr[tok] = frn;
        __syncthreads();
        if (j + 1 < ntiles) tile_load(j + 1, kn, vn, frn);           // in flight under this tile's MFMAs
        const int k0 = j * CKT;
        if (!wact || k0 / N > tq_max) continue;                      // every key of the tile lies in a later frame
        f32x16 st = dot64_rows(Ks, l31, h, qf);                      // S^T[key (r, h)][query l31], log2 domain
        if (valid != nullptr || (k0 + CKT - 1) / N > tq_min) {
            int fr[16];
            rows16i(fr, kfr, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                st[r] = (fr[r] <= tq || key == sq) ? st[r] : -INFINITY;
            }
        }
        float tmax = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, st[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float p[16], rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(st[r] - m_use); rs += p[r]; }
        rs += __shfl_xor(rs, 32);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {                                  // rescale the O rows by their query's factor
            if (h == 0) tr[w][l31] = alpha;
            __builtin_amdgcn_wave_barrier();
            float ar[16];
            rows16(ar, tr[w], h);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] *= ar[r]; o[1][r] *= ar[r]; }
        }
        contract32(o, p, Vt, l31, h);                                // O[query][dim] += P[query][key] V[key][dim]
    }
    if (!wact) return;
    if (h == 0) {
        tr[w][l31] = 1.0f / l_run;
        lse[((b * heads + hh) * (int64_t)S) + sq] = m_run + __builtin_amdgcn_logf(l_run);       // log2 domain
    }
    __builtin_amdgcn_wave_barrier();
    float li[16];
    rows16(li, tr[w], h);
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= li[r]; o[1][r] *= li[r]; }
    store_rows(out + hh * CHD, d, qrow + 32 * w, h, l31, o);
}

// ------------------------------------------------------------------------------------------- backward, query owner (dQ)
__global__ __launch_bounds__(256) void attn_clip_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ valid,
                                                           const float* __restrict__ out, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           float* __restrict__ dqkv, int T, int N, int d) {
    __shared__ __attribute__((aligned(16))) float Ks[CKT * CLDR];
    __shared__ __attribute__((aligned(16))) float Vs[CKT * CLDR];
    __shared__ __attribute__((aligned(16))) float Kt[CHD * CLDT];
    __shared__ __attribute__((aligned(16))) int kfr[CKT];
    __shared__ __attribute__((aligned(16))) int qrow[CQB];
    const int S = T * N, heads = d / CHD;
    // blocks are dealt in order of blockIdx.x, then .y: (clip, head) runs fastest and the LONGEST query blocks (the last
    // frames: they see every key) come first, so the launch ends on short blocks instead of one 32-tile straggler
    const int hh = blockIdx.x % heads;
    const int64_t b = blockIdx.x / heads;
    const int q0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * CQB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int nq = S - q0 < CQB ? S - q0 : CQB;
    const int64_t ld = 3 * (int64_t)d;
    if (tid < CQB) qrow[tid] = tid < nq ? (int)clip_row(b, q0 + tid, T, N) : 0;
    const bool wact = 32 * w < nq;
    const int sq = q0 + 32 * w + l31;
    const int tq = wact ? sq / N : 0;
    const int tq_min = (q0 + 32 * w) / N, tq_max = wact ? (q0 + 32 * w + 31) / N : -1;
    float qf[32], dof[32];
    float lse_q = 0.f, del_q = 0.f;
    if (wact) {
        const int64_t row = clip_row(b, sq, T, N);
        const float* qp = qkv + row * ld + hh * CHD + 32 * h;
        const float* gp = dout + row * d + hh * CHD + 32 * h;
        const float* op = out + row * d + hh * CHD + 32 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t = ld4(qp + 4 * c), g = ld4(gp + 4 * c), y = ld4(op + 4 * c);
            qf[4 * c] = t.x * CLIP_SCALE_LOG2; qf[4 * c + 1] = t.y * CLIP_SCALE_LOG2;
            qf[4 * c + 2] = t.z * CLIP_SCALE_LOG2; qf[4 * c + 3] = t.w * CLIP_SCALE_LOG2;
            dof[4 * c] = g.x; dof[4 * c + 1] = g.y; dof[4 * c + 2] = g.z; dof[4 * c + 3] = g.w;
            del_q += g.x * y.x + g.y * y.y + g.z * y.z + g.w * y.w;
        }
        del_q += __shfl_xor(del_q, 32);                              // <dO, O> over the 64 dims of the head
        const int64_t li = (b * heads + hh) * (int64_t)S + sq;
        lse_q = lse[li];
        if (h == 0) delta[li] = del_q;                               // the key-owner kernel reads it
    }
    const int kend = ((q0 + nq - 1) / N + 1) * N;
    const int ntiles = ((kend < S ? kend : S) + CKT - 1) / CKT;
    const int tok = tid >> 3, c4 = tid & 7;
    auto tile_load = [&](int j, ClipStage& kn, ClipStage& vn, int& fr) __attribute__((always_inline)) {
        const int sk = j * CKT + tok;
        const int tk = sk / N, nk = sk - tk * N;
        const float* p = qkv + ((b * N + nk) * (int64_t)T + tk) * ld + hh * CHD;
        kn = stage_load(p + d, c4);
        vn = stage_load(p + 2 * d, c4);
        fr = tk + ((valid != nullptr && valid[(b * T + tk) * N + nk] <= 0.f) ? CINVALID : 0);
    };
    ClipStage kn, vn;
    int frn;
    tile_load(0, kn, vn, frn);
    f32x16 dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq[0][r] = 0.f; dq[1][r] = 0.f; }
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        stage_rows(Ks, tok, c4, kn);
        stage_transposed(Kt, tok, c4, kn);
        stage_rows(Vs, tok, c4, vn);
        if (c4 == 0) kfr[tok] = frn;
        __syncthreads();
        if (j + 1 < ntiles) tile_load(j + 1, kn, vn, frn);
        const int k0 = j * CKT;
        if (!wact || k0 / N > tq_max) continue;
        float ds[16];
        {
            const f32x16 st = dot64_rows(Ks, l31, h, qf);            // S^T, log2 domain
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(st[r] - lse_q);      // P^T
        }
        if (valid != nullptr || (k0 + CKT - 1) / N > tq_min) {
            int fr[16];
            rows16i(fr, kfr, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                ds[r] = (fr[r] <= tq || key == sq) ? ds[r] : 0.f;
            }
        }
        {
            const f32x16 dpt = dot64_rows(Vs, l31, h, dof);          // dP^T[key][query] = <V[key], dO[query]>
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = ds[r] * (dpt[r] - del_q) * CLIP_SCALE;       // dS^T
        }
        contract32(dq, ds, Kt, l31, h);                              // dQ[query][dim] += dS[query][key] K[key][dim]
    }
    if (!wact) return;
    store_rows(dqkv + hh * CHD, ld, qrow + 32 * w, h, l31, dq);
}

// --------------------------------------------------------------------------------------- backward, key owner (dK, dV)
__global__ __launch_bounds__(256, 2) void attn_clip_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ valid,
                                                            const float* __restrict__ dout, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, float* __restrict__ dqkv,
                                                            int T, int N, int d) {
    __shared__ __attribute__((aligned(16))) float Qs[CKT * CLDR];
    __shared__ __attribute__((aligned(16))) float Gs[CKT * CLDR];
    __shared__ __attribute__((aligned(16))) float Qt[CHD * CLDT];
    __shared__ __attribute__((aligned(16))) float Gt[CHD * CLDT];
    __shared__ __attribute__((aligned(16))) float qlse[CKT];
    __shared__ __attribute__((aligned(16))) float qdel[CKT];
    __shared__ __attribute__((aligned(16))) int qfr[CKT];
    __shared__ __attribute__((aligned(16))) int krow[CQB];
    const int S = T * N, heads = d / CHD;
    const int hh = blockIdx.x % heads;                               // (clip, head) fastest; the first key blocks (seen by every
    const int64_t b = blockIdx.x / heads;                            // later query) are the longest and come first
    const int kb0 = blockIdx.y * CQB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int nk = S - kb0 < CQB ? S - kb0 : CQB;
    const int64_t ld = 3 * (int64_t)d;
    if (tid < CQB) krow[tid] = tid < nk ? (int)clip_row(b, kb0 + tid, T, N) : 0;
    const bool wact = 32 * w < nk;
    const int sk = kb0 + 32 * w + l31;
    int kfr_own = 0;                                                 // this lane's key: frame (+ CINVALID if padded)
    const int tk_min = (kb0 + 32 * w) / N;
    float kf[32], vf[32];
    if (wact) {
        const int tk = sk / N, nn = sk - tk * N;
        kfr_own = tk + ((valid != nullptr && valid[(b * T + tk) * N + nn] <= 0.f) ? CINVALID : 0);
        const float* kp = qkv + ((b * N + nn) * (int64_t)T + tk) * ld + hh * CHD + 32 * h;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 t = ld4(kp + d + 4 * c), u = ld4(kp + 2 * d + 4 * c);
            kf[4 * c] = t.x * CLIP_SCALE_LOG2; kf[4 * c + 1] = t.y * CLIP_SCALE_LOG2;
            kf[4 * c + 2] = t.z * CLIP_SCALE_LOG2; kf[4 * c + 3] = t.w * CLIP_SCALE_LOG2;
            vf[4 * c] = u.x; vf[4 * c + 1] = u.y; vf[4 * c + 2] = u.z; vf[4 * c + 3] = u.w;
        }
    }
    const int jfirst = ((kb0 / N) * N) / CKT;                        // first query tile holding a frame >= the block's first key frame
    const int ntiles = S / CKT;
    const int tok = tid >> 3, c4 = tid & 7;
    const int64_t lbase = (b * heads + hh) * (int64_t)S;
    auto tile_load = [&](int j, ClipStage& qn, ClipStage& gn, float& ls, float& dl, int& fr) __attribute__((always_inline)) {
        const int sq = j * CKT + tok;
        const int tq = sq / N, nq_ = sq - tq * N;
        const int64_t row = (b * N + nq_) * (int64_t)T + tq;
        qn = stage_load(qkv + row * ld + hh * CHD, c4);
        gn = stage_load(dout + row * d + hh * CHD, c4);
        ls = lse[lbase + sq];
        dl = delta[lbase + sq];
        fr = tq;
    };
    ClipStage qn, gn;
    float lsn, dln;
    int frn;
    tile_load(jfirst, qn, gn, lsn, dln, frn);
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    for (int j = jfirst; j < ntiles; ++j) {
        __syncthreads();
        stage_rows(Qs, tok, c4, qn);
        stage_transposed(Qt, tok, c4, qn);
        stage_rows(Gs, tok, c4, gn);
        stage_transposed(Gt, tok, c4, gn);
        if (c4 == 0) { qlse[tok] = lsn; qdel[tok] = dln; qfr[tok] = frn; }
        __syncthreads();
        if (j + 1 < ntiles) tile_load(j + 1, qn, gn, lsn, dln, frn);
        const int q0 = j * CKT;
        if (!wact || (q0 + CKT - 1) / N < tk_min) continue;          // every query of the tile lies in an earlier frame
        float p[16], ds[16];
        {
            const f32x16 s = dot64_rows(Qs, l31, h, kf);             // S[query (r, h)][key l31], log2 domain
            float ls[16];
            rows16(ls, qlse, h);
            int fr[16];
            rows16i(fr, qfr, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int query = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool ok = kfr_own <= fr[r] || query == sk;
                p[r] = ok ? __builtin_amdgcn_exp2f(s[r] - ls[r]) : 0.f;
            }
        }
        {
            const f32x16 dp = dot64_rows(Gs, l31, h, vf);            // dP[query][key] = <dO[query], V[key]>
            float dl[16];
            rows16(dl, qdel, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = p[r] * (dp[r] - dl[r]) * CLIP_SCALE;
        }
        contract32(dv, p, Gt, l31, h);                               // dV[key][dim] += P[query][key] dO[query][dim]
        contract32(dk, ds, Qt, l31, h);                              // dK[key][dim] += dS[query][key] Q[query][dim]
    }
    if (!wact) return;
    store_rows(dqkv + d + hh * CHD, ld, krow + 32 * w, h, l31, dk);
    store_rows(dqkv + 2 * d + hh * CHD, ld, krow + 32 * w, h, l31, dv);
}

// ---------------------------------------------------------------------------------------------------------- C ABI
static int clip_check(int64_t B, int T, int N, int d) {
    if (B < 1 || T < 1 || N < 1 || d < CHD || (d % CHD) != 0) return VLG_ERR_SHAPE;
    const int64_t S = (int64_t)T * N;
    if ((S % CKT) != 0 || S > (1 << 20) || B * (d / CHD) >= (1ll << 31)) return VLG_ERR_SHAPE;
    if (B * S >= (1ll << 31)) return VLG_ERR_SHAPE;                  // row numbers are ints
    return 0;
}

extern "C" int vlg_attention_clip_fwd(const float* qkv, const float* valid, float* out, float* lse, int64_t B, int T, int N,
                                      int d, void* stream) {
    if (const int rc = clip_check(B, T, N, d)) return rc;
    if (!vlg_aligned16(qkv) || !vlg_aligned16(out) || !lse) return VLG_ERR_ALIGN;
    const int S = T * N;
    const dim3 grid((unsigned)(B * (d / CHD)), (unsigned)((S + CQB - 1) / CQB));
    hipLaunchKernelGGL(attn_clip_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, valid, out, lse, T, N, d);
    return vlg_last_error();
}

extern "C" int vlg_attention_clip_bwd(const float* qkv, const float* valid, const float* out, const float* dout,
                                      const float* lse, float* delta, float* dqkv, int64_t B, int T, int N, int d, void* stream) {
    if (const int rc = clip_check(B, T, N, d)) return rc;
    if (!vlg_aligned16(qkv) || !vlg_aligned16(out) || !vlg_aligned16(dout) || !vlg_aligned16(dqkv) || !lse || !delta) return VLG_ERR_ALIGN;
    const int S = T * N;
    const dim3 grid((unsigned)(B * (d / CHD)), (unsigned)((S + CQB - 1) / CQB));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_clip_dq_kernel, grid, dim3(256), 0, s, qkv, valid, out, dout, lse, delta, dqkv, T, N, d);
    hipLaunchKernelGGL(attn_clip_dkv_kernel, grid, dim3(256), 0, s, qkv, valid, dout, lse, delta, dqkv, T, N, d);
    return vlg_last_error();
}
