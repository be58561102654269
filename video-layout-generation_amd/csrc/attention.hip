// Temporal encoder core: causal softmax attention along T per (clip, slot, head).
//
// Rows are in the internal order m = (b*N + n)*T + t, so the T frames of one slot are T
// consecutive rows of qkv [rows, 3d]: one wavefront stages the T x 64 q, k, v tiles of ONE
// head in LDS (row stride 68 floats: 16-B aligned and float4-conflict-free), computes the
// T x T scores with lanes laid out (row i, column group), reduces each softmax row with
// wavefront shuffles over the LPR lanes that share it, and writes the T x 64 output tile.
// Everything is fp32 VALU work: 4*T*T*64 flop per head is negligible next to the 1 KB/row it
// streams, so the kernel is HBM-bound.  Algorithmic bytes per (slot, head): fwd 4*T*64*4
// (q,k,v in, o out); bwd 7*T*64*4 (q,k,v,do in; dq,dk,dv out).
// Nothing is saved for backward: P is recomputed from q,k (cheaper than 4*T*T bytes of traffic).
#include "common.h"

#define HD 64
#define LDT 68     // LDS row stride of a T x 64 tile

template <int T>
struct AttnGeom {
    static constexpr int LPR = (T >= 8) ? 64 / T : T;   // lanes sharing one score row
    static constexpr int JPL = T / LPR;                 // score columns per lane
    static constexpr int ACTIVE = T * LPR;              // lanes that own a row (T=4: 16)
    static constexpr int CPL = HD / LPR;                // output channels per lane
};

// stage a T x 64 tile whose rows are `ld` floats apart into LDS
template <int T, typename EL>
__device__ __forceinline__ void stage_tile(float* S, const EL* __restrict__ src, int64_t ld, int lane) {
#pragma unroll
    for (int r = 0; r < (T * 16 + 63) / 64; ++r) {
        const int idx = r * 64 + lane;
        if (idx < T * 16) {
            const int row = idx >> 4, c4 = idx & 15;
            st4(S + row * LDT + c4 * 4, ld4(src + row * ld + c4 * 4));
        }
    }
}

// s[jj] = scale * <A[i,:], Bm[j,:]> for j = jg*JPL + jj, causal (j <= i) else -inf
template <int T>
__device__ __forceinline__ void score_rows(const float* A, const float* Bm, int i, int jg, float scale,
                                           float (&s)[AttnGeom<T>::JPL]) {
    constexpr int JPL = AttnGeom<T>::JPL;
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) s[jj] = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
        const float4 a = ld4(A + i * LDT + c4 * 4);
#pragma unroll
        for (int jj = 0; jj < JPL; ++jj) {
            const float4 b = ld4(Bm + (jg * JPL + jj) * LDT + c4 * 4);
            s[jj] += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
    }
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) s[jj] = (jg * JPL + jj <= i) ? s[jj] * scale : -INFINITY;
}

template <int T>
__device__ __forceinline__ void softmax_rows(float (&s)[AttnGeom<T>::JPL]) {
    constexpr int JPL = AttnGeom<T>::JPL, LPR = AttnGeom<T>::LPR;
    float mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) mx = fmaxf(mx, s[jj]);
    mx = group_max<LPR>(mx);
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) { s[jj] = expf(s[jj] - mx); sum += s[jj]; }
    sum = group_sum<LPR>(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jj = 0; jj < JPL; ++jj) s[jj] *= inv;
}

template <int T, typename EL = float>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const EL* __restrict__ qkv, EL* __restrict__ o,
                                                      int d, int n_heads) {
    using G = AttnGeom<T>;
    __shared__ __attribute__((aligned(16))) float sm[3 * T * LDT + T * (T + 1)];
    float* Qs = sm; float* Ks = sm + T * LDT; float* Vs = sm + 2 * T * LDT; float* Ps = sm + 3 * T * LDT;
    const int lane = threadIdx.x;
    const int64_t seq = blockIdx.x / n_heads;
    const int hh = blockIdx.x - (int)(seq * n_heads);
    const int64_t ld = 3 * (int64_t)d;
    const EL* base = qkv + seq * T * ld + hh * HD;
    stage_tile<T>(Qs, base, ld, lane);
    stage_tile<T>(Ks, base + d, ld, lane);
    stage_tile<T>(Vs, base + 2 * d, ld, lane);
    __syncthreads();
    const bool active = lane < G::ACTIVE;
    const int i = active ? lane / G::LPR : 0, jg = lane % G::LPR;
    float s[G::JPL];
    score_rows<T>(Qs, Ks, i, jg, 0.125f, s);          // 1/sqrt(64)
    softmax_rows<T>(s);
    if (active) {
#pragma unroll
        for (int jj = 0; jj < G::JPL; ++jj) Ps[i * (T + 1) + jg * G::JPL + jj] = s[jj];
    }
    __syncthreads();
    if (active) {
        float4 acc[G::CPL / 4];
#pragma unroll
        for (int c = 0; c < G::CPL / 4; ++c) acc[c] = f4_zero();
        for (int j = 0; j <= i; ++j) {
            const float p = Ps[i * (T + 1) + j];
#pragma unroll
            for (int c = 0; c < G::CPL / 4; ++c) {
                const float4 v = ld4(Vs + j * LDT + jg * G::CPL + c * 4);
                acc[c].x += p * v.x; acc[c].y += p * v.y; acc[c].z += p * v.z; acc[c].w += p * v.w;
            }
        }
        EL* dst = o + (seq * T + i) * (int64_t)d + hh * HD + jg * G::CPL;
#pragma unroll
        for (int c = 0; c < G::CPL / 4; ++c) st4(dst + c * 4, acc[c]);
    }
}

template <int T, typename EL = float>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const EL* __restrict__ qkv, const EL* __restrict__ dout,
                                                      EL* __restrict__ dqkv, int d, int n_heads) {
    using G = AttnGeom<T>;
    __shared__ __attribute__((aligned(16))) float sm[4 * T * LDT + 2 * T * (T + 1)];
    float* Qs = sm; float* Ks = sm + T * LDT; float* Vs = sm + 2 * T * LDT; float* Os = sm + 3 * T * LDT;
    float* Ps = sm + 4 * T * LDT; float* Ds = Ps + T * (T + 1);
    const int lane = threadIdx.x;
    const int64_t seq = blockIdx.x / n_heads;
    const int hh = blockIdx.x - (int)(seq * n_heads);
    const int64_t ld = 3 * (int64_t)d;
    const EL* base = qkv + seq * T * ld + hh * HD;
    stage_tile<T>(Qs, base, ld, lane);
    stage_tile<T>(Ks, base + d, ld, lane);
    stage_tile<T>(Vs, base + 2 * d, ld, lane);
    stage_tile<T>(Os, dout + seq * T * (int64_t)d + hh * HD, d, lane);
    __syncthreads();
    const bool active = lane < G::ACTIVE;
    const int i = active ? lane / G::LPR : 0, jg = lane % G::LPR;
    const float scale = 0.125f;
    float p[G::JPL], dp[G::JPL];
    score_rows<T>(Qs, Ks, i, jg, scale, p);
    softmax_rows<T>(p);
    score_rows<T>(Os, Vs, i, jg, 1.0f, dp);           // dP = dO . V^T (masked entries are -inf: unused)
    float delta = 0.f;
#pragma unroll
    for (int jj = 0; jj < G::JPL; ++jj) {
        const bool in = jg * G::JPL + jj <= i;
        dp[jj] = in ? dp[jj] : 0.f;
        delta += p[jj] * dp[jj];
    }
    delta = group_sum<G::LPR>(delta);
    if (active) {
#pragma unroll
        for (int jj = 0; jj < G::JPL; ++jj) {
            const int j = jg * G::JPL + jj;
            Ps[i * (T + 1) + j] = p[jj];
            Ds[i * (T + 1) + j] = p[jj] * (dp[jj] - delta) * scale;   // dS, pre-scaled
        }
    }
    __syncthreads();
    if (active) {
        // lane (row r = i, channel group jg): dQ[r] = sum_{j<=r} dS[r,j] K[j];
        // dK[r] = sum_{i>=r} dS[i,r] Q[i];  dV[r] = sum_{i>=r} P[i,r] dO[i]
        const int r = i, c0 = jg * G::CPL;
        float4 aq[G::CPL / 4], ak[G::CPL / 4], av[G::CPL / 4];
#pragma unroll
        for (int c = 0; c < G::CPL / 4; ++c) { aq[c] = f4_zero(); ak[c] = f4_zero(); av[c] = f4_zero(); }
        for (int j = 0; j <= r; ++j) {
            const float ds = Ds[r * (T + 1) + j];
#pragma unroll
            for (int c = 0; c < G::CPL / 4; ++c) {
                const float4 k = ld4(Ks + j * LDT + c0 + c * 4);
                aq[c].x += ds * k.x; aq[c].y += ds * k.y; aq[c].z += ds * k.z; aq[c].w += ds * k.w;
            }
        }
        for (int ii = r; ii < T; ++ii) {
            const float ds = Ds[ii * (T + 1) + r];
            const float pp = Ps[ii * (T + 1) + r];
#pragma unroll
            for (int c = 0; c < G::CPL / 4; ++c) {
                const float4 q = ld4(Qs + ii * LDT + c0 + c * 4);
                const float4 g = ld4(Os + ii * LDT + c0 + c * 4);
                ak[c].x += ds * q.x; ak[c].y += ds * q.y; ak[c].z += ds * q.z; ak[c].w += ds * q.w;
                av[c].x += pp * g.x; av[c].y += pp * g.y; av[c].z += pp * g.z; av[c].w += pp * g.w;
            }
        }
        EL* dst = dqkv + (seq * T + r) * ld + hh * HD + c0;
#pragma unroll
        for (int c = 0; c < G::CPL / 4; ++c) {
            st4(dst + c * 4, aq[c]);
            st4(dst + d + c * 4, ak[c]);
            st4(dst + 2 * d + c * 4, av[c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// T = 16 fast path: no LDS at all.  A 16 x 16 score tile is exactly one v_mfma_f32_16x16x4_f32 tile,
// so one wavefront keeps q, k, v (and dO) of its (slot, head) in registers, loaded straight from HBM
// in the MFMA operand layouts, and every product is an exact-fp32 MFMA chain:
//   row-style load   lane (r = l&15, g = l>>4) holds X[r][16f + 4g + e]   (f = float4 index, e = element)
//                    -> as operand A it is X[row r][k], as operand B it is X[col r][k]: the SAME
//                    registers give  S^T = K.Q^T  (A = K, B = Q)  and  S = Q.K^T  (A = Q, B = K)
//   k-style load     lane (c = l&15, g = l>>4) holds X[4g + rho][4c + e], rho = 0..3
//                    -> operand A of the products that contract over the 16 frames
//   C/D layout       col = l&15, row = 4*(l>>4) + reg.  A score tile in this layout is ALREADY the B
//                    operand of the next product (k-slot g <-> frame 4g + reg at MFMA step reg), so P
//                    never moves between lanes (guide: "an accumulator tile as the next MFMA's operand").
// With output-channel mapping  row c' of tile ct  <->  channel 4c' + ct, a lane ends up owning 16
// consecutive channels of one frame: four float4 stores, 256 B contiguous per frame.
// Softmax statistics are shuffle reductions (over lane groups for the transposed tile, over the 16
// lanes of a group for the plain tile).
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <typename EL>
__device__ __forceinline__ void load_rows16(float4 (&x)[4], const EL* __restrict__ base, int64_t ld, int r, int g) {
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = ld4(base + r * ld + 16 * f + 4 * g);
}
template <typename EL>
__device__ __forceinline__ void load_kmajor16(float4 (&x)[4], const EL* __restrict__ base, int64_t ld, int c, int g) {
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) x[rho] = ld4(base + (4 * g + rho) * ld + 4 * c);
}
// D = sum_c A[.][c] * B[.][c] over the 64 channels held row-style (two chains to halve the latency)
__device__ __forceinline__ f32x4 dot_rows16(const float4 (&a)[4], const float4 (&b)[4]) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        c0 = MFMA16(a[f].x, b[f].x, c0);
        c1 = MFMA16(a[f].y, b[f].y, c1);
        c0 = MFMA16(a[f].z, b[f].z, c0);
        c1 = MFMA16(a[f].w, b[f].w, c1);
    }
    return c0 + c1;
}
// out[ct] (tile of 16 channels 4c'+ct x 16 columns) += sum_rho A_k[rho][ct] * Bt[rho]
__device__ __forceinline__ void contract_frames16(f32x4 (&out)[4], const float4 (&ak)[4], const f32x4& bt) {
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
        out[0] = MFMA16(ak[rho].x, bt[rho], out[0]);
        out[1] = MFMA16(ak[rho].y, bt[rho], out[1]);
        out[2] = MFMA16(ak[rho].z, bt[rho], out[2]);
        out[3] = MFMA16(ak[rho].w, bt[rho], out[3]);
    }
}
// lane (col, g) stores its 16 consecutive channels 16g .. 16g+15 of row `col`
template <typename EL>
__device__ __forceinline__ void store_tiles16(EL* __restrict__ dst, const f32x4 (&o)[4]) {
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) st4(dst + 4 * rho, make_float4(o[0][rho], o[1][rho], o[2][rho], o[3][rho]));
}
// Same tile, stored through a wave-private LDS tile: the MFMA layout hands a lane 16 channels of one frame in four
// 16-B pieces 64 B apart, i.e. every direct store instruction writes 64 scattered 16-B pieces; turned through LDS
// (4 ds_write_b128 + 4 ds_read_b128, conflict-free at the 68-float stride) each store instruction writes four full
// 256-B rows.  `rows0` = first of the 16 rows (row stride ld elements), column base already applied.
template <typename EL>
__device__ __forceinline__ void store_tiles16_rows(float* __restrict__ Tl, EL* __restrict__ rows0, int64_t ld, int lane,
                                                   const f32x4 (&o)[4]) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) st4(Tl + r * LDT + 16 * g + 4 * rho, make_float4(o[0][rho], o[1][rho], o[2][rho], o[3][rho]));
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 4 * k + g;
        st4(rows0 + row * ld + 4 * r, ld4(Tl + row * LDT + 4 * r));
    }
    __builtin_amdgcn_wave_barrier();
}
// softmax of a TRANSPOSED tile: lane (i, g) holds scores of query i for keys j = 4g + reg
__device__ __forceinline__ void softmax_t16(f32x4& s, int i, int g, float scale) {
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = (4 * g + r <= i) ? s[r] * scale : -INFINITY; mx = fmaxf(mx, s[r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = expf(s[r] - mx); sum += s[r]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] *= inv;
}
// softmax of a PLAIN tile: lane (j, g) holds scores of key j for queries i = 4g + reg
__device__ __forceinline__ void softmax_p16(f32x4& s, int j, int g, float scale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = (j <= 4 * g + r) ? s[r] * scale : -INFINITY;
        const float mx = group_max<16>(v);
        v = expf(v - mx);
        s[r] = v / group_sum<16>(v);
    }
}

template <typename EL>
__device__ __forceinline__ void attn16_fwd_body(const EL* __restrict__ qkv, EL* __restrict__ o, int d, int n_heads, int64_t n_items) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;                             // wave-uniform
    const int64_t seq = item / n_heads;
    const int hh = (int)(item - seq * n_heads);
    const int r = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const EL* base = qkv + seq * 16 * ld + hh * HD;
    float4 qf[4], kf[4], vk[4];
    load_rows16(qf, base, ld, r, g);
    load_rows16(kf, base + d, ld, r, g);
    load_kmajor16(vk, base + 2 * d, ld, r, g);
    f32x4 pt = dot_rows16(kf, qf);                            // S^T[j = 4g+reg][i = r]
    softmax_t16(pt, r, g, 0.125f);
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    contract_frames16(acc, vk, pt);                           // O^T[c][i] = sum_j V[j][c] P^T[j][i]
    __shared__ __attribute__((aligned(16))) float out_tile[4][16 * LDT];
    store_tiles16_rows(out_tile[threadIdx.x >> 6], o + seq * 16 * (int64_t)d + hh * HD, (int64_t)d, lane, acc);
}
__global__ __launch_bounds__(256) void attn16_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                         int d, int n_heads, int64_t n_items) {
    attn16_fwd_body(qkv, o, d, n_heads, n_items);
}
__global__ __launch_bounds__(256) void attn16_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                              int d, int n_heads, int64_t n_items) {
    attn16_fwd_body(qkv, o, d, n_heads, n_items);
}

template <typename EL>
__device__ __forceinline__ void attn16_bwd_body(const EL* __restrict__ qkv, const EL* __restrict__ dout, EL* __restrict__ dqkv,
                                                int d, int n_heads, int64_t n_items) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int64_t seq = item / n_heads;
    const int hh = (int)(item - seq * n_heads);
    const int r = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const EL* base = qkv + seq * 16 * ld + hh * HD;
    const EL* gbase = dout + seq * 16 * (int64_t)d + hh * HD;
    const float scale = 0.125f;
    f32x4 p, pt, ds, dst_;                                     // plain / transposed probabilities and dS
    __shared__ __attribute__((aligned(16))) float turn_tile[4][16 * LDT];
    float* const T = turn_tile[threadIdx.x >> 6];
    float4 qf[4], kf[4], gf[4];
    {
        float4 vf[4];
        load_rows16(qf, base, ld, r, g);
        load_rows16(kf, base + d, ld, r, g);
        load_rows16(vf, base + 2 * d, ld, r, g);
        load_rows16(gf, gbase, d, r, g);
        pt = dot_rows16(kf, qf);                               // S^T : lane (i, g), keys 4g+reg
        p = dot_rows16(qf, kf);                                // S   : lane (j, g), queries 4g+reg
        f32x4 dpt = dot_rows16(vf, gf);                        // dP^T[j][i] = sum_c V[j][c] dO[i][c]
        f32x4 dp = dot_rows16(gf, vf);                         // dP  [i][j]
        softmax_t16(pt, r, g, scale);
        softmax_p16(p, r, g, scale);
        // delta_i = sum_j P[i][j] dP[i][j]
        float dl = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dl += pt[k] * dpt[k];       // masked entries have P = 0 (dP there is finite)
        dl += __shfl_xor(dl, 16);
        dl += __shfl_xor(dl, 32);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dst_[k] = pt[k] * (dpt[k] - dl) * scale;           // dS^T[j][i]
            const float dk = group_sum<16>(p[k] * dp[k]);      // delta of query 4g+k, summed over the 16 key lanes
            ds[k] = p[k] * (dp[k] - dk) * scale;               // dS[i][j]
        }
    }
    EL* obase = dqkv + seq * 16 * ld + hh * HD;
    f32x4 acc[4];
    float4 xk[4];
    // The three products below contract over the 16 frames, so they need dO, Q, K frame-major ("k-style").  Round 1 fetched
    // them from HBM a second time in that layout (7 tile loads for 4 tiles; PMC: 1.35x the algorithmic bytes); now the
    // row-style registers are turned through a 4 KB LDS tile private to the wave: 4 ds_write_b128 + 4 ds_read_b128 per
    // tile, both conflict-free at the 68-float row stride.  One wave owns the tile and DS operations of a wave execute in
    // order, so no barrier is needed - only the compiler must not reorder the accesses (wave_barrier).
    auto turn = [&](const float4 (&rows)[4]) {
#pragma unroll
        for (int f = 0; f < 4; ++f) st4(T + r * LDT + 16 * f + 4 * g, rows[f]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) xk[rho] = ld4(T + (4 * g + rho) * LDT + 4 * r);
        __builtin_amdgcn_wave_barrier();
    };
    // dV[j][c] = sum_i P[i][j] dO[i][c]
    turn(gf);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    contract_frames16(acc, xk, p);
    store_tiles16_rows(T, obase + 2 * d, ld, lane, acc);
    // dK[j][c] = sum_i dS[i][j] Q[i][c]
    turn(qf);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    contract_frames16(acc, xk, ds);
    store_tiles16_rows(T, obase + d, ld, lane, acc);
    // dQ[i][c] = sum_j dS^T[j][i] K[j][c]
    turn(kf);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    contract_frames16(acc, xk, dst_);
    store_tiles16_rows(T, obase, ld, lane, acc);
}
__global__ __launch_bounds__(256) void attn16_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                         float* __restrict__ dqkv, int d, int n_heads, int64_t n_items) {
    attn16_bwd_body(qkv, dout, dqkv, d, n_heads, n_items);
}
__global__ __launch_bounds__(256) void attn16_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              bf16_t* __restrict__ dqkv, int d, int n_heads, int64_t n_items) {
    attn16_bwd_body(qkv, dout, dqkv, d, n_heads, n_items);
}

// ------------------------------------------------------------------------------------------------
// T = 32: the T = 16 scheme on 2 x 2 score tiles of 16 x 16 (three of them under the causal mask).  One wavefront per
// (slot, head) keeps q, k, v of all 32 frames in registers in the same operand layouts; a query block's softmax runs over
// the key tiles it sees (one or two).  The LDS-tile vector-ALU kernel this replaces ran at 0.23 / 0.14 of the HBM
// roofline (forward / backward): 4 T T 64 flop per head on the vector ALU is no longer negligible at T = 32.
template <int NT>
__device__ __forceinline__ void softmax_t16n(f32x4 (&s)[NT], int i, int g, float scale) {
    // transposed tiles: lane (i, g) holds, for query i of the block, keys 4g + reg of key tile t; the LAST tile is the diagonal one
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = t + 1 < NT || 4 * g + r <= i;
            s[t][r] = ok ? s[t][r] * scale : -INFINITY;
            mx = fmaxf(mx, s[t][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[t][r] = expf(s[t][r] - mx); sum += s[t][r]; }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] *= inv;
}

template <typename EL>
__device__ __forceinline__ void attn32_fwd_body(const EL* __restrict__ qkv, EL* __restrict__ o, int d, int n_heads, int64_t n_items) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;                             // wave-uniform
    const int64_t seq = item / n_heads;
    const int hh = (int)(item - seq * n_heads);
    const int r = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const EL* base = qkv + seq * 32 * ld + hh * HD;
    float4 qf[2][4], kf[2][4], vk[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        load_rows16(qf[b], base + b * 16 * ld, ld, r, g);
        load_rows16(kf[b], base + b * 16 * ld + d, ld, r, g);
        load_kmajor16(vk[b], base + b * 16 * ld + 2 * d, ld, r, g);
    }
    __shared__ __attribute__((aligned(16))) float out_tile[4][16 * LDT];
    float* const Tl = out_tile[threadIdx.x >> 6];
    EL* const obase = o + seq * 32 * (int64_t)d + hh * HD;
    f32x4 acc[4];
    {   // queries 0..15 see keys 0..15
        f32x4 pt[1] = {dot_rows16(kf[0], qf[0])};
        softmax_t16n<1>(pt, r, g, 0.125f);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        contract_frames16(acc, vk[0], pt[0]);
        store_tiles16_rows(Tl, obase, (int64_t)d, lane, acc);
    }
    {   // queries 16..31 see keys 0..15 (all) and 16..31 (causal)
        f32x4 pt[2] = {dot_rows16(kf[0], qf[1]), dot_rows16(kf[1], qf[1])};
        softmax_t16n<2>(pt, r, g, 0.125f);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        contract_frames16(acc, vk[0], pt[0]);
        contract_frames16(acc, vk[1], pt[1]);
        store_tiles16_rows(Tl, obase + 16 * (int64_t)d, (int64_t)d, lane, acc);
    }
}
__global__ __launch_bounds__(256) void attn32_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                         int d, int n_heads, int64_t n_items) {
    attn32_fwd_body(qkv, o, d, n_heads, n_items);
}
__global__ __launch_bounds__(256) void attn32_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                              int d, int n_heads, int64_t n_items) {
    attn32_fwd_body(qkv, o, d, n_heads, n_items);
}

// plain tiles: lane (j, g) holds key j of key tile t for queries 4g + reg; the LAST tile is the diagonal one
template <int NT>
__device__ __forceinline__ void softmax_p16n(f32x4 (&s)[NT], int j, int g, float scale) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v[NT], mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            v[t] = (t + 1 < NT || j <= 4 * g + r) ? s[t][r] * scale : -INFINITY;
            mx = fmaxf(mx, group_max<16>(v[t]));
        }
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) { v[t] = expf(v[t] - mx); sum += group_sum<16>(v[t]); }
        const float inv = 1.0f / sum;
#pragma unroll
        for (int t = 0; t < NT; ++t) s[t][r] = v[t] * inv;
    }
}

// probabilities and score gradients of one query block against its NT key tiles, in both layouts:
//   p[t], ds[t]  plain (lane (j, g): key j, queries 4g + reg)        dst[t]  transposed (lane (i, g): query i, keys 4g + reg)
template <int NT>
__device__ __forceinline__ void attn_block_grads(const float4 (&q)[4], const float4 (&gq)[4], const float4 (*k)[4], const float4 (*v)[4],
                                                 int r, int g, float scale, f32x4 (&p)[NT], f32x4 (&ds)[NT], f32x4 (&dst)[NT]) {
    f32x4 pt[NT], dp[NT], dpt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        pt[t] = dot_rows16(k[t], q);                          // S^T : lane (i, g), keys 4g + reg of tile t
        p[t] = dot_rows16(q, k[t]);                           // S   : lane (j, g), queries 4g + reg
        dpt[t] = dot_rows16(v[t], gq);                        // dP^T[j][i] = sum_c V[j][c] dO[i][c]
        dp[t] = dot_rows16(gq, v[t]);                         // dP  [i][j]
    }
    softmax_t16n<NT>(pt, r, g, scale);
    softmax_p16n<NT>(p, r, g, scale);
    float dl = 0.f;                                           // delta_i = sum_j P[i][j] dP[i][j]  (masked entries have P = 0)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dl += pt[t][kk] * dpt[t][kk];
    dl += __shfl_xor(dl, 16);
    dl += __shfl_xor(dl, 32);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float dk = 0.f;                                       // the same delta for query 4g + kk, summed over the key lanes
#pragma unroll
        for (int t = 0; t < NT; ++t) dk += group_sum<16>(p[t][kk] * dp[t][kk]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            dst[t][kk] = pt[t][kk] * (dpt[t][kk] - dl) * scale;
            ds[t][kk] = p[t][kk] * (dp[t][kk] - dk) * scale;
        }
    }
}

template <typename EL>
__device__ __forceinline__ void attn32_bwd_body(const EL* __restrict__ qkv, const EL* __restrict__ dout, EL* __restrict__ dqkv,
                                                int d, int n_heads, int64_t n_items) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int64_t seq = item / n_heads;
    const int hh = (int)(item - seq * n_heads);
    const int r = lane & 15, g = lane >> 4;
    const int64_t ld = 3 * (int64_t)d;
    const EL* base = qkv + seq * 32 * ld + hh * HD;
    const EL* gbase = dout + seq * 32 * (int64_t)d + hh * HD;
    __shared__ __attribute__((aligned(16))) float turn_tile[4][16 * LDT];
    float* const T = turn_tile[threadIdx.x >> 6];
    float4 qf[2][4], kf[2][4], gf[2][4];
    f32x4 p0[1], ds0[1], dst0[1], p1[2], ds1[2], dst1[2];       // query block 0 (key tile 0), query block 1 (key tiles 0, 1)
    {
        float4 vf[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            load_rows16(qf[b], base + b * 16 * ld, ld, r, g);
            load_rows16(kf[b], base + b * 16 * ld + d, ld, r, g);
            load_rows16(vf[b], base + b * 16 * ld + 2 * d, ld, r, g);
            load_rows16(gf[b], gbase + b * 16 * (int64_t)d, (int64_t)d, r, g);
        }
        attn_block_grads<1>(qf[0], gf[0], kf, vf, r, g, 0.125f, p0, ds0, dst0);
        attn_block_grads<2>(qf[1], gf[1], kf, vf, r, g, 0.125f, p1, ds1, dst1);
    }
    EL* obase = dqkv + seq * 32 * ld + hh * HD;
    f32x4 a0[4], a1[4];
    float4 xk[4];
    auto turn = [&](const float4 (&rows)[4]) {                  // row-style registers -> frame-major, through the wave's LDS tile
#pragma unroll
        for (int f = 0; f < 4; ++f) st4(T + r * LDT + 16 * f + 4 * g, rows[f]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rho = 0; rho < 4; ++rho) xk[rho] = ld4(T + (4 * g + rho) * LDT + 4 * r);
        __builtin_amdgcn_wave_barrier();
    };
    auto zero = [&](f32x4 (&a)[4]) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) a[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // dV[j][c] = sum_i P[i][j] dO[i][c]:  keys 0..15 from both query blocks, keys 16..31 from the second
    zero(a0); zero(a1);
    turn(gf[0]); contract_frames16(a0, xk, p0[0]);
    turn(gf[1]); contract_frames16(a0, xk, p1[0]); contract_frames16(a1, xk, p1[1]);
    store_tiles16_rows(T, obase + 2 * d, ld, lane, a0);
    store_tiles16_rows(T, obase + 16 * ld + 2 * d, ld, lane, a1);
    // dK[j][c] = sum_i dS[i][j] Q[i][c]
    zero(a0); zero(a1);
    turn(qf[0]); contract_frames16(a0, xk, ds0[0]);
    turn(qf[1]); contract_frames16(a0, xk, ds1[0]); contract_frames16(a1, xk, ds1[1]);
    store_tiles16_rows(T, obase + d, ld, lane, a0);
    store_tiles16_rows(T, obase + 16 * ld + d, ld, lane, a1);
    // dQ[i][c] = sum_j dS^T[j][i] K[j][c]
    zero(a0); zero(a1);
    turn(kf[0]); contract_frames16(a0, xk, dst0[0]); contract_frames16(a1, xk, dst1[0]);
    turn(kf[1]); contract_frames16(a1, xk, dst1[1]);
    store_tiles16_rows(T, obase, ld, lane, a0);
    store_tiles16_rows(T, obase + 16 * ld, ld, lane, a1);
}
__global__ __launch_bounds__(256) void attn32_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                         float* __restrict__ dqkv, int d, int n_heads, int64_t n_items) {
    attn32_bwd_body(qkv, dout, dqkv, d, n_heads, n_items);
}
__global__ __launch_bounds__(256) void attn32_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                              bf16_t* __restrict__ dqkv, int d, int n_heads, int64_t n_items) {
    attn32_bwd_body(qkv, dout, dqkv, d, n_heads, n_items);
}

static int attn_check(const void* a, const void* b, int64_t n_seq, int T, int d) {
    if (n_seq < 1 || d < HD || (d % HD) != 0) return VLG_ERR_SHAPE;
    if (T != 4 && T != 8 && T != 16 && T != 32) return VLG_ERR_SHAPE;
    if (n_seq * (d / HD) > 0x7fffffff) return VLG_ERR_SHAPE;
    if (!vlg_aligned8(a) || !vlg_aligned8(b)) return VLG_ERR_ALIGN;
    return 0;
}

extern "C" int vlg_attention_fwd_bf16(const vlg_bf16* qkv_, vlg_bf16* o_, int64_t n_seq, int T, int d, void* stream) {
    if (int e = attn_check(qkv_, o_, n_seq, T, d)) return e;
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
    bf16_t* o = reinterpret_cast<bf16_t*>(o_);
    const int H = d / HD;
    const dim3 grid((unsigned)(n_seq * H)), block(64);
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 4:  hipLaunchKernelGGL((attn_fwd_kernel<4, bf16_t>),  grid, block, 0, s, qkv, o, d, H); break;
        case 8:  hipLaunchKernelGGL((attn_fwd_kernel<8, bf16_t>),  grid, block, 0, s, qkv, o, d, H); break;
        case 16: hipLaunchKernelGGL(attn16_fwd_bf16_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, o, d,
                                    H, n_seq * H); break;
        default: hipLaunchKernelGGL(attn32_fwd_bf16_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, o, d,
                                    H, n_seq * H); break;
    }
    return vlg_last_error();
}

extern "C" int vlg_attention_bwd_bf16(const vlg_bf16* qkv_, const vlg_bf16* dout_, vlg_bf16* dqkv_, int64_t n_seq, int T, int d,
                                      void* stream) {
    if (int e = attn_check(qkv_, dqkv_, n_seq, T, d)) return e;
    if (!vlg_aligned8(dout_)) return VLG_ERR_ALIGN;
    const bf16_t* qkv = reinterpret_cast<const bf16_t*>(qkv_);
    const bf16_t* dout = reinterpret_cast<const bf16_t*>(dout_);
    bf16_t* dqkv = reinterpret_cast<bf16_t*>(dqkv_);
    const int H = d / HD;
    const dim3 grid((unsigned)(n_seq * H)), block(64);
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 4:  hipLaunchKernelGGL((attn_bwd_kernel<4, bf16_t>),  grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        case 8:  hipLaunchKernelGGL((attn_bwd_kernel<8, bf16_t>),  grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        case 16: hipLaunchKernelGGL(attn16_bwd_bf16_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, dout,
                                    dqkv, d, H, n_seq * H); break;
        default: hipLaunchKernelGGL(attn32_bwd_bf16_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, dout,
                                    dqkv, d, H, n_seq * H); break;
    }
    return vlg_last_error();
}

extern "C" int vlg_attention_fwd(const float* qkv, float* o, int64_t n_seq, int T, int d, void* stream) {
    if (int e = attn_check(qkv, o, n_seq, T, d)) return e;
    if (!vlg_aligned16(qkv) || !vlg_aligned16(o)) return VLG_ERR_ALIGN;
    const int H = d / HD;
    const dim3 grid((unsigned)(n_seq * H)), block(64);
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 4:  hipLaunchKernelGGL(attn_fwd_kernel<4>,  grid, block, 0, s, qkv, o, d, H); break;
        case 8:  hipLaunchKernelGGL(attn_fwd_kernel<8>,  grid, block, 0, s, qkv, o, d, H); break;
        case 16: hipLaunchKernelGGL(attn16_fwd_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, o, d,
                                    H, n_seq * H); break;
        default: hipLaunchKernelGGL(attn32_fwd_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, o, d,
                                    H, n_seq * H); break;
    }
    return vlg_last_error();
}

extern "C" int vlg_attention_bwd(const float* qkv, const float* dout, float* dqkv, int64_t n_seq, int T, int d,
                                 void* stream) {
    if (int e = attn_check(qkv, dqkv, n_seq, T, d)) return e;
    if (!vlg_aligned16(dout) || !vlg_aligned16(qkv) || !vlg_aligned16(dqkv)) return VLG_ERR_ALIGN;
    const int H = d / HD;
    const dim3 grid((unsigned)(n_seq * H)), block(64);
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 4:  hipLaunchKernelGGL(attn_bwd_kernel<4>,  grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        case 8:  hipLaunchKernelGGL(attn_bwd_kernel<8>,  grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        case 16: hipLaunchKernelGGL(attn16_bwd_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, dout,
                                    dqkv, d, H, n_seq * H); break;
        default: hipLaunchKernelGGL(attn32_bwd_kernel, dim3((unsigned)((n_seq * H + 3) / 4)), dim3(256), 0, s, qkv, dout,
                                    dqkv, d, H, n_seq * H); break;
    }
    return vlg_last_error();
}
