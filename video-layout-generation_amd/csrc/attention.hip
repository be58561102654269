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
template <int T>
__device__ __forceinline__ void stage_tile(float* S, const float* __restrict__ src, int64_t ld, int lane) {
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

template <int T>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                      int d, int n_heads) {
    using G = AttnGeom<T>;
    __shared__ __attribute__((aligned(16))) float sm[3 * T * LDT + T * (T + 1)];
    float* Qs = sm; float* Ks = sm + T * LDT; float* Vs = sm + 2 * T * LDT; float* Ps = sm + 3 * T * LDT;
    const int lane = threadIdx.x;
    const int64_t seq = blockIdx.x / n_heads;
    const int hh = blockIdx.x - (int)(seq * n_heads);
    const int64_t ld = 3 * (int64_t)d;
    const float* base = qkv + seq * T * ld + hh * HD;
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
        float* dst = o + (seq * T + i) * (int64_t)d + hh * HD + jg * G::CPL;
#pragma unroll
        for (int c = 0; c < G::CPL / 4; ++c) st4(dst + c * 4, acc[c]);
    }
}

template <int T>
__global__ __launch_bounds__(64) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                      float* __restrict__ dqkv, int d, int n_heads) {
    using G = AttnGeom<T>;
    __shared__ __attribute__((aligned(16))) float sm[4 * T * LDT + 2 * T * (T + 1)];
    float* Qs = sm; float* Ks = sm + T * LDT; float* Vs = sm + 2 * T * LDT; float* Os = sm + 3 * T * LDT;
    float* Ps = sm + 4 * T * LDT; float* Ds = Ps + T * (T + 1);
    const int lane = threadIdx.x;
    const int64_t seq = blockIdx.x / n_heads;
    const int hh = blockIdx.x - (int)(seq * n_heads);
    const int64_t ld = 3 * (int64_t)d;
    const float* base = qkv + seq * T * ld + hh * HD;
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
        float* dst = dqkv + (seq * T + r) * ld + hh * HD + c0;
#pragma unroll
        for (int c = 0; c < G::CPL / 4; ++c) {
            st4(dst + c * 4, aq[c]);
            st4(dst + d + c * 4, ak[c]);
            st4(dst + 2 * d + c * 4, av[c]);
        }
    }
}

static int attn_check(const void* a, const void* b, int64_t n_seq, int T, int d) {
    if (n_seq < 1 || d < HD || (d % HD) != 0) return VLG_ERR_SHAPE;
    if (T != 4 && T != 8 && T != 16 && T != 32) return VLG_ERR_SHAPE;
    if (n_seq * (d / HD) > 0x7fffffff) return VLG_ERR_SHAPE;
    if (!vlg_aligned16(a) || !vlg_aligned16(b)) return VLG_ERR_ALIGN;
    return 0;
}

extern "C" int vlg_attention_fwd(const float* qkv, float* o, int64_t n_seq, int T, int d, void* stream) {
    if (int e = attn_check(qkv, o, n_seq, T, d)) return e;
    const int H = d / HD;
    const dim3 grid((unsigned)(n_seq * H)), block(64);
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 4:  hipLaunchKernelGGL(attn_fwd_kernel<4>,  grid, block, 0, s, qkv, o, d, H); break;
        case 8:  hipLaunchKernelGGL(attn_fwd_kernel<8>,  grid, block, 0, s, qkv, o, d, H); break;
        case 16: hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, block, 0, s, qkv, o, d, H); break;
        default: hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, block, 0, s, qkv, o, d, H); break;
    }
    return vlg_last_error();
}

extern "C" int vlg_attention_bwd(const float* qkv, const float* dout, float* dqkv, int64_t n_seq, int T, int d,
                                 void* stream) {
    if (int e = attn_check(qkv, dqkv, n_seq, T, d)) return e;
    if (!vlg_aligned16(dout)) return VLG_ERR_ALIGN;
    const int H = d / HD;
    const dim3 grid((unsigned)(n_seq * H)), block(64);
    hipStream_t s = (hipStream_t)stream;
    switch (T) {
        case 4:  hipLaunchKernelGGL(attn_bwd_kernel<4>,  grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        case 8:  hipLaunchKernelGGL(attn_bwd_kernel<8>,  grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        case 16: hipLaunchKernelGGL(attn_bwd_kernel<16>, grid, block, 0, s, qkv, dout, dqkv, d, H); break;
        default: hipLaunchKernelGGL(attn_bwd_kernel<32>, grid, block, 0, s, qkv, dout, dqkv, d, H); break;
    }
    return vlg_last_error();
}
