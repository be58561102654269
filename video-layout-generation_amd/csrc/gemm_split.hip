// fp32 GEMMs on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 terms
//     x = x1 + x2 + x3,   x1 = top 8 significand bits, x2 = next 8, x3 = last 8   (truncating splits, so each
//                                                                                  remainder is exact in fp32)
// and a product block is six bf16 MFMAs with fp32 accumulation:
//     a.b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1          (dropped: a2b3 + a3b2 + a3b3 <= 3 * 2^-24 |a||b|)
// i.e. every product is reproduced to about one fp32 ulp, the sums are the MFMA's fp32 accumulation - the result is
// fp32-grade (measured against fp64 in tests/test_hip_ops.py next to the native v_mfma_f32_32x32x2_f32 path), while six
// v_mfma_f32_32x32x16_bf16 cost 6/16 of the fp32 MFMA cycles for the same k-depth.  Tensors stay fp32 in HBM.
//
// Structure = gemm.hip's (128x128 block tile, 2x2 wave tiles, mid-tile staging, XCD-aware tile order, slab outputs,
// fused epilogues); tiles = gemm_tile16.h's bf16 images, 16 contraction steps deep, three planes per operand
// (36 KB per stage, 72 KB double-buffered: 2 blocks / CU).  The split happens once per element on the way to LDS.
#include "common.h"
#include "gemm_tile16.h"

// three bf16 planes of 8 fp32 values: hi = x & 0xffff0000 (exactly a bf16), r = x - hi (exact), mid, lo likewise
__device__ __forceinline__ void split3(const float (&x)[8], uint4& p1, uint4& p2, uint4& p3) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned b1 = __float_as_uint(x[j]) & 0xffff0000u;
        const float r1 = x[j] - __uint_as_float(b1);
        const unsigned b2 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(b2);
        h[j] = b1; m[j] = b2; l[j] = __float_as_uint(r2);          // r2 has at most 8 significant bits: its low half is zero
    }
    // element 2q sits in the low half of dword q, element 2q+1 in the high half
#define PK(a, q) ((a[2 * (q)] >> 16) | (a[2 * (q) + 1] & 0xffff0000u))
    p1 = make_uint4(PK(h, 0), PK(h, 1), PK(h, 2), PK(h, 3));
    p2 = make_uint4(PK(m, 0), PK(m, 1), PK(m, 2), PK(m, 3));
    p3 = make_uint4(PK(l, 0), PK(l, 1), PK(l, 2), PK(l, 3));
#undef PK
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_split_kernel(const GemmArgs g) {
    constexpr int BKS = 16;
    const float* const gA = static_cast<const float*>(g.A);
    const float* const gB = static_cast<const float*>(g.B);
    const float* const gAuxIn = static_cast<const float*>(g.aux_in);
    float* const gAuxOut = static_cast<float*>(g.aux_out);
    constexpr int WM = (BM == 128 && BN == 128) ? 2 : (BM == 128 ? 4 : 1);
    constexpr int WN = 4 / WM;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    using TA = Tile16<BM, A_KC, BKS>;
    using TB = Tile16<BN, B_KC, BKS>;
    constexpr int STAGE = 3 * (TA::ELEMS + TB::ELEMS);                 // one buffer: A planes 1..3, then B planes 1..3
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * STAGE];

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    const int ntile = g.tiles_m * g.tiles_n;
    const int split = swz / ntile;
    const int tile = swz - split * ntile;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int64_t kbeg = (int64_t)split * g.kc_per_split;
    int64_t kend = kbeg + g.kc_per_split;
    if (kend > g.Kc) kend = g.Kc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] = 0.f;
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        bv[j] = 0.f;
        if constexpr ((EPI & VLG_EPI_BIAS) != 0) {
            const int col = n0 + (wn * TN + j) * 32 + l31;
            bv[j] = g.bias[col < g.N ? col : g.N - 1];
        }
    }

    const int nk = (int)((kend - kbeg + BKS - 1) / BKS);
    auto store3 = [&](const Stage16<float, TA::NV>& xa, const Stage16<float, TB::NV>& xb, int buf) {
        bf16_t* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            if (TA::SLOTS % GEMM_THREADS != 0 && idx >= TA::SLOTS) continue;
            const float x[8] = {xa.lo[i].x, xa.lo[i].y, xa.lo[i].z, xa.lo[i].w, xa.hi[i].x, xa.hi[i].y, xa.hi[i].z, xa.hi[i].w};
            uint4 p1, p2, p3;
            split3(x, p1, p2, p3);
            bf16_t* d = base + (idx / TA::PER_ROW) * TA::RS + ((idx % TA::PER_ROW) << 3);
            *reinterpret_cast<uint4*>(d) = p1;
            *reinterpret_cast<uint4*>(d + TA::ELEMS) = p2;
            *reinterpret_cast<uint4*>(d + 2 * TA::ELEMS) = p3;
        }
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            if (TB::SLOTS % GEMM_THREADS != 0 && idx >= TB::SLOTS) continue;
            const float x[8] = {xb.lo[i].x, xb.lo[i].y, xb.lo[i].z, xb.lo[i].w, xb.hi[i].x, xb.hi[i].y, xb.hi[i].z, xb.hi[i].w};
            uint4 p1, p2, p3;
            split3(x, p1, p2, p3);
            bf16_t* d = base + 3 * TA::ELEMS + (idx / TB::PER_ROW) * TB::RS + ((idx % TB::PER_ROW) << 3);
            *reinterpret_cast<uint4*>(d) = p1;
            *reinterpret_cast<uint4*>(d + TB::ELEMS) = p2;
            *reinterpret_cast<uint4*>(d + 2 * TB::ELEMS) = p3;
        }
    };
    const float cs_on = (COLSUM && tn == 0) ? 1.f : 0.f;              // bias gradient is taken by the tn == 0 column of blocks
    auto colsum = [&](const Stage16<float, TA::NV>& st) {               // branch-free: keeps the K loop one basic block
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            csum[0] = fmaf(cs_on, st.lo[i].x, csum[0]); csum[1] = fmaf(cs_on, st.lo[i].y, csum[1]);
            csum[2] = fmaf(cs_on, st.lo[i].z, csum[2]); csum[3] = fmaf(cs_on, st.lo[i].w, csum[3]);
            csum[4] = fmaf(cs_on, st.hi[i].x, csum[4]); csum[5] = fmaf(cs_on, st.hi[i].y, csum[5]);
            csum[6] = fmaf(cs_on, st.hi[i].z, csum[6]); csum[7] = fmaf(cs_on, st.hi[i].w, csum[7]);
        }
    };
    auto mainloop = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        Stage16<float, TA::NV> sa;
        Stage16<float, TB::NV> sb;
        auto load = [&](int t) {
            const int64_t k0 = kbeg + (int64_t)t * BKS;
            TA::template gload<GUARD>(sa, gA, g.lda, m0, g.M, k0, kend, tid);
            TB::template gload<GUARD>(sb, gB, g.ldb, n0, g.N, k0, kend, tid);
        };
        auto frags = [&](int kt, bf16x8 (&a)[TM][3], bf16x8 (&b)[TN][3]) {
            const bf16_t* as = smem + (kt & 1) * STAGE;
            const bf16_t* bs = as + 3 * TA::ELEMS;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[i][p] = TA::frag(as + p * TA::ELEMS, (wm * TM + i) * 32, 0, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = TB::frag(bs + p * TB::ELEMS, (wn * TN + j) * 32, 0, lane);
        };
        auto term = [&](const bf16x8 (&a)[TM][3], const bf16x8 (&b)[TN][3], int pa, int pb) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
        };
        if (nk > 0) {
            load(0);
            if constexpr (COLSUM) colsum(sa);
            store3(sa, sb, 0);
        }
        if (nk > 1) load(1);
        __syncthreads();
        // steady state: ONE basic block per tile - the split of tile kt+1 (about 5 vector instructions per MFMA) is
        // interleaved with the 24 MFMAs of tile kt instead of sitting between two MFMA bursts; tile kt+2 is requested
        // (index clamped, so the last request is a harmless repeat) once the registers are free
        for (int kt = 0; kt + 1 < nk; ++kt) {
            bf16x8 a[TM][3], b[TN][3];
            frags(kt, a, b);
            if constexpr (COLSUM) colsum(sa);
            term(a, b, 2, 0); term(a, b, 0, 2); term(a, b, 1, 1);      // smallest terms first, the leading term last
            store3(sa, sb, (kt & 1) ^ 1);
            load(kt + 2 < nk ? kt + 2 : nk - 1);
            term(a, b, 1, 0); term(a, b, 0, 1); term(a, b, 0, 0);
            __syncthreads();
        }
        if (nk > 0) {
            bf16x8 a[TM][3], b[TN][3];
            frags(nk - 1, a, b);
            term(a, b, 2, 0); term(a, b, 0, 2); term(a, b, 1, 1);
            term(a, b, 1, 0); term(a, b, 0, 1); term(a, b, 0, 0);
            __syncthreads();
        }
    };
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N) && (((kend - kbeg) % BKS) == 0);
    if (interior) mainloop(std::false_type{});
    else mainloop(std::true_type{});

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h)
    float* Cs = static_cast<float*>(g.C) + (int64_t)split * g.slab_stride;
    const int64_t row0 = m0 + wm * TM * 32 + 4 * h;
    const int col0 = n0 + wn * TN * 32 + l31;
    const int64_t base = row0 * g.ldc + col0;
    const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    auto emit = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (GUARD && col0 + j * 32 >= g.N) continue;
                float aux[16];
                if constexpr ((EPI & (VLG_EPI_RESID | VLG_EPI_DGELU)) != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                        aux[r] = (!GUARD || row0 + ro < g.M) ? gAuxIn[base + ro * g.ldc + j * 32] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (GUARD && row0 + ro >= g.M) continue;
                    const int64_t o = base + ro * g.ldc + j * 32;
                    float v = acc[i][j][r] + bv[j];
                    if constexpr ((EPI & VLG_EPI_GELU) != 0) { gAuxOut[o] = v; v = gelu_f(v); }
                    if constexpr ((EPI & VLG_EPI_RESID) != 0) v += aux[r];
                    if constexpr ((EPI & VLG_EPI_DGELU) != 0) v *= dgelu_f(aux[r]);
                    Cs[o] = v;
                }
            }
    };
    if (full) emit(std::false_type{});
    else emit(std::true_type{});
    if constexpr (COLSUM) {
        if (tn == 0) {
            float* red = reinterpret_cast<float*>(smem);
            constexpr int PR = TA::PER_ROW;
            if ((TA::SLOTS % GEMM_THREADS == 0) || tid < TA::SLOTS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(tid / PR) * BM + 8 * (tid % PR) + j] = csum[j];
            }
            __syncthreads();
            constexpr int NTL = (TA::SLOTS < GEMM_THREADS ? TA::SLOTS : GEMM_THREADS) / PR;
            if (tid < BM && m0 + tid < g.M) {
                float s = 0.f;
#pragma unroll 4
                for (int t = 0; t < NTL; ++t) s += red[t * BM + tid];
                Cs[g.colsum_off + m0 + tid] = s;
            }
        }
    }
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM>
static int launch_split(GemmArgs g, hipStream_t s) {
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (g.N + BN - 1) / BN;
    const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n * g.splits;
    if (blocks < 1 || blocks > 0x7fffffff) return VLG_ERR_SHAPE;
    g.clock_probe = nullptr;
    hipLaunchKernelGGL((gemm_split_kernel<BM, BN, A_KC, B_KC, EPI, COLSUM>), dim3((unsigned)blocks), dim3(GEMM_THREADS), 0, s, g);
    return vlg_last_error();
}

// Entry points for gemm.hip's C ABI functions (shapes validated there; this layer adds the 8-element slot granularity).
int vlg_gemm_split_fwd(GemmArgs g, int epilogue, hipStream_t s) {
    if (g.Kc & 7) return VLG_ERR_SHAPE;
    const bool narrow = g.N <= 32;
    if (epilogue == VLG_EPI_BIAS)
        return narrow ? launch_split<128, 32, true, true, VLG_EPI_BIAS, false>(g, s) : launch_split<128, 128, true, true, VLG_EPI_BIAS, false>(g, s);
    if (narrow) return VLG_ERR_SHAPE;
    if (epilogue == (VLG_EPI_BIAS | VLG_EPI_GELU)) return launch_split<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU, false>(g, s);
    if (epilogue == (VLG_EPI_BIAS | VLG_EPI_RESID)) return launch_split<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_RESID, false>(g, s);
    return VLG_ERR_SHAPE;
}
int vlg_gemm_split_dgrad(GemmArgs g, int epilogue, hipStream_t s) {
    if ((g.Kc & 7) || (g.N & 7)) return VLG_ERR_SHAPE;
    if (epilogue == VLG_EPI_NONE) return launch_split<128, 128, true, false, VLG_EPI_NONE, false>(g, s);
    if (epilogue == VLG_EPI_DGELU) return launch_split<128, 128, true, false, VLG_EPI_DGELU, false>(g, s);
    return VLG_ERR_SHAPE;
}
int vlg_gemm_split_wgrad(GemmArgs g, hipStream_t s) {
    if ((g.M & 7) || (g.N & 7)) return VLG_ERR_SHAPE;
    return g.M <= 32 ? launch_split<32, 128, false, false, VLG_EPI_NONE, true>(g, s)
                     : launch_split<128, 128, false, false, VLG_EPI_NONE, true>(g, s);
}
