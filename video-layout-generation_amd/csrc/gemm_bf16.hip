// bf16-MFMA GEMMs of the bf16 mode (BASELINE.json configs[2]): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Same three products as gemm.hip (forward A.W^T, data gradient dY.W, weight gradient dY^T.X split over tokens),
// same 128x128 block tile / 2x2 wave tiles / XCD-aware tile order / slab outputs, but the operand tiles live in LDS
// as bf16, 64 contraction steps deep, whatever the HBM element type (bf16 activations are copied, fp32 operands -
// weights, residual-stream gradients - are rounded once on their way to LDS):
//   KC image  [row][k]   rows of 64 bf16 + 8 pad = 36 dwords: the per-lane 16-byte fragment reads (ds_read_b128,
//                        lane = row, half-wave h picks k = 16s + 8h ..+7) hit 16 distinct 4-bank slots per 16 lanes
//   MC image  [k][row]   the operand is contraction-major in memory (W for the data gradient, dY and X for the weight
//                        gradient), so the tile is stored exactly as it arrives and read with gfx950's TRANSPOSING
//                        LDS read ds_read_b64_tr_b16: each 16-lane group fetches a 4(k) x 16(row) block and every
//                        lane receives the 4 k-values of ITS row - two reads make the 8-deep MFMA operand.  Row
//                        stride = BR*2 + 64 bytes, so the four k-rows of a block start 64 B apart modulo the 256-B
//                        bank window: conflict-free.
// Both images give lane (row = l & 31, h = l >> 5) the k-values 16s + 8h + j, j = 0..7, so A and B agree on the
// contraction order whatever their layouts.
// Algorithmic work per launch: 2*M*N*K flop; bytes = element sizes x (M*K + N*K + M*N) (+ aux operands): with
// bf16 activations the 128 x 128 x 64 tile needs 16 MFMAs (512 cycles) per 32 KB staged - the kernel is priced
// against HBM, not MFMA.
#include "common.h"
#include <type_traits>
#include "gemm_tile16.h"
#include "reduce_body.h"
#include <stdlib.h>

// LDS elements (bf16) of one block: both operand tiles, double-buffered
template <int BM, int BN, int BK16, bool A_KC, bool B_KC>
constexpr int gemm16_smem_elems() { return 2 * (Tile16<BM, A_KC, BK16>::ELEMS + Tile16<BN, B_KC, BK16>::ELEMS); }

// the kernel body as a function of (arguments, LDS, block number, block count): one problem per launch (gemm_bf16_kernel) or
// a projection's data gradient and weight gradient side by side (gemm16_pair_kernel), as in gemm.hip
template <int BM, int BN, int BK16, bool A_KC, bool B_KC, int EPI, bool COLSUM, typename EA, typename EB, typename EO>
__device__ __forceinline__ void gemm16_body(const GemmArgs& g, bf16_t* const smem, const int bid, const int nwg) {
    constexpr int DEPTH = BK16 == 64 ? 2 : 1;          // register-resident tiles in flight besides the two LDS buffers
    // (a 32-deep, DEPTH 1, 4-blocks-per-CU instantiation was measured: +4 % on forward / data gradient, -30 % on the
    // weight gradient whose column-sum registers then spill - not kept)
    constexpr int NS = BK16 / 16;                      // 16-deep MFMA steps per tile
    static_assert(!COLSUM || !A_KC, "column sums are taken from a contraction-major A tile");
    const EA* const gA = static_cast<const EA*>(g.A);
    const EB* const gB = static_cast<const EB*>(g.B);
    const EO* const gAuxIn = static_cast<const EO*>(g.aux_in);
    EO* const gAuxOut = static_cast<EO*>(g.aux_out);
    constexpr int WM = (BM == 128 && BN == 128) ? 2 : (BM == 128 ? 4 : 1);
    constexpr int WN = 4 / WM;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    using TA = Tile16<BM, A_KC, BK16>;
    using TB = Tile16<BN, B_KC, BK16>;
    bf16_t* const As0 = smem;
    bf16_t* const Bs0 = smem + 2 * TA::ELEMS;

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

    const int nk = (int)((kend - kbeg + BK16 - 1) / BK16);
    auto kstep = [&](const bf16_t* as, const bf16_t* bs, int s) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = TA::frag(as, (wm * TM + i) * 32, s, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = TB::frag(bs, (wn * TN + j) * 32, s, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    auto mainloop = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        Stage16<EA, TA::NV> sa0, sa1;
        Stage16<EB, TB::NV> sb0, sb1;
        auto load = [&](Stage16<EA, TA::NV>& xa, Stage16<EB, TB::NV>& xb, int t) {
            const int64_t k0 = kbeg + (int64_t)t * BK16;
            TA::template gload<GUARD>(xa, gA, g.lda, m0, g.M, k0, kend, tid);
            TB::template gload<GUARD>(xb, gB, g.ldb, n0, g.N, k0, kend, tid);
        };
        // one iteration: the MFMAs of the tile in LDS buffer kt&1; in the middle, tile kt+1 (in registers) goes to the other
        // buffer and its registers are refilled with tile kt+3 - two tiles are in flight between HBM and LDS at any time
        auto iter = [&](int kt, Stage16<EA, TA::NV>& xa, Stage16<EB, TB::NV>& xb) {
            const int cur = kt & 1;
            const bf16_t* as = As0 + cur * TA::ELEMS;
            const bf16_t* bs = Bs0 + cur * TB::ELEMS;
#pragma unroll
            for (int ks = 0; ks < NS / 2; ++ks) kstep(as, bs, ks);
            if (kt + 1 < nk) {
                if constexpr (COLSUM) { if (tn == 0) TA::colsum_add(csum, xa); }
                TA::sstore(xa, As0 + (cur ^ 1) * TA::ELEMS, tid);
                TB::sstore(xb, Bs0 + (cur ^ 1) * TB::ELEMS, tid);
            }
            if (kt + 1 + DEPTH < nk) load(xa, xb, kt + 1 + DEPTH);
#pragma unroll
            for (int ks = NS / 2; ks < NS; ++ks) kstep(as, bs, ks);
            __syncthreads();
        };
        if (nk > 0) {
            load(sa0, sb0, 0);
            if constexpr (COLSUM) { if (tn == 0) TA::colsum_add(csum, sa0); }
            TA::sstore(sa0, As0, tid);
            TB::sstore(sb0, Bs0, tid);
        }
        if (nk > 1) load(sa0, sb0, 1);
        if constexpr (DEPTH == 2) {
            if (nk > 2) load(sa1, sb1, 2);
        }
        __syncthreads();
        if constexpr (DEPTH == 2) {
            for (int kt = 0; kt < nk; kt += 2) {        // tile kt+1 sits in (sa0, sb0) on even, in (sa1, sb1) on odd iterations
                iter(kt, sa0, sb0);
                if (kt + 1 < nk) iter(kt + 1, sa1, sb1);
            }
        } else {
            for (int kt = 0; kt < nk; ++kt) iter(kt, sa0, sb0);
        }
    };
    const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N) && (((kend - kbeg) % BK16) == 0);
    if (interior) mainloop(std::false_type{});
    else mainloop(std::true_type{});

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h)
    EO* Cs = static_cast<EO*>(g.C) + (int64_t)split * g.slab_stride;
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
                if constexpr ((EPI & (VLG_EPI_RESID | VLG_EPI_DGELU | VLG_EPI_MUL)) != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                        aux[r] = (!GUARD || row0 + ro < g.M) ? ld1(gAuxIn + base + ro * g.ldc + j * 32) : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (GUARD && row0 + ro >= g.M) continue;
                    const int64_t o = base + ro * g.ldc + j * 32;
                    float v = acc[i][j][r] + bv[j];
                    if constexpr ((EPI & VLG_EPI_GELU) != 0) {
                        if constexpr ((EPI & VLG_EPI_GELU_GRAD) != 0) {      // aux_out = gelu'(pre) (common.h, gelu_parts): the backward
                            float c, pd;                                     // pass then multiplies (VLG_EPI_MUL) instead of recomputing
                            gelu_parts(v, c, pd);
                            st1(gAuxOut + o, c + v * pd);
                            v *= c;
                        } else {
                            st1(gAuxOut + o, v);
                            v = gelu_f(v);
                        }
                    }
                    if constexpr ((EPI & VLG_EPI_RESID) != 0) v += aux[r];
                    if constexpr ((EPI & VLG_EPI_DGELU) != 0) v *= dgelu_f(aux[r]);
                    if constexpr ((EPI & VLG_EPI_MUL) != 0) v *= aux[r];
                    st1(Cs + o, v);
                }
            }
    };
    // (An LDS-staged epilogue for bf16 outputs - column pairs packed with a DPP swap, 16-byte row-contiguous stores - was
    // measured and is NOT faster than these 2-byte stores: the kernels are not bound by store requests.)
    if (full) emit(std::false_type{});
    else emit(std::true_type{});
    if constexpr (COLSUM) {
        // every thread summed the 8 rows 8*(tid % PER_ROW) .. +7 of the tiles it staged: combine the GEMM_THREADS / PER_ROW
        // threads that share a row group through LDS (the tiles are dead: the main loop ended on a barrier)
        if (tn == 0) {
            float* red = reinterpret_cast<float*>(smem);
            constexpr int PR = TA::PER_ROW, NT = GEMM_THREADS / PR;         // threads per row group
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
                st1(Cs + g.colsum_off + m0 + tid, s);
            }
            (void)NT;
        }
    }
}

// IO bit 0: A is bf16, bit 1: B is bf16, bit 2: C and the epilogue's auxiliary operands are bf16 (else float)
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, int IO>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_bf16_kernel(const GemmArgs g) {
    using EA = std::conditional_t<(IO & 1) != 0, bf16_t, float>;
    using EB = std::conditional_t<(IO & 2) != 0, bf16_t, float>;
    using EO = std::conditional_t<(IO & 4) != 0, bf16_t, float>;
    __shared__ __attribute__((aligned(16))) bf16_t smem[gemm16_smem_elems<BM, BN, 64, A_KC, B_KC>()];
    gemm16_body<BM, BN, 64, A_KC, B_KC, EPI, COLSUM, EA, EB, EO>(g, smem, (int)blockIdx.x, (int)gridDim.x);
}

// A projection's data gradient (blocks [0, nd_pad): nd real, padded to a multiple of 8 for the XCD mapping of the second
// problem) and weight gradient (the rest) in ONE launch - csrc/gemm.hip, gemm_pair_kernel.  The bf16-storage step's
// combinations: dY bf16 or fp32 (A16), W / X bf16, dX (+ the multiply epilogue's operand) bf16.
// (blocks past both problems: rider rows of a slab-reduction table, as in gemm_pair_kernel)
template <int EPI_D, bool A16>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm16_pair_kernel(const GemmArgs gd, const GemmArgs gw, const int nd, const int nd_pad,
                                                                     const int nw, const int64_t* __restrict__ rider, const int rider_bpr) {
    using EA = std::conditional_t<A16, bf16_t, float>;
    constexpr int FD = gemm16_smem_elems<128, 128, 64, true, false>(), FW = gemm16_smem_elems<128, 128, 64, false, false>();
    __shared__ __attribute__((aligned(16))) bf16_t smem[FD > FW ? FD : FW];
    const int b = (int)blockIdx.x;
    if (b < nd_pad) {
        if (b < nd) gemm16_body<128, 128, 64, true, false, EPI_D, false, EA, bf16_t, bf16_t>(gd, smem, b, nd);
    } else if (b < nd_pad + nw) {
        gemm16_body<128, 128, 64, false, false, VLG_EPI_NONE, true, EA, bf16_t, float>(gw, smem, b - nd_pad, nw);
    } else {
        const int rb = b - nd_pad - nw;
        reduce_table_row(rider, rb / rider_bpr, rb % rider_bpr, rider_bpr, reinterpret_cast<float4(*)[16]>(smem));
    }
}
template <int EPI_D, bool A16>
static int launch16_pair(GemmArgs gd, GemmArgs gw, const int64_t* rider, int rider_rows, int rider_bpr, hipStream_t s) {
    gd.tiles_m = (int)((gd.M + 127) / 128); gd.tiles_n = (gd.N + 127) / 128; gd.clock_probe = nullptr; gd.run = 1;
    gw.tiles_m = (int)((gw.M + 127) / 128); gw.tiles_n = (gw.N + 127) / 128; gw.clock_probe = nullptr; gw.run = 1;
    const int64_t nd = (int64_t)gd.tiles_m * gd.tiles_n, nw = (int64_t)gw.tiles_m * gw.tiles_n * gw.splits;
    const int64_t nd_pad = (nd + 7) / 8 * 8;
    const int64_t nr = rider ? (int64_t)rider_rows * rider_bpr : 0;
    if (nd < 1 || nw < 1 || nd_pad + nw + nr > 0x7fffffff) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL((gemm16_pair_kernel<EPI_D, A16>), dim3((unsigned)(nd_pad + nw + nr)), dim3(GEMM_THREADS), 0, s, gd, gw, (int)nd, (int)nd_pad,
                       (int)nw, rider, rider_bpr);
    return vlg_last_error();
}
// bf16-storage pair (W / X / dX bf16): dy_bf16 = the shared dY is bf16 (else fp32: a residual-stream gradient)
int vlg_gemm16_pair(GemmArgs gd, GemmArgs gw, int epilogue, bool dy_bf16, const int64_t* rider, int rider_rows, int rider_bpr, hipStream_t s) {
    if ((gd.ldc & 7) || !vlg_aligned16(gd.C) || (gd.Kc & 7) || (gd.N & 7) || (gw.M & 7) || (gw.N & 7) || gw.M <= 32) return VLG_ERR_SHAPE;
#define PAIR16(EPI, A16) launch16_pair<EPI, A16>(gd, gw, rider, rider_rows, rider_bpr, s)
    if (epilogue == VLG_EPI_NONE) return dy_bf16 ? PAIR16(VLG_EPI_NONE, true) : PAIR16(VLG_EPI_NONE, false);
    if (epilogue == VLG_EPI_MUL) return dy_bf16 ? PAIR16(VLG_EPI_MUL, true) : PAIR16(VLG_EPI_MUL, false);
#undef PAIR16
    return VLG_ERR_SHAPE;
}
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM, int IO>
static int launch16(GemmArgs g, hipStream_t s) {
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (g.N + BN - 1) / BN;
    const int64_t blocks = (int64_t)g.tiles_m * g.tiles_n * g.splits;
    if (blocks < 1 || blocks > 0x7fffffff) return VLG_ERR_SHAPE;
    g.clock_probe = nullptr;
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, A_KC, B_KC, EPI, COLSUM, IO>), dim3((unsigned)blocks), dim3(GEMM_THREADS), 0, s, g);
    return vlg_last_error();
}

// Entry points for gemm.hip's C ABI functions.  `io` = storage bits (see gemm_bf16_kernel); shapes were validated by the
// caller, this layer adds the 8-element granularity of the bf16 slots.  Instantiated: everything fp32 (io 0, the
// "bf16_mfma" mode) and the combinations the bf16-storage step launches (bf16 activations AND bf16 shadow weights).
int vlg_gemm16_fwd(GemmArgs g, int epilogue, int io, hipStream_t s) {
    if ((io & 4) && ((g.ldc & 7) || !vlg_aligned16(g.C) || (g.aux_out && !vlg_aligned16(g.aux_out)))) return VLG_ERR_ALIGN;
    if (g.Kc & 7) return VLG_ERR_SHAPE;
    const bool narrow = g.N <= 32;
#define FWD(EPI, IO) (narrow ? launch16<128, 32, true, true, EPI, false, IO>(g, s) : launch16<128, 128, true, true, EPI, false, IO>(g, s))
    if (epilogue == VLG_EPI_BIAS) {
        switch (io) { case 0: return FWD(VLG_EPI_BIAS, 0); case 3: return FWD(VLG_EPI_BIAS, 3); case 7: return FWD(VLG_EPI_BIAS, 7); default: return VLG_ERR_SHAPE; }
    }
#undef FWD
    if (narrow) return VLG_ERR_SHAPE;
    if (epilogue == (VLG_EPI_BIAS | VLG_EPI_GELU)) {
        if (io == 0) return launch16<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU, false, 0>(g, s);
        if (io == 7) return launch16<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU, false, 7>(g, s);
        return VLG_ERR_SHAPE;
    }
    if (epilogue == (VLG_EPI_BIAS | VLG_EPI_GELU | VLG_EPI_GELU_GRAD)) {
        if (io == 0) return launch16<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU | VLG_EPI_GELU_GRAD, false, 0>(g, s);
        if (io == 7) return launch16<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU | VLG_EPI_GELU_GRAD, false, 7>(g, s);
        return VLG_ERR_SHAPE;
    }
    if (epilogue == (VLG_EPI_BIAS | VLG_EPI_RESID)) {
        if (io == 0) return launch16<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_RESID, false, 0>(g, s);
        if (io == 3) return launch16<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_RESID, false, 3>(g, s);
        return VLG_ERR_SHAPE;
    }
    return VLG_ERR_SHAPE;
}

int vlg_gemm16_dgrad(GemmArgs g, int epilogue, int io, hipStream_t s) {
    // contraction over N (g.Kc), W rows are K-contiguous: both extents in 8-element slots
    if ((io & 4) && ((g.ldc & 7) || !vlg_aligned16(g.C))) return VLG_ERR_ALIGN;
    if ((g.Kc & 7) || (g.N & 7)) return VLG_ERR_SHAPE;
    if (epilogue == VLG_EPI_NONE) {
        switch (io) {
            case 0: return launch16<128, 128, true, false, VLG_EPI_NONE, false, 0>(g, s);
            case 6: return launch16<128, 128, true, false, VLG_EPI_NONE, false, 6>(g, s);
            case 7: return launch16<128, 128, true, false, VLG_EPI_NONE, false, 7>(g, s);
            default: return VLG_ERR_SHAPE;
        }
    }
    if (epilogue == VLG_EPI_DGELU) {
        if (io == 0) return launch16<128, 128, true, false, VLG_EPI_DGELU, false, 0>(g, s);
        if (io == 6) return launch16<128, 128, true, false, VLG_EPI_DGELU, false, 6>(g, s);
        return VLG_ERR_SHAPE;
    }
    if (epilogue == VLG_EPI_MUL) {
        if (io == 0) return launch16<128, 128, true, false, VLG_EPI_MUL, false, 0>(g, s);
        if (io == 6) return launch16<128, 128, true, false, VLG_EPI_MUL, false, 6>(g, s);
        return VLG_ERR_SHAPE;
    }
    return VLG_ERR_SHAPE;
}

int vlg_gemm16_wgrad(GemmArgs g, int io, hipStream_t s) {
    // A = dY [tokens][N] and B = X [tokens][K] are both contraction-major: row extents in 8-element slots
    if ((g.M & 7) || (g.N & 7)) return VLG_ERR_SHAPE;
    const bool narrow = g.M <= 32;
#define WG(IO) (narrow ? launch16<32, 128, false, false, VLG_EPI_NONE, true, IO>(g, s) : launch16<128, 128, false, false, VLG_EPI_NONE, true, IO>(g, s))
    switch (io) { case 0: return WG(0); case 2: return WG(2); case 3: return WG(3); default: return VLG_ERR_SHAPE; }
#undef WG
}
