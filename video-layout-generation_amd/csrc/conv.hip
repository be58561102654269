// 3x3 convolutions of the reference's GridNet (reference src/models/modules.py:5-58) as fp32 MFMA
// implicit GEMMs - forward, data gradient and weight gradient - on halo-padded channels-last tensors.
//
// Activation format ("padded NHWC"): a (b, C, H, W) tensor is stored as rows p = (n*(H+2) + y')*(W+2) + x'
// of Cp floats (C padded to a multiple of 32, extra channels zero), with a ZERO one-pixel halo and a zero
// guard band of rows before and after.  Then tap (ky,kx) of a stride-1 3x3 convolution is just a constant
// ROW SHIFT (ky-1)*(W+2) + (kx-1) of the same matrix:
//     out[p, co] = mask[p] * ( bias[co] + sum_tap sum_ci act(in[p + shift(tap), ci]) * W[co][tap][ci] )
// i.e. a GEMM with M = rows, N = Cout, K = 9*Cp whose A-operand base pointer moves once per K tile (Cp is
// a multiple of BK = 32, so a K tile never straddles taps).  No im2col buffer, no boundary branches: the
// halo supplies the zero padding and the epilogue multiplies halo rows by mask = 0 so the halo stays zero.
// Stride-2 (down-sampling) convolutions use one int row table (output row -> input row of the window
// centre); their data gradient uses nine per-tap tables that point at a guard (zero) row where a tap does
// not exist.  Weights live as [Cout][9][Cp] so forward reads them K-contiguous, the data gradient reads
// them contraction-major with a per-tap column offset, and the weight gradient writes that layout directly.
// PReLU (single shared slope, nn.PReLU() default) is applied to the gathered operand on its way to LDS
// (forward: A; weight gradient: B); its derivative and the slope gradient are fused into the data
// gradient's epilogue.  The MFMA tile machinery (fragment layouts, LDS staging, mid-tile pipeline) is the
// one of gemm.hip (gemm_tile.h).
#include "gemm_tile.h"
#include <type_traits>
#include <stdlib.h>

#define CONV_FWD 0
#define CONV_DGRAD 1
#define CONV_WGRAD 2

struct ConvArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* aux_in; const float* rowmask; const float* prelu; float* da_slab;
    const int* rowtab; int64_t tab_stride;
    int64_t M;               // rows of C
    int N;                   // cols of C that are stored
    int64_t Kc;              // contraction extent
    int lda, ldb, ldc;
    int cin;                 // channels per tap of the gathered operand (multiple of 32)
    int b_tap_stride;        // dgrad: column offset of one tap inside a weight row
    int shift[9];
    int wp, sign;            // shift[tap] = sign * ((tap / 3 - 1) * wp + tap % 3 - 1) (0 / 0 with per-tap row tables)
    int tiles_m, tiles_n, splits;
    int64_t kc_per_split, slab_stride, colsum_off;
    int epi;
    int act_ch;              // PReLU applies to channels < act_ch only (AddCoords channels stay linear)
    // tail of the tile order cut along K ("data-parallel + split-K remainder"): the last tail_tiles tiles (whole row tiles)
    // are computed by tail_splits blocks each, which store raw partial tiles to tail_ws[split][row - tail_row0][ldc];
    // conv_finish_* sums them and applies the epilogue.  0 = every tile by one block.
    int tail_tiles, tail_splits;
    int64_t tail_kc, tail_row0, tail_stride;
    float* tail_ws;
    unsigned long long* probe;   // diagnostic build only (-DVLG_TIMELINE, tools/diag/conv_timeline.py)
};

__device__ __forceinline__ float prelu_f(float v, float a) { return v > 0.f ? v : a * v; }
// branch-free and exact: max(v,0) + a*min(v,0) is v for v > 0 and the singly-rounded a*v otherwise; a = 1 is the identity,
// a = 0 is ReLU.  med3 keeps the compiler from inserting NaN-canonicalising moves around max/min.
__device__ __forceinline__ float act_f(float v, float a) {
    return __builtin_fmaf(a, __builtin_amdgcn_fmed3f(v, -__builtin_inff(), 0.f), __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()));
}
__device__ __forceinline__ float4 act4(float4 v, float4 a) { return make_float4(act_f(v.x, a.x), act_f(v.y, a.y), act_f(v.z, a.z), act_f(v.w, a.w)); }
// slopes of channels c .. c+3: channels >= act_ch (the AddCoords pair, padding) stay linear
__device__ __forceinline__ float4 slope4(float a, int c, int act_ch) {
    return make_float4(c < act_ch ? a : 1.f, c + 1 < act_ch ? a : 1.f, c + 2 < act_ch ? a : 1.f, c + 3 < act_ch ? a : 1.f);
}

template <int MODE, int BM, int BN, int BK>
__global__ __launch_bounds__(GEMM_THREADS, 2) void conv_gemm_kernel(const ConvArgs g) {
    constexpr int NCH = BK / 8;                            // 8-deep MFMA chunks per K tile
    constexpr int KPR = BK / 4;                            // float4 per K-contiguous tile row
    constexpr bool A_KC = MODE != CONV_WGRAD;
    constexpr bool B_KC = MODE == CONV_FWD;
    // weight gradient: the four waves split the 128 columns, each takes every row tile (BM = Cout = 32, 64 or 96)
    constexpr int WM = MODE == CONV_WGRAD ? 1 : (BM == 32) ? 1 : (BM == 64 ? 2 : ((BN == 128 || BN == 64) ? 2 : 4));
    constexpr int WN = 4 / WM;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    static_assert(TM >= 1 && TN >= 1 && TM * 32 * WM == BM && TN * 32 * WN == BN, "tile / wave layout");
    using TA = Tile<BM, A_KC, BK>;
    using TB = Tile<BN, B_KC, BK>;
    // (+ a 16-byte dump slot per thread where a tile has fewer float4 than the block has threads: the fast path's surplus
    // threads write there instead of branching around the LDS write)
    constexpr int DUMP = (TA::F4 % GEMM_THREADS != 0 || TB::F4 % GEMM_THREADS != 0) ? GEMM_THREADS * 4 : 0;
    __shared__ __attribute__((aligned(16))) float smem[2 * (TA::FLOATS + TB::FLOATS) + DUMP];
    __shared__ float red[GEMM_THREADS / 64];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * TA::FLOATS;

    const int bid = blockIdx.x;
#ifdef VLG_TIMELINE
    const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    unsigned long long tl_loop0 = 0, tl_loop1 = 0, tl_vm = 0, tl_bar = 0, tl_c0 = 0, tl_c1 = 0;
#endif
    const int ntile = g.tiles_m * g.tiles_n;
    // The K ranges of the tail tiles come FIRST in launch order (padded to a multiple of 8 blocks so that the whole tiles
    // behind them keep block % 8 = XCD): an empty chip deals them one per CU, next to a whole tile each.  Put last, they
    // would go two at a time to the first CUs that drain (measured: 136 short blocks cost 25 us instead of ~12).
    const int ntail = g.tail_tiles * g.tail_splits, ntp = (ntail + 7) & ~7;
    const int nmain = g.tail_tiles > 0 ? ntile - g.tail_tiles : (int)gridDim.x;
    const bool part = bid < ntp;
    if (part && bid >= ntail) return;                          // padding
    int split, tile;
    if (!part) {
        const int j = bid - ntp;
        const int q = nmain >> 3, rr = nmain & 7, xcd = j & 7;
        const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (j >> 3);
        split = swz / ntile;
        tile = swz - split * ntile;
    } else {
        tile = nmain + bid / g.tail_splits;
        split = bid - (tile - nmain) * g.tail_splits;
    }
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int64_t kper = part ? g.tail_kc : g.kc_per_split;
    const int64_t kbeg = (int64_t)split * kper;
    int64_t kend = kbeg + kper;
    if (kend > g.Kc) kend = g.Kc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
    const float slope = g.prelu ? g.prelu[0] : 1.0f;          // slope 1 == identity

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float colacc = 0.f;
    float bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + l31;
        bv[j] = (MODE == CONV_FWD && g.bias != nullptr && !part) ? g.bias[col < g.N ? col : g.N - 1] : 0.f;
    }

    // which main loop this launch takes (block-uniform; decided first so that the general loop's per-thread gather
    // constants - 64-bit pointer arithmetic, ~150 vector instructions - are not computed by blocks that never use them:
    // a 128 x 32 tile with K = 288 has only 144 MFMAs per wave to hide them behind)
    bool fast = false;
    if constexpr (MODE != CONV_WGRAD && BM >= 64) {
        const int pad = g.wp + 1;
        const int64_t a_bytes = (g.M + 2 * (int64_t)pad) * g.lda * 4;
        const int64_t b_bytes = (MODE == CONV_FWD ? (int64_t)g.N : (int64_t)g.cin) * g.ldb * 4;
        fast = g.rowtab == nullptr && g.wp > 0 && (g.N % BN) == 0 && a_bytes + (int64_t)BM * g.lda * 4 < (1ll << 31) && b_bytes < (1ll << 31) &&
               (MODE != CONV_FWD || g.act_ch >= g.cin) && ((kend - kbeg) % BK) == 0;
    }
    if constexpr (MODE == CONV_WGRAD) {
        const int pad = g.wp + 1;
        const int64_t a_bytes = g.Kc * (int64_t)g.lda * 4, b_bytes = (g.Kc + 2 * (int64_t)pad) * g.ldb * 4;
        fast = g.rowtab == nullptr && g.wp > 0 && g.act_ch >= g.cin && a_bytes < (1ll << 31) && b_bytes < (1ll << 31) &&
               (int64_t)g.M * g.ldc * 4 < (1ll << 31) && (kbeg % BK) == 0;
    }

    // ---- per-thread constants of the gathers.  Everything that does not change from K tile to K tile is folded
    // into per-thread base pointers here; a K tile then adds ONE wave-uniform offset (tap shift and channel block),
    // which the scalar unit tracks with two running counters instead of a division per tile.
    const float* pa[TA::NV];                                   // forward / data gradient: rows of A (fixed across K tiles)
    const float* pb[TB::NV];
    int arow[TA::NV];                                          // data gradient of a stride-2 conv: row -> per-tap table index
    if (MODE != CONV_WGRAD && !fast) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            int64_t r = m0 + ((tid + GEMM_THREADS * i) / KPR);
            r = r < g.M ? r : g.M - 1;                         // rows past M are computed but never stored
            arow[i] = (int)r;
            if (g.rowtab != nullptr && g.tab_stride == 0) r = g.rowtab[r];
            pa[i] = g.A + r * g.lda + (((tid + GEMM_THREADS * i) % KPR) << 2);
        }
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            if constexpr (MODE == CONV_FWD) {
                int row = n0 + idx / KPR;
                row = row < g.N ? row : g.N - 1;
                pb[i] = g.B + (int64_t)row * g.ldb + ((idx % KPR) << 2);
            } else {
                pb[i] = g.B + (int64_t)(idx / (BN / 4)) * g.ldb + n0 + ((idx % (BN / 4)) << 2);
            }
        }
    }
    // weight gradient: the (tap, channel) of every B column this thread stages
    int64_t bcoloff[TB::NV];
    int bch[TB::NV];
    float4 bslope[TB::NV];                                     // slopes of this thread's four channels (fixed across K tiles)
    if (MODE == CONV_WGRAD && !fast) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            pa[i] = g.A + (int64_t)(idx / (BM / 4)) * g.lda + m0 + ((idx % (BM / 4)) << 2);
        }
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            int col = n0 + (((tid + GEMM_THREADS * i) % (BN / 4)) << 2);
            if (col > g.N - 4) col = g.N - 4;                  // columns past N are computed but never stored
            const int tap = col / g.cin;
            bcoloff[i] = (int64_t)g.shift[tap] * g.ldb + (col - tap * g.cin);
            bch[i] = col - tap * g.cin;
            bslope[i] = slope4(slope, bch[i], g.act_ch);
            pb[i] = g.B + (int64_t)((tid + GEMM_THREADS * i) / (BN / 4)) * g.ldb + bcoloff[i];
        }
    }

    float4 ra[TA::NV], rb[TB::NV];
    (void)bslope; (void)arow; (void)bch;
    int ach = 0;                                              // first channel of this thread's float4 in the staged A tile
    int g_tap = (int)(kbeg / g.cin), g_ci = (int)(kbeg - (int64_t)g_tap * g.cin);   // (tap, channel) of the next gload
    auto gload = [&](int64_t k0) {
        if constexpr (MODE != CONV_WGRAD) {
            const int tap = g_tap, ci0 = g_ci;
            g_ci += BK;
            if (g_ci >= g.cin) { g_ci = 0; ++g_tap; }
            ach = ci0 + ((tid % KPR) << 2);
            if (g.tab_stride != 0) {                           // per-tap row tables (data gradient of a stride-2 conv)
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) {
                    const int64_t r = g.rowtab[(int64_t)tap * g.tab_stride + arow[i]];
                    ra[i] = ld4(g.A + r * g.lda + ci0 + (((tid + GEMM_THREADS * i) % KPR) << 2));
                }
            } else {
                const int64_t off = (int64_t)g.shift[tap] * g.lda + ci0;
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) ra[i] = ld4(pa[i] + off);
            }
            const int64_t boff = MODE == CONV_FWD ? k0 : (int64_t)ci0 * g.ldb + (int64_t)tap * g.b_tap_stride;
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) {
                const int idx = tid + GEMM_THREADS * i;
                if (TB::F4 % GEMM_THREADS != 0 && idx >= TB::F4) { rb[i] = f4_zero(); continue; }
                rb[i] = ld4(pb[i] + boff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) {
                const int idx = tid + GEMM_THREADS * i;
                if (TA::F4 % GEMM_THREADS != 0 && idx >= TA::F4) { ra[i] = f4_zero(); continue; }
                ra[i] = ld4(pa[i] + k0 * g.lda);
            }
            if (g.rowtab != nullptr) {
#pragma unroll
                for (int i = 0; i < TB::NV; ++i) {
                    const int64_t kr = g.rowtab[k0 + (tid + GEMM_THREADS * i) / (BN / 4)];
                    rb[i] = ld4(g.B + kr * g.ldb + bcoloff[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TB::NV; ++i) rb[i] = ld4(pb[i] + k0 * g.ldb);
            }
        }
    };
    auto sstore = [&](int buf) {
        if constexpr (MODE == CONV_FWD) {
            const float4 sl = slope4(slope, ach, g.act_ch);      // one float4 per thread and K tile: every staged row shares it
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) ra[i] = act4(ra[i], sl);
        } else if constexpr (MODE == CONV_WGRAD) {
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) rb[i] = act4(rb[i], bslope[i]);
        }
        TA::sstore(ra, As0 + buf * TA::FLOATS, tid);
        TB::sstore(rb, Bs0 + buf * TB::FLOATS, tid);
    };
    // Main loop scheduled like gemm.hip's (see there): fragment reads double-buffered in registers (chunk s+1's LDS reads
    // are issued before chunk s's MFMAs), the steady-state iteration one basic block with the staging instructions pinned
    // between the MFMAs of chunk SS, the iteration's one barrier in front of the LAST chunk's MFMAs with the next tile's
    // first fragments read behind it.
    auto ldfrag = [&](float (&a)[TM][4], float (&b)[TN][4], const float* as, const float* bs, int s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) TA::frag(a[i], as, (wm * TM + i) * 32 + l31, s, h);
#pragma unroll
        for (int j = 0; j < TN; ++j) TB::frag(b[j], bs, (wn * TN + j) * 32 + l31, s, h);
    };
    auto mma = [&](const float (&a)[TM][4], const float (&b)[TN][4]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
    };
    static_assert(NCH >= 2 && NCH % 2 == 0, "fragment sets alternate by chunk parity");
    constexpr int SS = NCH / 2 - 1;
    float fa[2][TM][4], fb[2][TN][4];
    const int nk = (int)((kend - kbeg + BK - 1) / BK);
    auto iter = [&](int kt, auto store_tag, auto load_tag, auto next_tag) {
        constexpr bool STORE = decltype(store_tag)::value, LOAD = decltype(load_tag)::value, NEXT = decltype(next_tag)::value;
        const int cur = kt & 1;
        const float* as = As0 + cur * TA::FLOATS;
        const float* bs = Bs0 + cur * TB::FLOATS;
#pragma unroll
        for (int s = 0; s < NCH; ++s) {
            if (s + 1 < NCH) ldfrag(fa[(s + 1) & 1], fb[(s + 1) & 1], as, bs, s + 1);
            if (s == NCH - 1) {
                if constexpr (MODE == CONV_WGRAD) {
                    if (tn == 0 && tid < BM) {                  // bias gradient: column sums of the dOut tile (before the barrier
#pragma unroll 8                                               // frees this buffer for the next iteration's staging)
                        for (int kk = 0; kk < BK; ++kk) colacc += as[kk * BM + tid];
                    }
                }
                if constexpr (NEXT) {
                    __syncthreads();
                    ldfrag(fa[0], fb[0], As0 + (cur ^ 1) * TA::FLOATS, Bs0 + (cur ^ 1) * TB::FLOATS, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == SS) {
                if constexpr (STORE) sstore(cur ^ 1);
                if constexpr (LOAD) gload(kbeg + (int64_t)(kt + 2) * BK);
            }
            mma(fa[s & 1], fb[s & 1]);
            if (s == SS && (STORE || LOAD)) {
                constexpr int N_MFMA = 4 * TM * TN, N_ST = (STORE ? TA::NV + TB::NV : 0), N_LD = (LOAD ? TA::NV + TB::NV : 0);
                constexpr int PER = (N_ST + N_LD + N_MFMA - 1) / N_MFMA;
#pragma unroll
                for (int i = 0; i < N_MFMA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int slot = i * PER + q;
                        if (slot < N_ST) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        else if (slot < N_ST + N_LD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
        }
    };
#ifdef VLG_TIMELINE
    tl_loop0 = __builtin_amdgcn_s_memrealtime();
    tl_c0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- fast path (forward / data gradient of a stride-1 convolution with one slope for every staged channel): the main
    // loop of gemm.hip's fast path - NO vector-ALU instruction for addresses.  v_mfma_f32_32x32x2_f32 and the vector ALU of
    // a SIMD execute serially (tools/micro/mfma_f32_valu_share.hip): the general loop above spends 60-140 vector instructions
    // per K tile (64-bit pointers, LDS addresses, a three-instruction PReLU with per-channel slope selects) against 16-64
    // MFMAs, i.e. 20-45 % of the matrix pipe's time.  Here: buffer loads (per-thread byte offset constant, the tap's row shift,
    // the channel block and the float4 number in the scalar offset; the descriptor ends at the last row any tap of a valid
    // output row can reach, so the rows a partial row tile reads beyond it come back as zeros), two iterations unrolled (LDS
    // addresses are immediates), the tap / channel counters on the scalar unit, PReLU as a packed multiply + one med3.
    if constexpr (MODE != CONV_WGRAD && BM >= 64) {
        const int pad = g.wp + 1;
        const int64_t a_bytes = (g.M + 2 * (int64_t)pad) * g.lda * 4;
        const int64_t b_bytes = (MODE == CONV_FWD ? (int64_t)g.N : (int64_t)g.cin) * g.ldb * 4;
        if (fast) {
            const int lda = g.lda, ldb = g.ldb;
            const __amdgpu_buffer_rsrc_t da = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A - (int64_t)pad * lda), 0, (int)a_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t db = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)b_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, 0, 0x00020000);   // empty: zeros
            const int va = TA::voff_bytes(lda, tid), vb = TB::voff_bytes(ldb, tid);
            const int a_is = TA::ISTEP * lda * 4, b_is = TB::ISTEP * ldb * 4;
            float* const aw = smem + TA::soff(tid);
            // threads beyond a short B tile (TB::F4 < threads; NV = 1 then) write their float4 to the dump area: for buffer 1 the
            // store adds TB::FLOATS to the base, so their base is the dump slot minus nothing for buffer 0 ... both land inside
            // [tiles end - TB::FLOATS, tiles end + DUMP) only if the base is chosen per buffer: two bases
            const bool surplus = TB::F4 % GEMM_THREADS != 0 && tid >= TB::F4;
            float* const bw0 = surplus ? smem + 2 * (TA::FLOATS + TB::FLOATS) + 4 * tid : smem + 2 * TA::FLOATS + TB::soff(tid);
            float* const bw1 = surplus ? bw0 : bw0 + TB::FLOATS;
            const float* ar[TM];
            const float* br[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ar[i] = smem + TA::roff((wm * TM + i) * 32 + l31, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) br[j] = smem + 2 * TA::FLOATS + TB::roff((wn * TN + j) * 32 + l31, h);
            v4f xa[TA::NV], xb[TB::NV];
            // scalar K-tile cursor of the NEXT load: tap = 3 * ky + kx, channel block ci
            int l_t = 0, l_ci = (int)(kbeg % g.cin), l_tap = (int)(kbeg / g.cin);
            int l_ky = l_tap / 3, l_kx = l_tap - 3 * l_ky;
            // (block-uniform by construction; said explicitly, or hipcc wraps every buffer load in a waterfall loop)
            const int m0i = __builtin_amdgcn_readfirstlane((int)m0), n0u = __builtin_amdgcn_readfirstlane(n0);
            l_ci = __builtin_amdgcn_readfirstlane(l_ci); l_tap = __builtin_amdgcn_readfirstlane(l_tap);
            l_ky = __builtin_amdgcn_readfirstlane(l_ky); l_kx = __builtin_amdgcn_readfirstlane(l_kx);
            const int kbeg_i = __builtin_amdgcn_readfirstlane((int)kbeg);
            auto load = [&]() __attribute__((always_inline)) {
                const bool in = l_t < nk;
                const int shift = g.sign * ((l_ky - 1) * g.wp + l_kx - 1);
                const int oa = ((m0i + shift + pad) * lda + l_ci) * 4;
                const int ob = MODE == CONV_FWD ? (n0u * ldb + kbeg_i + l_t * BK) * 4 : (l_ci * ldb + l_tap * g.b_tap_stride + n0u) * 4;
                const __amdgpu_buffer_rsrc_t ua = in ? da : dz, ub = in ? db : dz;
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) xa[i] = __builtin_amdgcn_raw_buffer_load_b128(ua, va, oa + i * a_is, 0);
#pragma unroll
                for (int i = 0; i < TB::NV; ++i) xb[i] = __builtin_amdgcn_raw_buffer_load_b128(ub, vb, ob + i * b_is, 0);
                ++l_t;                                          // (selects, not branches: the iteration stays one basic block)
                const bool wrap_c = l_ci + BK >= g.cin, wrap_x = wrap_c && l_kx == 2;
                l_ci = wrap_c ? 0 : l_ci + BK;
                l_tap += wrap_c ? 1 : 0;
                l_kx = wrap_x ? 0 : l_kx + (wrap_c ? 1 : 0);
                l_ky += wrap_x ? 1 : 0;
            };
            // PReLU of the staged operand: v > 0 ? v : a v  ==  a <= 1 ? max(v, a v) : min(v, a v)  ==  med3(v, a v, +-inf), bit for
            // bit what the general path computes (one rounding, in a v); slope 1 (no activation) skips it
            const bool act_on = MODE == CONV_FWD && slope != 1.0f;
            const float sel = slope <= 1.0f ? __builtin_inff() : -__builtin_inff();
            const v2f slope2 = v2(slope);
            auto store = [&](int c, auto act_tag) __attribute__((always_inline)) {
#ifdef VLG_TIMELINE
                {   // time this wave spends waiting for its tile loads
                    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    tl_vm += __builtin_amdgcn_s_memtime() - w0;
                }
#endif
                if constexpr (decltype(act_tag)::value) {
#pragma unroll
                    for (int i = 0; i < TA::NV; ++i) {
                        // (as asm: left to itself hipcc multiplies the four lanes one by one)
                        v2f lo, hi;
                        const v2f p0 = __builtin_shufflevector(xa[i], xa[i], 0, 1), p1 = __builtin_shufflevector(xa[i], xa[i], 2, 3);
                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(p0), "v"(slope2));
                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(p1), "v"(slope2));
                        xa[i] = v4f{__builtin_amdgcn_fmed3f(xa[i].x, lo.x, sel), __builtin_amdgcn_fmed3f(xa[i].y, lo.y, sel),
                                    __builtin_amdgcn_fmed3f(xa[i].z, hi.x, sel), __builtin_amdgcn_fmed3f(xa[i].w, hi.y, sel)};
                    }
                }
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) *reinterpret_cast<v4f*>(aw + c * TA::FLOATS + i * TA::SSTEP) = xa[i];
#pragma unroll
                for (int i = 0; i < TB::NV; ++i) *reinterpret_cast<v4f*>((c ? bw1 : bw0) + i * TB::SSTEP) = xb[i];
            };
            auto ldf = [&](float (&a)[TM][4], float (&b)[TN][4], int c, int sch) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) TA::frag_at(a[i], ar[i] + c * TA::FLOATS, 0, sch);
#pragma unroll
                for (int j = 0; j < TN; ++j) TB::frag_at(b[j], br[j] + c * TB::FLOATS, 0, sch);
            };
            auto fiter = [&](int cur, auto act_tag) __attribute__((always_inline)) {
#pragma unroll
                for (int sch = 0; sch < NCH; ++sch) {
                    if (sch + 1 < NCH) ldf(fa[(sch + 1) & 1], fb[(sch + 1) & 1], cur, sch + 1);
                    if (sch == NCH - 1) {
#ifdef VLG_TIMELINE
                        const unsigned long long b0 = __builtin_amdgcn_s_memtime();
                        __syncthreads();
                        tl_bar += __builtin_amdgcn_s_memtime() - b0;
#else
                        __syncthreads();
#endif
                        ldf(fa[0], fb[0], cur ^ 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (sch == SS) { store(cur ^ 1, act_tag); load(); }
                    mma(fa[sch & 1], fb[sch & 1]);
                    if (sch == SS) {
                        constexpr int N_MFMA = 4 * TM * TN, N_ST = TA::NV + TB::NV, N_LD = TA::NV + TB::NV;
                        constexpr int PER = (N_ST + N_LD + N_MFMA - 1) / N_MFMA;
#pragma unroll
                        for (int i = 0; i < N_MFMA; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                            for (int q = 0; q < PER; ++q) {
                                const int slot = i * PER + q;
                                if (slot < N_ST) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                                else if (slot < N_ST + N_LD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            }
                        }
                    }
                }
            };
            load();
            if (act_on) store(0, std::true_type{});
            else store(0, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            load();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            ldf(fa[0], fb[0], 0, 0);
            // (an odd nk runs one iteration on a zero tile; one copy of the loop with, one without the activation)
            if (act_on) { for (int kt = 0; kt < nk; kt += 2) { fiter(0, std::true_type{}); fiter(1, std::true_type{}); } }
            else { for (int kt = 0; kt < nk; kt += 2) { fiter(0, std::false_type{}); fiter(1, std::false_type{}); } }
        }
    }
    // ---- weight gradient, fast path (stride 1, one slope for every gathered channel): the same loop; A = dOut rows
    // [k][cout], B = gathered activation rows [k][(tap, channel)] - a thread's column, hence its tap's row shift, is fixed,
    // so its byte offset is a constant and the K advance is scalar.  The bias gradient is summed from the staging registers
    // (as gemm.hip does) instead of re-reading the LDS tile by a few threads behind a divergent branch.
    v2f wcol[TA::NV][2];
#pragma unroll
    for (int i = 0; i < TA::NV; ++i) { wcol[i][0] = v2(0.f); wcol[i][1] = v2(0.f); }
    if constexpr (MODE == CONV_WGRAD) {
        const int pad = g.wp + 1;
        const int64_t a_bytes = g.Kc * (int64_t)g.lda * 4, b_bytes = (g.Kc + 2 * (int64_t)pad) * g.ldb * 4;
        if (fast) {
            const int lda = g.lda, ldb = g.ldb;
            const __amdgpu_buffer_rsrc_t da = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, (int)a_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t db = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B - (int64_t)pad * ldb), 0, (int)b_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0, 0x00020000);
            const int m0u = __builtin_amdgcn_readfirstlane((int)m0);
            int va[TA::NV];
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) {
                const int idx = tid + GEMM_THREADS * i;
                va[i] = ((idx / (BM / 4)) * lda + m0u + ((idx % (BM / 4)) << 2)) * 4;
            }
            int col = n0 + ((tid % (BN / 4)) << 2);
            if (col > g.N - 4) col = g.N - 4;                  // columns past N are computed but never stored
            const int tapc = col / g.cin;
            const int shc = g.sign * ((tapc / 3 - 1) * g.wp + tapc % 3 - 1);
            const int vb = ((tid / (BN / 4) + shc + pad) * ldb + (col - tapc * g.cin)) * 4;
            const int b_is = TB::ISTEP * ldb * 4;
            float* const aw = smem + TA::soff(tid);
            float* const bw = smem + 2 * TA::FLOATS + TB::soff(tid);
            const float* ar[TM];
            const float* br[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ar[i] = smem + TA::roff((wm * TM + i) * 32 + l31, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) br[j] = smem + 2 * TA::FLOATS + TB::roff((wn * TN + j) * 32 + l31, h);
            v4f xa[TA::NV], xb[TB::NV];
            const int kbeg_i = __builtin_amdgcn_readfirstlane((int)kbeg);
            int l_t = 0;
            auto load = [&]() __attribute__((always_inline)) {
                const bool in = l_t < nk;
                const int k0 = kbeg_i + l_t * BK;
                const __amdgpu_buffer_rsrc_t ua = in ? da : dz, ub = in ? db : dz;
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) xa[i] = __builtin_amdgcn_raw_buffer_load_b128(ua, va[i], k0 * lda * 4, 0);
#pragma unroll
                for (int i = 0; i < TB::NV; ++i) xb[i] = __builtin_amdgcn_raw_buffer_load_b128(ub, vb, k0 * ldb * 4 + i * b_is, 0);
                ++l_t;
            };
            const bool act_on = slope != 1.0f;
            const float sel = slope <= 1.0f ? __builtin_inff() : -__builtin_inff();
            const v2f slope2 = v2(slope);
            auto store = [&](int c, auto act_tag) __attribute__((always_inline)) {
#ifdef VLG_TIMELINE
                {
                    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    tl_vm += __builtin_amdgcn_s_memtime() - w0;
                }
#endif
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) {
                    const v2f lo = __builtin_shufflevector(xa[i], xa[i], 0, 1), hi = __builtin_shufflevector(xa[i], xa[i], 2, 3);
                    asm("v_pk_add_f32 %0, %0, %1" : "+v"(wcol[i][0]) : "v"(lo));
                    asm("v_pk_add_f32 %0, %0, %1" : "+v"(wcol[i][1]) : "v"(hi));
                }
                if constexpr (decltype(act_tag)::value) {
#pragma unroll
                    for (int i = 0; i < TB::NV; ++i) {
                        v2f lo, hi;
                        const v2f p0 = __builtin_shufflevector(xb[i], xb[i], 0, 1), p1 = __builtin_shufflevector(xb[i], xb[i], 2, 3);
                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(p0), "v"(slope2));
                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(p1), "v"(slope2));
                        xb[i] = v4f{__builtin_amdgcn_fmed3f(xb[i].x, lo.x, sel), __builtin_amdgcn_fmed3f(xb[i].y, lo.y, sel),
                                    __builtin_amdgcn_fmed3f(xb[i].z, hi.x, sel), __builtin_amdgcn_fmed3f(xb[i].w, hi.y, sel)};
                    }
                }
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) *reinterpret_cast<v4f*>(aw + c * TA::FLOATS + i * TA::SSTEP) = xa[i];
#pragma unroll
                for (int i = 0; i < TB::NV; ++i) *reinterpret_cast<v4f*>(bw + c * TB::FLOATS + i * TB::SSTEP) = xb[i];
            };
            auto ldf = [&](float (&a)[TM][4], float (&b)[TN][4], int c, int sch) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) TA::frag_at(a[i], ar[i] + c * TA::FLOATS, 0, sch);
#pragma unroll
                for (int j = 0; j < TN; ++j) TB::frag_at(b[j], br[j] + c * TB::FLOATS, 0, sch);
            };
            auto fiter = [&](int cur, auto act_tag) __attribute__((always_inline)) {
#pragma unroll
                for (int sch = 0; sch < NCH; ++sch) {
                    if (sch + 1 < NCH) ldf(fa[(sch + 1) & 1], fb[(sch + 1) & 1], cur, sch + 1);
                    if (sch == NCH - 1) {
#ifdef VLG_TIMELINE
                        const unsigned long long b0 = __builtin_amdgcn_s_memtime();
                        __syncthreads();
                        tl_bar += __builtin_amdgcn_s_memtime() - b0;
#else
                        __syncthreads();
#endif
                        ldf(fa[0], fb[0], cur ^ 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (sch == SS) { store(cur ^ 1, act_tag); load(); }
                    mma(fa[sch & 1], fb[sch & 1]);
                    if (sch == SS) {
                        constexpr int N_MFMA = 4 * TM * TN, N_ST = TA::NV + TB::NV, N_LD = TA::NV + TB::NV;
                        constexpr int PER = (N_ST + N_LD + N_MFMA - 1) / N_MFMA;
#pragma unroll
                        for (int i = 0; i < N_MFMA; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                            for (int q = 0; q < PER; ++q) {
                                const int slot = i * PER + q;
                                if (slot < N_ST) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                                else if (slot < N_ST + N_LD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            }
                        }
                    }
                }
            };
            load();
            if (act_on) store(0, std::true_type{});
            else store(0, std::false_type{});
            __builtin_amdgcn_sched_barrier(0);
            load();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            ldf(fa[0], fb[0], 0, 0);
            if (act_on) { for (int kt = 0; kt < nk; kt += 2) { fiter(0, std::true_type{}); fiter(1, std::true_type{}); } }
            else { for (int kt = 0; kt < nk; kt += 2) { fiter(0, std::false_type{}); fiter(1, std::false_type{}); } }
        }
    }
    if (!fast) {
    if (nk > 0) { gload(kbeg); sstore(0); }
    if (nk > 1) gload(kbeg + BK);
    __syncthreads();
    if (nk > 0) ldfrag(fa[0], fb[0], As0, Bs0, 0);
    {
        int kt = 0;
        for (; kt + 2 < nk; ++kt) iter(kt, std::true_type{}, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { iter(kt, std::true_type{}, std::false_type{}, std::true_type{}); ++kt; }
        if (kt < nk) iter(kt, std::false_type{}, std::false_type{}, std::false_type{});
    }
    }
    __syncthreads();
#ifdef VLG_TIMELINE
    tl_loop1 = __builtin_amdgcn_s_memrealtime();
    tl_c1 = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h)
    // (a K range of a tail tile stores its raw partial sums: rows are addressed as in the full tensor, so the base is moved back)
    float* Cs = part ? g.tail_ws + (int64_t)split * g.tail_stride - g.tail_row0 * g.ldc : g.C + (int64_t)split * g.slab_stride;
    const int epi = part ? 0 : g.epi;
    const float* const rowmask = part ? nullptr : g.rowmask;
    const int act_cut = part ? g.N : g.act_ch;                 // data gradient: first constant (AddCoords) channel
    const int64_t row0 = m0 + wm * TM * 32 + 4 * h;
    const int col0 = n0 + wn * TN * 32 + l31;
    float da = 0.f;
    // fast epilogue (with the fast main loop; every column of the tile is stored, the data gradient cuts no constant
    // channels): buffer loads / stores - the descriptors end with row M, so the rows of a partial row tile beyond it are
    // dropped by the hardware instead of by a compare per element - one byte offset per thread, the row in the scalar
    // offset; two rows at a time on packed instructions.  The general epilogue below spends ~15 vector instructions per
    // element (64-bit offsets, bounds, flag tests): a quarter of a 32-channel tile's matrix time.
    bool fast_epi = false;
    if constexpr (MODE != CONV_WGRAD) fast_epi = fast && (MODE != CONV_DGRAD || act_cut >= g.N) && g.M * (int64_t)g.ldc * 4 < (1ll << 31);
    if constexpr (MODE != CONV_WGRAD) {
      if (fast_epi) {
        const int m0u = __builtin_amdgcn_readfirstlane((int)m0), n0u = __builtin_amdgcn_readfirstlane(n0);
        const int cbytes = (int)(g.M * (int64_t)g.ldc * 4);
        const __amdgpu_buffer_rsrc_t dC = __builtin_amdgcn_make_buffer_rsrc(Cs, 0, cbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t dX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.aux_in ? g.aux_in : g.A), 0, g.aux_in ? cbytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t dMk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowmask ? rowmask : g.A), 0, rowmask ? (int)(g.M * 4) : 0, 0x00020000);
        const int vc = ((wm * TM * 32 + 4 * h) * g.ldc + wn * TN * 32 + l31) * 4;
        const int vm = (wm * TM * 32 + 4 * h) * 4;
        const bool e_resid = (epi & VLG_CEPI_RESID) != 0, e_prelu = (epi & VLG_CEPI_PRELU) != 0, e_dprelu = (epi & VLG_CEPI_DPRELU) != 0,
                   e_accum = (epi & VLG_CEPI_ACCUM) != 0, e_mask = rowmask != nullptr;
        const float sel = slope <= 1.0f ? __builtin_inff() : -__builtin_inff();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            v2f mk[8];                                            // the row mask of this 32-row group (shared by its column tiles)
            if (e_mask) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mk[r >> 1][r & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dMk, vm, (m0u + i * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const auto so = [&](int r) { return ((m0u + i * 32 + (r & 3) + 8 * (r >> 2)) * g.ldc + n0u) * 4; };
                v2f x[8], o[8], v[8];
                if (e_resid || e_dprelu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[r >> 1][r & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dX, vc + j * 128, so(r), 0));
                }
                if (e_accum) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r >> 1][r & 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dC, vc + j * 128, so(r), 0));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = v2f{acc[i][j][2 * q], acc[i][j][2 * q + 1]};
                if constexpr (MODE == CONV_FWD) {
                    if (g.bias != nullptr && !part) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += v2(bv[j]);
                    }
                    if (e_resid) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += x[q];
                    }
                    if (e_prelu) {                              // v > 0 ? v : slope v  ==  med3(v, slope v, +-inf)
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const v2f t = v[q] * v2(slope);
                            v[q] = v2f{__builtin_amdgcn_fmed3f(v[q].x, t.x, sel), __builtin_amdgcn_fmed3f(v[q].y, t.y, sel)};
                        }
                    }
                }
                if (e_mask) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= mk[q];
                }
                if constexpr (MODE == CONV_DGRAD) {
                    if (e_dprelu) {                             // slope gradient += v x [x <= 0];  v *= x > 0 ? 1 : slope
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            da = fmaf(v[q].x, __builtin_amdgcn_fmed3f(x[q].x, 0.f, -__builtin_inff()), da);
                            da = fmaf(v[q].y, __builtin_amdgcn_fmed3f(x[q].y, 0.f, -__builtin_inff()), da);
                            v[q] *= v2f{x[q].x > 0.f ? 1.0f : slope, x[q].y > 0.f ? 1.0f : slope};
                        }
                    }
                    if (e_accum) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += o[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].x), dC, vc + j * 128, so(2 * q), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].y), dC, vc + j * 128, so(2 * q + 1), 0);
                }
            }
        }
      }
    }
    if (!fast_epi)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = col0 + j * 32;
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= g.M) continue;
                const int64_t o = row * g.ldc + col;
                float v = acc[i][j][r] + bv[j];
                if constexpr (MODE != CONV_WGRAD) {
                    if (epi & VLG_CEPI_RESID) v += g.aux_in[o];
                    if (epi & VLG_CEPI_PRELU) v = prelu_f(v, slope);
                    if (rowmask != nullptr) v *= rowmask[row];
                    // data gradient: appended AddCoords channels are constants, not outputs of the producing conv -
                    // their gradient must not flow on (it would train weight rows that no forward pass uses)
                    if (MODE == CONV_DGRAD && col >= act_cut) v = 0.f;
                    if (epi & VLG_CEPI_DPRELU) {
                        const float x = g.aux_in[o];
                        if (col < act_cut) {
                            da += x > 0.f ? 0.f : v * x;
                            v *= x > 0.f ? 1.0f : slope;
                        }
                    }
                    if (epi & VLG_CEPI_ACCUM) v += Cs[o];
                }
                Cs[o] = v;
            }
        }
    if constexpr (MODE == CONV_WGRAD) {
        if (fast) {
            // bias gradient: (thread, float4 i) summed columns 4 * (idx % (BM / 4)) .. + 3 over its K rows; 32 such partials per column
            if (tn == 0) {
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) st4(smem + 4 * (tid + GEMM_THREADS * i), make_float4(wcol[i][0].x, wcol[i][0].y, wcol[i][1].x, wcol[i][1].y));
                __syncthreads();
                if (tid < BM && m0 + tid < g.M) {
                    float sacc = 0.f;
#pragma unroll 8
                    for (int e = 0; e < GEMM_THREADS * TA::NV / (BM / 4); ++e) sacc += smem[4 * ((tid >> 2) + e * (BM / 4)) + (tid & 3)];
                    Cs[g.colsum_off + m0 + tid] = sacc;
                }
            }
        } else if (tn == 0 && tid < BM && m0 + tid < g.M) Cs[g.colsum_off + m0 + tid] = colacc;
    }
    if constexpr (MODE == CONV_DGRAD) {
        if (g.da_slab != nullptr) {                             // slope gradient: one partial per block
            da = block_sum(da, red);
            if (tid == 0) g.da_slab[bid] = da;
        }
    }
#ifdef VLG_TIMELINE
    if (g.probe) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tl_exit = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* o = g.probe + 8 * (size_t)bid;
            o[0] = tl_entry; o[1] = tl_loop0; o[2] = tl_loop1; o[3] = tl_exit;
            o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID
            o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));       // HW_REG_XCC_ID
            o[6] = (unsigned long long)tile;
            o[7] = ((tl_c1 - tl_c0) << 40) | ((tl_vm & 0xfffff) << 20) | (tl_bar & 0xfffff);   // shader clocks: loop | load waits | barrier waits (wave 0)
        }
    }
#endif
}

// ---- first layers of the frozen trunks (3 image channels in a 32-channel padded tensor): as a 9 x 32-deep implicit GEMM they
// multiply 29 zero channels per tap (6-8 TFLOP/s algorithmic, 117 us at 4 x 256 x 256).  Here the contraction is over (tap, 4
// channels): k = 4 * tap + c, 12 tap slots (9 real), K = 48.  A row's tap slot is ONE 16-byte load (channels 0..3 of the
// shifted pixel) that lands 16-byte aligned in the [rows][48 + 4] LDS tile; the weights are read from the standard
// [cout][9][cin_p] layout the same way (channels 0..3 of each tap; slots 9..11 are zero).  One K range, no main loop: a block
// loads its 128 x 48 and 64 x 48 tiles once, runs 24 MFMAs per 32 x 32 output tile and stores; several blocks per CU
// (40 KB of LDS each) overlap each other's loads.  Output bound: 64 channels x 4 B per pixel.
__global__ __launch_bounds__(GEMM_THREADS, 2) void conv_first_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     const float* __restrict__ rowmask, int M, int lda, int ldb,
                                                                     int ldc, int N, int wp) {
    constexpr int BM = 128, BN = 64, KS = 12, LDK = 4 * KS + 4;
    __shared__ __attribute__((aligned(16))) float As[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int tiles_n = (N + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = __builtin_amdgcn_readfirstlane(tm * BM), n0 = __builtin_amdgcn_readfirstlane(tn * BN);
    // Buffer loads: a thread owns tap slot 4g + (tid & 3) of rows tid / 4 and tid / 4 + 64 for g = 0, 1, 2 - its byte offset
    // (row, tap shift) is one register per g, the tile's first row is the scalar offset.  The descriptor spans the guard band
    // in front of the tensor and ends behind the last row a tap can reach: rows past M read as zeros, no clamps.  Slots 9..11
    // (g = 2, tid & 3 != 0) get an offset beyond the descriptor: zeros on both operands.
    const int pad = wp + 1, t3 = tid & 3, r4 = tid >> 2;
    const __amdgpu_buffer_rsrc_t dA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in - (int64_t)pad * lda), 0,
                                                                        (int)(((int64_t)M + 2 * pad) * lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t dB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)((int64_t)N * ldb * 4), 0x00020000);
    v4f xa[6], xb[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int tap = 4 * g + t3;
        const bool real = tap < 9;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int va = real ? (r4 + (ky - 1) * wp + kx - 1 + pad) * lda * 4 : 0x7ffffff0;
        const int vb = real ? (r4 * ldb + tap * lda) * 4 : 0x7ffffff0;
        xa[2 * g] = __builtin_amdgcn_raw_buffer_load_b128(dA, va, m0 * lda * 4, 0);
        xa[2 * g + 1] = __builtin_amdgcn_raw_buffer_load_b128(dA, va, (m0 + 64) * lda * 4, 0);
        xb[g] = __builtin_amdgcn_raw_buffer_load_b128(dB, vb, n0 * ldb * 4, 0);
    }
    // epilogue operands fetched now, behind the tile loads: the halo mask of this lane's 32 rows, the bias of its column
    const int wm = wave >> 1, wn = wave & 1;                   // 2 x 2 waves: 64 rows x 32 columns each
    const __amdgpu_buffer_rsrc_t dMk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rowmask ? rowmask : in), 0, rowmask ? M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dBi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias ? bias : in), 0, bias ? N * 4 : 0, 0x00020000);
    float mk[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            mk[i][r] = rowmask ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dMk, (wm * 64 + 4 * h) * 4, (m0 + i * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0)) : 1.0f;
    const float bv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dBi, (wn * 32 + l31) * 4, n0 * 4, 0));
    float* const aw = As + r4 * LDK + 4 * t3;
    float* const bw = Bs + r4 * LDK + 4 * t3;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        *reinterpret_cast<v4f*>(aw + 16 * g) = xa[2 * g];
        *reinterpret_cast<v4f*>(aw + 64 * LDK + 16 * g) = xa[2 * g + 1];
        *reinterpret_cast<v4f*>(bw + 16 * g) = xb[g];
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float* const ar = As + (wm * 64 + l31) * LDK + 4 * h;
    const float* const br = Bs + (wn * 32 + l31) * LDK + 4 * h;
#pragma unroll
    for (int sc = 0; sc < 4 * KS / 8; ++sc) {
        const float4 b = ld4(br + 8 * sc);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 a = ld4(ar + i * 32 * LDK + 8 * sc);
            const float aa[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[kk], bb[kk], acc[i], 0, 0, 0);
        }
    }
    // stores through a descriptor that ends with row M and (per row) is entered only by columns < N: a lane of a column past
    // N takes an offset beyond it, rows past M fall off its end - dropped by the hardware
    const __amdgpu_buffer_rsrc_t dC = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((int64_t)M * ldc * 4), 0x00020000);
    const int vc = n0 + wn * 32 + l31 < N ? ((wm * 64 + 4 * h) * ldc + wn * 32 + l31) * 4 : 0x7ffffff0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (acc[i][r] + bv) * mk[i][r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dC, vc, ((m0 + i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + n0) * 4, 0);
        }
}

#ifdef VLG_TIMELINE
#ifdef VLG_DIAG          /* diagnostic build only (csrc/gemm.hip): the product library keeps no mutable process-wide state */
static unsigned long long* vlg_conv_probe = nullptr;
extern "C" void vlg_debug_set_conv_probe(unsigned long long* p) { vlg_conv_probe = p; }
#else
static constexpr unsigned long long* vlg_conv_probe = nullptr;
#endif
#endif
template <int MODE, int BM, int BN, int BK = 32>
static int launch_conv(ConvArgs g, hipStream_t s) {
#ifdef VLG_TIMELINE
    g.probe = vlg_conv_probe;
#endif
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (g.N + BN - 1) / BN;
    const int64_t blocks = g.tail_tiles > 0 ? (int64_t)g.tiles_m * g.tiles_n - g.tail_tiles + (((int64_t)g.tail_tiles * g.tail_splits + 7) & ~7ll)
                                            : (int64_t)g.tiles_m * g.tiles_n * g.splits;
    if (blocks < 1 || blocks > 0x7fffffff) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, BM, BN, BK>), dim3((unsigned)blocks), dim3(GEMM_THREADS), 0, s, g);
    return vlg_last_error();
}

static void fill_shifts(ConvArgs& g, int wp, int sign) {
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g.shift[ky * 3 + kx] = sign * ((ky - 1) * wp + (kx - 1));
    g.wp = wp; g.sign = sign;
}

// development switch: VLG_CONV_NARROW_BK=16|32 selects the K-tile depth of the 32-channel tiles
static int conv_narrow_bk() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VLG_CONV_NARROW_BK"); v = e ? atoi(e) : 16; }   // 16: +2.7 % on CoordGridNet b=4 256x256 (more blocks per CU)
    return v;
}

// Tile choice at small batch / coarse levels.  Blocks on one CU share its MFMA pipes, so a launch takes about
// max-blocks-per-CU x (work of one block): 529 blocks of 128 rows cost 3 units on 256 CUs, 1057 blocks of 64 rows
// cost 5 half-units.  Prefer the finer tiling whenever it lowers that bound by more than the efficiency it loses.
static int64_t cu_units(int64_t blocks) { return (blocks + 255) / 256; }
static bool few_blocks(int64_t rows, int tiles_n) {
    const int64_t b128 = ((rows + 127) / 128) * tiles_n, b64 = ((rows + 63) / 64) * tiles_n;
    return b128 < 384 || 10 * cu_units(b64) < 17 * cu_units(b128);      // half-size blocks: 2*u128 vs u64, keep a 15 % margin
}
// 96 output channels as three 32-wide column tiles when there are too few row tiles to fill the chip
static bool split_96(int64_t rows) { return (rows + 127) / 128 < 400; }

static bool conv_ok(const void* p) { return p != nullptr && vlg_aligned16(p); }

struct ConvTile { int bm, bn, bk; };

// Tile-count quantisation.  The blocks of a CU share its matrix pipes, so a launch lasts (tiles of the busiest CU) x (time of
// one tile): 1057 tiles on 256 CUs cost 5 tile times for 4.13 tile times of work (measured: 128 -> 128 channels at 4 x 126 x 126
// pixels, exactly 1024 tiles, 156 us; at 4 x 128 x 128, 1057 tiles, 199 us).  Plan: the tiles beyond the last full round of 256
// (whole row tiles) are cut along K into `splits` ranges, computed by the last blocks of the same launch into a workspace and
// summed (+ epilogue) by the finish kernel - when the tile is long enough to pay for that second launch.
// VLG_CONV_TAIL=0 switches it off (development).
struct ConvTail { int tiles, splits; int64_t kc, row0, floats; };
static bool conv_tail_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VLG_CONV_TAIL"); v = e ? atoi(e) : 1; }
    return v != 0;
}
static ConvTail conv_tail_plan(int64_t rows, const ConvTile t, int n_cols, int64_t kc, int ldc) {
    ConvTail none{0, 1, 0, 0, 0};
    if (!conv_tail_on() || (kc % t.bk) != 0) return none;
    const int64_t tiles_m = (rows + t.bm - 1) / t.bm;
    const int tiles_n = (n_cols + t.bn - 1) / t.bn;
    const int64_t tiles = tiles_m * tiles_n;
    const int64_t rem = tiles % 256;
    if (tiles < 256 || rem == 0) return none;
    const int64_t tail_rows_t = (rem + tiles_n - 1) / tiles_n;
    const int64_t tail_tiles = tail_rows_t * tiles_n;
    if (tail_tiles >= tiles) return none;
    const int ktiles = (int)(kc / t.bk);
    const double tile_us = 2.0 * t.bm * t.bn * (double)kc / 0.5e6;         // one tile on one CU at ~0.5 TFLOP/s (80 % of its share)
    double best = 0.0;
    int best_per = 0;
    for (int sp = 2; sp <= 8; ++sp) {
        const int per = (ktiles + sp - 1) / sp;
        if (per * t.bk < 128) break;                                        // at least 128 of contraction per block
        const int real = (ktiles + per - 1) / per;
        const double rounds = (double)((tail_tiles * real + 255) / 256);
        const double saving = tile_us * (1.0 - rounds * per / ktiles) - rounds * 3.0 - 5.0;   // 3 us per short block, 5 us finish launch
        if (saving > best) { best = saving; best_per = per; }
    }
    if (best < 3.0) return none;
    ConvTail r;
    r.tiles = (int)tail_tiles;
    r.splits = (ktiles + best_per - 1) / best_per;
    r.kc = (int64_t)best_per * t.bk;
    r.row0 = (tiles_m - tail_rows_t) * t.bm;
    r.floats = (int64_t)r.splits * (rows - r.row0) * ldc;
    return r;
}

// tile of a forward (n_cols = cout_p, n_valid = cout) or data-gradient (n_cols = n_valid = cin_p) launch; kc > 0 = the launch
// may use the tail plan (a workspace was given): then 128-row tiles whenever they fill the chip once, the remainder is the plan's
static ConvTile conv_tile(int64_t rows, int n_cols, int n_valid, int64_t kc = 0) {
    if (n_cols == 32) return ConvTile{128, 32, conv_narrow_bk() == 16 ? 16 : 32};
    if (n_cols == 96) return split_96(rows) ? ConvTile{128, 32, 32} : ConvTile{128, 96, 32};
    const int bn = n_cols == 64 ? 64 : 128;
    const int tiles_n = n_cols == 64 ? 1 : (n_valid + 127) / 128;
    if (kc > 0) {
        const ConvTile big{128, bn, 32};
        const int64_t t128 = ((rows + 127) / 128) * tiles_n;
        if (t128 >= 256 && (t128 % 256 == 0 || conv_tail_plan(rows, big, n_valid, kc, 1).tiles > 0)) return big;
    }
    // small grids (coarse levels, small batches) would leave CUs idle with 128-row tiles: halve the tile height
    const bool small = few_blocks(rows, tiles_n);
    return ConvTile{small ? 64 : 128, bn, 32};
}
template <int MODE>
static int launch_conv_tile(const ConvTile t, const ConvArgs& g, hipStream_t s) {
    if (t.bn == 32) return t.bk == 16 ? launch_conv<MODE, 128, 32, 16>(g, s) : launch_conv<MODE, 128, 32>(g, s);
    if (t.bn == 96) return launch_conv<MODE, 128, 96>(g, s);
    if (t.bn == 64) return t.bm == 64 ? launch_conv<MODE, 64, 64>(g, s) : launch_conv<MODE, 128, 64>(g, s);
    return t.bm == 64 ? launch_conv<MODE, 64, 128>(g, s) : launch_conv<MODE, 128, 128>(g, s);
}

// Split-K forward for the coarse levels of the VGG / HED trunks (4 x 32 x 32 or 16 x 16 pixels, 256-512 channels): the
// plain launch has 44-148 blocks for 256 CUs while K = 9*cin is 2304-4608, so the contraction is cut into `splits`
// ranges, every range writes a raw partial tile to the caller's workspace and conv_finish_kernel sums them and applies
// the epilogue (bias, residual, row mask).  Partial sums are added in a fixed order: reproducible.
static int conv_fwd_splits(int64_t rows_out, int cin_p, int cout, int cout_p) {
    if (cout_p < 128 || cout != cout_p || cin_p < 128) return 1;
    const int64_t b128 = ((rows_out + 127) / 128) * ((cout_p + 127) / 128);
    if (b128 >= 200) return 1;                                 // (splitting 200-400 block launches in two was measured: no gain)
    int s = (int)(512 / b128);
    if (s > 8) s = 8;
    const int ktiles = 9 * cin_p / 32;
    while (s > 1 && ktiles / s < 8) --s;                       // keep at least 8 K tiles per block
    return s < 2 ? 1 : s;
}
extern "C" int vlg_conv3x3_fwd_splits(int64_t rows_out, int cin_p, int cout, int cout_p) {
    return conv_fwd_splits(rows_out, cin_p, cout, cout_p);
}
extern "C" int64_t vlg_conv3x3_fwd_workspace(int64_t rows_out, int cin_p, int cout, int cout_p) {
    if (rows_out < 1 || cin_p < 32 || cout_p < 32) return 0;
    const int splits = conv_fwd_splits(rows_out, cin_p, cout, cout_p);
    if (splits > 1) return (int64_t)splits * rows_out * cout_p;
    return conv_tail_plan(rows_out, conv_tile(rows_out, cout_p, cout, 9 * (int64_t)cin_p), cout, 9 * (int64_t)cin_p, cout_p).floats;
}

__global__ __launch_bounds__(256) void conv_finish_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride,
                                                          const float* __restrict__ bias, const float* __restrict__ resid,
                                                          const float* __restrict__ rowmask, float* __restrict__ out,
                                                          int64_t rows, int cq /* float4 per row */) {
    const int64_t n4 = rows * cq;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / cq;
        const int c4 = (int)(e - row * cq);
        float4 acc = ld4(slabs + e * 4);
        for (int s = 1; s < splits; ++s) acc = f4_add(acc, ld4(slabs + s * slab_stride + e * 4));
        if (bias != nullptr) acc = f4_add(acc, ld4(bias + c4 * 4));
        if (resid != nullptr) acc = f4_add(acc, ld4(resid + e * 4));
        if (rowmask != nullptr) { const float m = rowmask[row]; acc = make_float4(acc.x * m, acc.y * m, acc.z * m, acc.w * m); }
        st4(out + e * 4, acc);
    }
}

// data-gradient counterpart: din = mask * sum (zeroed on the constant AddCoords channels) * act'(x_in) (+ din)
__global__ __launch_bounds__(256) void conv_finish_dgrad_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride,
                                                                const float* __restrict__ x_in, const float* __restrict__ rowmask,
                                                                float slope_is_valid, const float* __restrict__ slope_p,
                                                                float* __restrict__ din, int64_t rows, int cq, int act_ch, int epi) {
    const int64_t n4 = rows * cq;
    const float slope = slope_is_valid != 0.f ? slope_p[0] : 1.0f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / cq;
        const int c = (int)(e - row * cq) * 4;
        float4 acc = ld4(slabs + e * 4);
        for (int s = 1; s < splits; ++s) acc = f4_add(acc, ld4(slabs + s * slab_stride + e * 4));
        float v[4] = {acc.x, acc.y, acc.z, acc.w};
        const float m = rowmask != nullptr ? rowmask[row] : 1.0f;
        float4 x = f4_zero();
        if (epi & VLG_CEPI_DPRELU) x = ld4(x_in + e * 4);
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] *= m;
            if (c + k >= act_ch) v[k] = 0.f;
            else if ((epi & VLG_CEPI_DPRELU) && !(xv[k] > 0.f)) v[k] *= slope;
        }
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (epi & VLG_CEPI_ACCUM) o = f4_add(o, ld4(din + e * 4));
        st4(din + e * 4, o);
    }
}

extern "C" int vlg_conv3x3_fwd(const float* in, const float* w, const float* bias, float* out, const float* resid,
                               const float* rowmask, const float* prelu_slope, const int* rowtab, int64_t rows_out,
                               int cin_p, int cout, int cout_p, int wp_in, int act_ch, int epilogue, float* workspace,
                               int64_t workspace_capacity, void* stream) {
    if (rows_out < 1 || cin_p < 32 || (cin_p & 31) || cout < 1 || cout > cout_p || (cout_p & 31)) return VLG_ERR_SHAPE;
    if (!conv_ok(in) || !conv_ok(w) || !conv_ok(out)) return VLG_ERR_ALIGN;
    if ((epilogue & VLG_CEPI_RESID) && !resid) return VLG_ERR_SHAPE;
    ConvArgs g{};
    g.A = in; g.B = w; g.C = out; g.bias = bias; g.aux_in = resid; g.rowmask = rowmask; g.prelu = prelu_slope;
    g.rowtab = rowtab; g.tab_stride = 0;
    g.M = rows_out; g.N = cout; g.Kc = 9 * (int64_t)cin_p;
    g.lda = cin_p; g.ldb = 9 * cin_p; g.ldc = cout_p; g.cin = cin_p;
    g.splits = 1; g.kc_per_split = g.Kc; g.epi = epilogue & ~(VLG_CEPI_DPRELU | VLG_CEPI_CIN4); g.act_ch = act_ch;
    fill_shifts(g, wp_in, rowtab ? 1 : 1);
    hipStream_t s = (hipStream_t)stream;
    if (epilogue & VLG_CEPI_CIN4) {                            // image-channel first layer: contraction over (tap, 4 channels)
        if (rowtab != nullptr || prelu_slope != nullptr || (epilogue & (VLG_CEPI_RESID | VLG_CEPI_PRELU)) || wp_in < 1) return VLG_ERR_SHAPE;
        const int64_t blocks = ((rows_out + 127) / 128) * ((cout + 63) / 64);
        // (32-bit byte offsets inside the kernel's buffer descriptors)
        if (blocks > 0x7fffffff || (rows_out + 2 * (int64_t)(wp_in + 1) + 128) * cin_p * 4 >= (1ll << 31) ||
            (rows_out + 128) * (int64_t)cout_p * 4 >= (1ll << 31)) return VLG_ERR_SHAPE;
        hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)blocks), dim3(GEMM_THREADS), 0, s, in, w, bias, out, rowmask, (int)rows_out,
                           cin_p, 9 * cin_p, cout_p, cout, wp_in);
        return vlg_last_error();
    }
    const int splits = workspace != nullptr ? conv_fwd_splits(rows_out, cin_p, cout, cout_p) : 1;
    if (splits > 1) {
        if (!vlg_aligned16(workspace) || (epilogue & VLG_CEPI_PRELU)) return VLG_ERR_ALIGN;
        if (workspace_capacity < (int64_t)splits * rows_out * cout_p) return VLG_ERR_SHAPE;     // host-side bound on the partial tiles
        const int ktiles = 9 * cin_p / 32;
        g.splits = splits;
        g.kc_per_split = (int64_t)((ktiles + splits - 1) / splits) * 32;
        g.slab_stride = rows_out * (int64_t)cout_p;
        g.C = workspace; g.bias = nullptr; g.aux_in = nullptr; g.rowmask = nullptr; g.epi = 0;    // raw partial tiles
        if (int e = launch_conv<CONV_FWD, 128, 128>(g, s)) return e;
        const int64_t n4 = rows_out * (cout_p / 4);
        int64_t blocks = (n4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, splits, g.slab_stride, bias,
                           (epilogue & VLG_CEPI_RESID) ? resid : nullptr, rowmask, out, rows_out, cout_p / 4);
        return vlg_last_error();
    }
    const bool tail_ok = workspace != nullptr && rowtab == nullptr && !(epilogue & VLG_CEPI_PRELU) && vlg_aligned16(workspace);
    const ConvTile t = conv_tile(rows_out, cout_p, cout, tail_ok ? g.Kc : 0);
    if (tail_ok) {
        const ConvTail tl = conv_tail_plan(rows_out, t, cout, g.Kc, cout_p);
        if (tl.tiles > 0 && tl.floats <= workspace_capacity) {
            g.tail_tiles = tl.tiles; g.tail_splits = tl.splits; g.tail_kc = tl.kc; g.tail_row0 = tl.row0;
            g.tail_stride = (rows_out - tl.row0) * (int64_t)cout_p; g.tail_ws = workspace;
            if (int e = launch_conv_tile<CONV_FWD>(t, g, s)) return e;
            const int64_t off = tl.row0 * (int64_t)cout_p, n4 = (rows_out - tl.row0) * (cout_p / 4);
            hipLaunchKernelGGL(conv_finish_kernel, dim3((unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256)), dim3(256), 0, s,
                               workspace, tl.splits, g.tail_stride, bias, (epilogue & VLG_CEPI_RESID) ? resid + off : nullptr,
                               rowmask ? rowmask + tl.row0 : nullptr, out + off, rows_out - tl.row0, cout_p / 4);
            return vlg_last_error();
        }
    }
    return launch_conv_tile<CONV_FWD>(t, g, s);
}

extern "C" int vlg_conv3x3_dgrad_slabs(int64_t rows_in, int cin_p) {
    const ConvTile t = conv_tile(rows_in, cin_p, cin_p);       // one partial per block of the launch
    return (int)((rows_in + t.bm - 1) / t.bm) * ((cin_p + t.bn - 1) / t.bn);
}

// K ranges of the data gradient when given a workspace: frozen trunks only (no slope gradient wanted), stride 1
static int conv_dgrad_splits(int64_t rows_in, int cin_p, int cout_p) { return conv_fwd_splits(rows_in, cout_p, cin_p, cin_p); }
extern "C" int vlg_conv3x3_dgrad_splits(int64_t rows_in, int cin_p, int cout_p) { return conv_dgrad_splits(rows_in, cin_p, cout_p); }
extern "C" int64_t vlg_conv3x3_dgrad_workspace(int64_t rows_in, int cin_p, int cout_p) {
    if (rows_in < 1 || cin_p < 32 || cout_p < 32) return 0;
    const int splits = conv_dgrad_splits(rows_in, cin_p, cout_p);
    if (splits > 1) return (int64_t)splits * rows_in * cin_p;
    return conv_tail_plan(rows_in, conv_tile(rows_in, cin_p, cin_p, 9 * (int64_t)cout_p), cin_p, 9 * (int64_t)cout_p, cin_p).floats;
}

extern "C" int vlg_conv3x3_dgrad(const float* dout, const float* w, float* din, const float* x_in,
                                 const float* rowmask_in, const float* prelu_slope, float* da_slab,
                                 const int* tap_tables, int64_t tab_stride, int64_t rows_in, int cin_p, int cout_p,
                                 int wp, int act_ch, int epilogue, float* workspace, int64_t workspace_capacity,
                                 int da_capacity, void* stream) {
    // din[p, ci] = mask[p] * sum_tap sum_co dout[p - shift(tap), co] * W[co][tap][ci]   (stride 1)
    // stride 2: tap_tables[tap][p] = output row feeding input row p through that tap (or a zero guard row)
    if (rows_in < 1 || cin_p < 32 || (cin_p & 31) || (cin_p > 128 && (cin_p & 127)) || cout_p < 32 || (cout_p & 31))
        return VLG_ERR_SHAPE;                                // wide inputs are tiled 128 columns at a time
    if (!conv_ok(dout) || !conv_ok(w) || !conv_ok(din)) return VLG_ERR_ALIGN;
    if ((epilogue & VLG_CEPI_DPRELU) && (!x_in || !prelu_slope)) return VLG_ERR_SHAPE;
    if (da_slab != nullptr && da_capacity < vlg_conv3x3_dgrad_slabs(rows_in, cin_p)) return VLG_ERR_SHAPE;   // one partial per block
    ConvArgs g{};
    g.A = dout; g.B = w; g.C = din; g.aux_in = x_in; g.rowmask = rowmask_in; g.prelu = prelu_slope; g.da_slab = da_slab;
    g.rowtab = tap_tables; g.tab_stride = tap_tables ? tab_stride : 0;
    g.M = rows_in; g.N = cin_p; g.Kc = 9 * (int64_t)cout_p;
    g.lda = cout_p; g.ldb = 9 * cin_p; g.ldc = cin_p; g.cin = cout_p; g.b_tap_stride = cin_p;
    g.splits = 1; g.kc_per_split = g.Kc; g.epi = epilogue & (VLG_CEPI_DPRELU | VLG_CEPI_ACCUM); g.act_ch = act_ch;
    if (tap_tables) { for (int t = 0; t < 9; ++t) g.shift[t] = 0; }
    else fill_shifts(g, wp, -1);
    hipStream_t s = (hipStream_t)stream;
    const int splits = (workspace != nullptr && da_slab == nullptr && tap_tables == nullptr) ? conv_dgrad_splits(rows_in, cin_p, cout_p) : 1;
    if (splits > 1) {
        if (!vlg_aligned16(workspace)) return VLG_ERR_ALIGN;
        if (workspace_capacity < (int64_t)splits * rows_in * cin_p) return VLG_ERR_SHAPE;
        const int ktiles = 9 * cout_p / 32;
        g.splits = splits;
        g.kc_per_split = (int64_t)((ktiles + splits - 1) / splits) * 32;
        g.slab_stride = rows_in * (int64_t)cin_p;
        g.C = workspace; g.rowmask = nullptr; g.epi = 0; g.aux_in = nullptr;       // raw partial tiles
        g.act_ch = cin_p;                                                          // (the finish kernel cuts the constant channels)
        if (int e = launch_conv<CONV_DGRAD, 128, 128>(g, s)) return e;
        const int64_t n4 = rows_in * (cin_p / 4);
        int64_t blocks = (n4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_finish_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, splits, g.slab_stride, x_in,
                           rowmask_in, prelu_slope ? 1.0f : 0.0f, prelu_slope, din, rows_in, cin_p / 4, act_ch,
                           epilogue & (VLG_CEPI_DPRELU | VLG_CEPI_ACCUM));
        return vlg_last_error();
    }
    const bool tail_ok = workspace != nullptr && da_slab == nullptr && tap_tables == nullptr && vlg_aligned16(workspace);
    const ConvTile t = conv_tile(rows_in, cin_p, cin_p, tail_ok ? g.Kc : 0);
    if (tail_ok) {
        const ConvTail tl = conv_tail_plan(rows_in, t, cin_p, g.Kc, cin_p);
        if (tl.tiles > 0 && tl.floats <= workspace_capacity) {
            g.tail_tiles = tl.tiles; g.tail_splits = tl.splits; g.tail_kc = tl.kc; g.tail_row0 = tl.row0;
            g.tail_stride = (rows_in - tl.row0) * (int64_t)cin_p; g.tail_ws = workspace;
            if (int e = launch_conv_tile<CONV_DGRAD>(t, g, s)) return e;
            const int64_t off = tl.row0 * (int64_t)cin_p, n4 = (rows_in - tl.row0) * (cin_p / 4);
            hipLaunchKernelGGL(conv_finish_dgrad_kernel, dim3((unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256)), dim3(256), 0, s,
                               workspace, tl.splits, g.tail_stride, x_in ? x_in + off : nullptr, rowmask_in ? rowmask_in + tl.row0 : nullptr,
                               prelu_slope ? 1.0f : 0.0f, prelu_slope, din + off, rows_in - tl.row0, cin_p / 4, act_ch,
                               epilogue & (VLG_CEPI_DPRELU | VLG_CEPI_ACCUM));
            return vlg_last_error();
        }
    }
    return launch_conv_tile<CONV_DGRAD>(t, g, s);
}

// row-tile height of the weight gradient: all of Cout for the GridNet widths (one pass over the gathered activation
// tile serves every output channel), 32-row tiles otherwise.  VLG_CONV_WGRAD_TALL=0 forces 32 (development switch).
static int conv_wgrad_bm(int cout_p) {
    static int tall = -1;
    if (tall < 0) { const char* e = getenv("VLG_CONV_WGRAD_TALL"); tall = e ? atoi(e) : 1; }
    return (tall && (cout_p == 64 || cout_p == 96)) ? cout_p : 32;
}
static void conv_wgrad_plan(int64_t rows, int cin_p, int cout_p, int* splits, int64_t* per) {
    const int64_t tiles = (cout_p / conv_wgrad_bm(cout_p)) * (int64_t)((9 * cin_p + 127) / 128);
    int64_t want = 512 / tiles;
    const int64_t max_splits = (rows + 255) / 256;
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int64_t p = (rows + want - 1) / want;
    p = (p + 31) / 32 * 32;
    *per = p;
    *splits = (int)((rows + p - 1) / p);
}

extern "C" int vlg_conv3x3_wgrad_slabs(int64_t rows, int cin_p, int cout_p) {
    int splits; int64_t per;
    conv_wgrad_plan(rows, cin_p, cout_p, &splits, &per);
    return splits;
}

extern "C" int vlg_conv3x3_wgrad(const float* dout, const float* in, float* slabs, int64_t slab_stride,
                                 int64_t slab_capacity, const int* rowtab, const float* prelu_slope, int64_t rows, int cin_p, int cout_p,
                                 int wp_in, int act_ch, void* stream) {
    // slab[s][co*(9*cin_p) + tap*cin_p + ci] = sum_{p in split s} dout[p, co] * act(in[row(p) + shift(tap), ci])
    // slab[s][cout_p*9*cin_p + co]           = sum_p dout[p, co]                (bias gradient)
    // rows past `rows` are read from the zero guard band (dout there is zero), so no edge handling is needed
    if (rows < 1 || cin_p < 32 || (cin_p & 31) || cout_p < 32 || (cout_p & 31)) return VLG_ERR_SHAPE;
    if (slab_stride < (int64_t)cout_p * 9 * cin_p + cout_p) return VLG_ERR_SHAPE;
    if (!conv_ok(dout) || !conv_ok(in) || !conv_ok(slabs)) return VLG_ERR_ALIGN;
    ConvArgs g{};
    g.A = dout; g.B = in; g.C = slabs; g.prelu = prelu_slope; g.rowtab = rowtab;
    g.M = cout_p; g.N = 9 * cin_p; g.Kc = rows;
    g.lda = cout_p; g.ldb = cin_p; g.ldc = 9 * cin_p; g.cin = cin_p;
    conv_wgrad_plan(rows, cin_p, cout_p, &g.splits, &g.kc_per_split);
    if (slab_capacity < (int64_t)g.splits * slab_stride) return VLG_ERR_SHAPE;      // the caller's buffer must hold every slab
    g.slab_stride = slab_stride; g.colsum_off = (int64_t)cout_p * 9 * cin_p; g.act_ch = act_ch;
    fill_shifts(g, wp_in, 1);
    switch (conv_wgrad_bm(cout_p)) {
        case 64: return launch_conv<CONV_WGRAD, 64, 128>(g, (hipStream_t)stream);
        case 96: return launch_conv<CONV_WGRAD, 96, 128>(g, (hipStream_t)stream);
        default: return launch_conv<CONV_WGRAD, 32, 128>(g, (hipStream_t)stream);
    }
}
