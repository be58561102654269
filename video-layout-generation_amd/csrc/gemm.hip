// fp32 MFMA GEMMs for the dense QKV / FFN / head projections.
//
// One kernel template serves forward (A.W^T), data-gradient (dY.W) and weight-gradient
// (dY^T.X, split over the token dimension) by choosing how each operand tile sits in LDS:
//   *_KC  tile[row][k]  - operand rows are contraction-contiguous in memory; row stride 36
//                         floats, so the per-lane float4 fragment reads (ds_read_b128) are
//                         bank-conflict-free (36*i mod 64 hits 16 distinct 4-bank slots)
//   *_MC  tile[k][row]  - operand is contraction-major in memory (W for dgrad, dY and X for
//                         wgrad); fragments are 4 ds_read_b32, lanes 0..31 on consecutive banks
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD, 157 TFLOP/s chip peak).  Its two
// k-slices per instruction go to lane halves h = lane>>5; we assign k = 8*s + 4*h + j to
// MFMA j of chunk s so ONE float4 read feeds four MFMAs (A and B use the same assignment,
// the contraction order inside a chunk is free).
// Block = 256 threads = 4 waves; 128x128 tile => each wave owns 64x64 = 2x2 MFMA tiles
// (64 accumulator VGPRs).  Staging is global -> VGPR -> LDS, double-buffered in LDS and one tile
// deep in registers: tile kt+1 is loaded from global one iteration ahead, written to the idle LDS
// buffer BETWEEN the MFMAs of one chunk of tile kt (one ds_write_b128 / load behind each MFMA, pinned
// with sched_group_barrier); fragment reads are double-buffered in registers (chunk s+1's LDS reads
// are issued before chunk s's 16 MFMAs); the iteration's one barrier sits in FRONT of the last
// chunk's MFMAs and the next tile's first fragments are read behind it, under those MFMAs.
// BK (contraction depth per tile) is a template parameter: 32 -> 73.7 KB LDS, 2 blocks / CU;
// 16 -> 41 KB, 3 blocks / CU.  blockIdx is remapped so tiles sharing an A panel sit on one XCD (L2).
//
// THE ONE PIPE.  On gfx950 this MFMA and the vector ALU of a SIMD execute serially, oldest wave first
// (tools/micro/mfma_f32_valu_share.hip): every vector instruction of ANY wave on the SIMD costs 8 cycles
// of matrix time (12 for a transcendental, packed or not), memory and LDS instructions cost none, and
// s_setprio changes nothing.  So the kernel has two paths:
//   fast     (128x128 tiles whose block lies inside both operands; mainloop_fast / emit_fast): no vector
//            instruction in the steady-state iteration - buffer loads with the K advance in the scalar
//            offset, two iterations unrolled so LDS addresses are immediates, zero tiles instead of
//            peeled tails; the bias in the accumulator start; buffer stores; GELU on packed
//            instructions in lockstep over 8 row pairs; optionally a RUN of N tiles per block, the K loop
//            continuing into the next tile.  151 TFLOP/s at K = 4096 (96 %), 115-143 per shape of the step.
//   general  (edge tiles, 32-wide tiles of the head, the GELU-on-load variants; mainloop / emit): 64-bit
//            pointers, clamps and selects - ~26 vector instructions per iteration, 141 TFLOP/s at best.
// Algorithmic work per launch: 2*M*N*K flop; bytes 4*(M*K + N*K + M*N) (+ aux operands).
#include "common.h"
#include <type_traits>
#include <stdlib.h>

#include "gemm_tile.h"
#include "reduce_body.h"

// keeps the hand-placed order "next chunk's LDS reads, then this chunk's MFMAs" (see ldfrag below); VLG_NO_SCHED_FENCE
// builds the loop without it (A/B switch for tools/kernel_bench.py)
// internal template bits on top of the public VLG_EPI_* ones: the operand named passes through GELU on its way to LDS
// (VLG_EPI_ACT_GELU: the FFN hidden activation is never stored - the second projection and its weight gradient
// recompute it from the pre-activation while they stage it, in the shadow of their own MFMAs)
#define GEMM_A_GELU (1 << 20)
#define GEMM_B_GELU (1 << 21)
__device__ __forceinline__ float4 gelu4(float4 v) { return make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w)); }

// raw buffer descriptor over [p, p + 2 GB): base in the descriptor, offsets in a VGPR (per thread) and an SGPR (per wave)
// (loads beyond `bytes` return zeros, stores beyond it are dropped)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vlg_rsrc(const void* p, int bytes = 0x7fffffff) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Epilogue pacing (compile-time option, OFF: measured neutral, tools/ab/gemm_ab.py).  The CU's vector-memory pipeline is one
// FIFO for all its waves and takes ~5 cycles per 128-B line (tools/micro/mfma_f32_valu_share.hip mem), so a 64 KB epilogue
// burst holds it for 1-2 us; sleeping between small groups of the epilogue's memory instructions (-DVLG_EPI_PACE=1) keeps the
// FIFO short for the co-resident block's tile loads - but that block's waves are not waiting on the FIFO: what they wait
// for is the vector ALU (oldest wave first), so nothing is gained.
#ifndef VLG_EPI_PACE
#define VLG_EPI_PACE 0          /* s_sleep argument (x 64 cycles) behind every VLG_EPI_PACE_EVERY memory instructions; 0 = off */
#endif
#ifndef VLG_EPI_PACE_EVERY
#define VLG_EPI_PACE_EVERY 2
#endif
__device__ __forceinline__ void vlg_epi_pace(int issued) {
#if VLG_EPI_PACE > 0
    if (issued % VLG_EPI_PACE_EVERY == 0) __builtin_amdgcn_s_sleep(VLG_EPI_PACE);
#endif
}

#ifdef VLG_NO_SCHED_FENCE
#define VLG_SCHED_FENCE()
#else
#define VLG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// PP ("ping-pong", chained BK = 32 launches only; a diagnostic option, OFF - measured 25-30 % slower than independent
// workgroups, cause not found): a workgroup is TWO groups of four waves, each computing its own run of
// tiles exactly as a 256-thread workgroup would (own LDS half, own virtual block id), but sharing the workgroup barrier -
// and running half a K range apart.  The vector ALU serves the oldest wave first, so two independent workgroups on a CU do
// not alternate: the older one streams its tiles, the younger one fills its gaps and is left alone with its own gaps
// (epilogue store issue, auxiliary loads) exposed.  With a common barrier per K tile the older group must wait for the
// younger every iteration, the two take the matrix pipe in turns, and one group's epilogue (cut into four slices, a
// barrier each) always runs beside the other group's main loop.  Both groups execute the same number of barriers: the
// second group starts with nk / 2 empty slots, the first ends with them.
// LDS floats of one four-wave group: both operand tiles, double-buffered
template <int BM, int BN, int BK, bool A_KC, bool B_KC>
constexpr int gemm_smem_floats() { return 2 * (Tile<BM, A_KC, BK>::FLOATS + Tile<BN, B_KC, BK>::FLOATS); }

// The kernel body as a device function of (arguments, LDS, block number, block count): gemm_f32_kernel runs one problem per
// launch, gemm_pair_kernel two independent ones (a projection's data gradient and weight gradient) side by side.
template <int BM, int BN, int BK, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool PP = false>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& g, float* const smem_all, const int blk, const int nblk) {
    // Raised priority until the main loop starts.  It does NOT get this block's vector instructions past an older block's
    // MFMA stream (the vector ALU serves the oldest wave that has a matrix or vector instruction ready, whatever s_setprio
    // says: tools/micro/mfma_f32_valu_share.hip prio / two), but the prologue's loads and LDS writes are issued ahead of the
    // neighbour's: measured 3-5 % per launch.
    __builtin_amdgcn_s_setprio(3);
    using EO = float;
    const float* const gA = static_cast<const float*>(g.A);
    const float* const gB = static_cast<const float*>(g.B);
    const float* const gAuxIn = static_cast<const float*>(g.aux_in);
    float* const gAuxOut = static_cast<float*>(g.aux_out);
    // 128x128: 2 x 2 waves of 64x64; 64x64 (launches whose 128-wide tiles would leave CUs empty: small M): 2 x 2 waves of 32x32
    constexpr int WM = (BM == BN && (BM == 128 || BM == 64)) ? 2 : (BM == 128 ? 4 : 1);
    constexpr int WN = 4 / WM;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int NCH = BK / 8;                    // 8-deep MFMA chunks per tile
    using TA = Tile<BM, A_KC, BK>;
    using TB = Tile<BN, B_KC, BK>;
    constexpr int SMEM_FLOATS = gemm_smem_floats<BM, BN, BK, A_KC, B_KC>();
    const int grp = PP ? (int)(threadIdx.x >> 8) : 0;          // ping-pong: which of the two four-wave groups
    float* const smem = smem_all + grp * SMEM_FLOATS;
    float* const As0 = smem;                       // As[buf] = As0 + buf * TA::FLOATS
    float* const Bs0 = smem + 2 * TA::FLOATS;      // Bs[buf] = Bs0 + buf * TB::FLOATS

    // XCD-aware remap: hardware deals blocks round-robin over 8 XCDs; give each XCD a
    // contiguous run of logical tiles (bijective for any grid size)
    // (ping-pong: the second group takes the upper half of the virtual block ids, so id & 7 still says which XCD)
    const int nwg = (PP ? 2 : 1) * nblk, bid = blk + grp * nblk;
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    // N tiles per block: the fast path of the BK = 32 kernels without bias-gradient sums (the host sets it); a compile-time 1
    // elsewhere, so those kernels have no loop around [main loop, epilogue]
    constexpr bool CHAIN = BM == 128 && BN == 128 && BK == 32 && !COLSUM && (EPI & ((1 << 20) | (1 << 21))) == 0;
    const int run = (CHAIN && g.run > 1) ? g.run : 1;
    const int tng = g.tiles_n / run;
    const int ntile = g.tiles_m * tng;
    const int split = swz / ntile;
    const int tile = swz - split * ntile;
    const int tm = tile / tng, tn = (tile - tm * tng) * run;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int64_t kbeg = (int64_t)split * g.kc_per_split;
    int64_t kend = kbeg + g.kc_per_split;
    if (kend > g.Kc) kend = g.Kc;

    const int tid = threadIdx.x & (GEMM_THREADS - 1), lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave - wm * WN;
#ifdef VLG_TIMELINE
    // diagnostic build only (tools/diag/gemm_timeline.py): 100 MHz stamps at entry / loop start / loop end / exit + placement
    const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();
    unsigned long long tl_loop0 = 0, tl_loop1 = 0;
#endif

    f32x16 acc[TM][TN];
    // bias gradient (COLSUM): every thread sums the four dY rows 4*(tid % PER_ROW) .. +3 of the tiles it stages, straight
    // from its staging registers (no LDS reads, no serial chain on two of the four waves); combined through LDS at the end
    // (two packed adds per float4, in every block: a block-uniform branch around them would cut the hand-scheduled
    // iteration into several basic blocks; only the tn == 0 blocks write their sums out)
    v2f col_lo = v2(0.f), col_hi = v2(0.f);
    auto colsum = [&](const float4 (&xa)[TA::NV]) {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {
            col_lo += v2f{xa[i].x, xa[i].y};
            col_hi += v2f{xa[i].z, xa[i].w};
        }
    };
    // the accumulators START from the bias (a lane's 16 accumulators of a 32x32 tile share one column): no vector
    // instruction is spent on it in the epilogue, where every one of them costs matrix time (see common.h, gelu2)
    float bv[TN];
    auto load_bias = [&](int n0v) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bv[j] = 0.f;
            if constexpr ((EPI & VLG_EPI_BIAS) != 0) {
                const int col = n0v + (wn * TN + j) * 32 + l31;
                bv[j] = g.bias[col < g.N ? col : g.N - 1];
            }
        }
    };
    auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = bv[j];
    };
    load_bias(n0);
    init_acc();

#ifndef VLG_TIMELINE
    unsigned long long t0 = 0, r0 = 0;
    if (g.clock_probe) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    float4 ra[TA::NV], rb[TB::NV];
    const int nk = (int)((kend - kbeg + BK - 1) / BK);
    bool fast_tile = false;
    auto act = [&](float4 (&xa)[TA::NV], float4 (&xb)[TB::NV]) {
        if constexpr ((EPI & GEMM_A_GELU) != 0) {
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) xa[i] = gelu4(xa[i]);
        }
        if constexpr ((EPI & GEMM_B_GELU) != 0) {
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) xb[i] = gelu4(xb[i]);
        }
    };
    // Fragment reads are software-pipelined by hand: the reads of chunk s+1 are issued BEFORE the 16 MFMAs of chunk s, into a
    // second register set, so a wave never waits on LDS between MFMAs.  Left to itself hipcc re-reads two registers at
    // a time with an s_waitcnt lgkmcnt(0) in front of every 2-4 MFMAs: each wave then idles ~30 % of the time and the
    // two waves of a SIMD idle together often enough to cost ~10 % of the matrix pipe.
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
    // One iteration: MFMAs of the tile in LDS buffer `cur`; in the middle, the register-resident tile kt+1 goes to the
    // other LDS buffer and the registers are refilled with tile kt+DEPTH (DEPTH tiles travel global -> register at once).
    auto mainloop = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
        constexpr int DEPTH = 1;                 // tiles travelling global -> register at once
        auto load = [&](float4 (&xa)[TA::NV], float4 (&xb)[TB::NV], int t) {
            const int64_t k0 = kbeg + (int64_t)t * BK;
            TA::template gload<GUARD>(xa, gA, g.lda, m0, g.M, k0, kend, tid);
            TB::template gload<GUARD>(xb, gB, g.ldb, n0, g.N, k0, kend, tid);
        };
        // STORE / LOAD / NEXT are compile-time so the steady-state iteration is ONE basic block: the compiler can then spread
        // the staging instructions (8 ds_write_b128, 8 global_load_dwordx4, address arithmetic) between the MFMAs instead
        // of issuing them as a blob with the matrix pipe idle; the last two iterations are peeled.
        // The iteration's ONE barrier sits in front of the LAST chunk's MFMAs, not behind them: by then every wave has
        // its last fragments of this tile in registers and has finished writing the next tile (staged during chunk SS), so
        // behind the barrier the next tile's first fragments are read while the last chunk's 16 MFMAs run - a wave
        // leaves the barrier with matrix work in hand instead of an LDS round trip.
        constexpr int SS = NCH / 2 - 1;          // chunk whose MFMAs cover the staging traffic
        float fa[2][TM][4], fb[2][TN][4];        // fragment sets: chunk s lives in set s & 1 (NCH is even)
        auto iter = [&](int kt, float4 (&xa)[TA::NV], float4 (&xb)[TB::NV], auto store_tag, auto load_tag, auto next_tag) {
            constexpr bool STORE = decltype(store_tag)::value, LOAD = decltype(load_tag)::value, NEXT = decltype(next_tag)::value;
            const int cur = kt & 1;
            const float* as = As0 + cur * TA::FLOATS;
            const float* bs = Bs0 + cur * TB::FLOATS;
#pragma unroll
            for (int s = 0; s < NCH; ++s) {
                if (s + 1 < NCH) ldfrag(fa[(s + 1) & 1], fb[(s + 1) & 1], as, bs, s + 1);
                if (s == NCH - 1 && NEXT) {
                    __syncthreads();
                    ldfrag(fa[0], fb[0], As0 + (cur ^ 1) * TA::FLOATS, Bs0 + (cur ^ 1) * TB::FLOATS, 0);
                }
                VLG_SCHED_FENCE();                 // reads above, MFMAs below; the staging code may mix with the MFMAs
                if (s == SS) {
                    if constexpr (STORE) {
                        if constexpr (COLSUM) colsum(xa);
                        act(xa, xb);
                        TA::sstore(xa, As0 + (cur ^ 1) * TA::FLOATS, tid);
                        TB::sstore(xb, Bs0 + (cur ^ 1) * TB::FLOATS, tid);
                    }
                    if constexpr (LOAD) load(xa, xb, kt + 1 + DEPTH);
                }
                mma(fa[s & 1], fb[s & 1]);
                if (s == SS && (STORE || LOAD)) {
                    // pin the interleave: one LDS write (then one global load) behind each MFMA of this chunk, so the
                    // matrix pipe keeps issuing while the tile is staged (hipcc left alone emits the writes as one blob)
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
        static_assert(NCH >= 2 && NCH % 2 == 0, "fragment sets alternate by chunk parity");
        if (nk > 0) {
            load(ra, rb, 0);
            if constexpr (COLSUM) colsum(ra);
            act(ra, rb);
            TA::sstore(ra, As0, tid);
            TB::sstore(rb, Bs0, tid);
        }
        if (nk > 1) load(ra, rb, 1);
        __syncthreads();
        if (nk > 0) ldfrag(fa[0], fb[0], As0, Bs0, 0);
        __builtin_amdgcn_s_setprio(0);
        int kt = 0;
        for (; kt + 2 < nk; ++kt) iter(kt, ra, rb, std::true_type{}, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { iter(kt, ra, rb, std::true_type{}, std::false_type{}, std::true_type{}); ++kt; }
        if (kt < nk) iter(kt, ra, rb, std::false_type{}, std::false_type{}, std::false_type{});
        __syncthreads();                           // the tiles are dead from here on (the COLSUM epilogue reuses the LDS)
    };
    // ---- fast path (round 2, second half): NO vector-ALU instruction in the steady-state iteration.
    // v_mfma_f32_32x32x2_f32 and the vector ALU of a SIMD execute serially (tools/micro/mfma_f32_valu_share.hip: every
    // vector instruction of any wave on the SIMD costs 8 cycles of matrix time), and the loop above spent ~26 of them per
    // iteration on addresses (64-bit global pointers, LDS addresses that depend on the buffer parity): 6-8 % of the
    // matrix pipe.  Here the global loads are buffer loads - one loop-constant byte offset per thread and operand, the K
    // advance and the float4 number in the SCALAR offset - and two iterations are unrolled so the LDS buffer is a compile-
    // time constant and every LDS address is one loop-constant register plus an instruction immediate.
    constexpr bool FAST = BM == BN && (BM == 128 || BM == 64) && (EPI & (GEMM_A_GELU | GEMM_B_GELU)) == 0;
    // A block may compute `run` consecutive N tiles of one row panel back to back (multi-round launches): the K loop
    // then simply CONTINUES into the next tile - the loads its last two iterations issue are the next tile's first two K
    // tiles, the LDS write of its last iteration is the next tile's tile 0 and the fragments read behind its last
    // barrier are the next tile's first - so only the epilogue separates the MFMA streams of two tiles.  (A block that
    // starts while an older one streams MFMAs cannot do that: the vector ALU serves the MFMA stream first - s_setprio
    // does not change it, tools/micro/mfma_f32_valu_share.hip prio - so its index arithmetic, and with it its first loads,
    // wait for the neighbour's loop end: per-CU timelines show 1.3-3.4 us of idle matrix pipe per block.)
#ifndef VLG_GEMM_DEPTH
#define VLG_GEMM_DEPTH 1     /* 2 measured: no gain (tools/ab/gemm_ab.py), +16..32 registers */
#endif
    constexpr int DEPTH = VLG_GEMM_DEPTH;
    v4f xas[DEPTH][TA::NV], xbs[DEPTH][TB::NV];      // staging registers and fragment sets live across the tiles of a run
    float ffa[2][TM][4], ffb[2][TN][4];
    auto mainloop_fast = [&](int n0v, bool first, bool has_next) __attribute__((always_inline)) {
        const float* pa = A_KC ? gA + m0 * g.lda + kbeg : gA + kbeg * g.lda + m0;
        const float* pb = B_KC ? gB + (int64_t)n0v * g.ldb + kbeg : gB + kbeg * g.ldb + n0v;
        // a contraction-major operand's descriptor ends with the K range of this block: loads beyond it return zeros
        const int ext_b = (int)(kend - kbeg);
        const __amdgpu_buffer_rsrc_t da = A_KC ? vlg_rsrc(pa) : vlg_rsrc(pa, ext_b * g.lda * 4);
        const __amdgpu_buffer_rsrc_t db = B_KC ? vlg_rsrc(pb) : vlg_rsrc(pb, ext_b * g.ldb * 4);
        const __amdgpu_buffer_rsrc_t dza = vlg_rsrc(pa, 0), dzb = vlg_rsrc(pb, 0);      // empty: every load returns zeros
        const int va = TA::voff_bytes(g.lda, tid), vb = TB::voff_bytes(g.ldb, tid);
        const int a_is = TA::ISTEP * g.lda * 4, b_is = TB::ISTEP * g.ldb * 4;              // bytes between a thread's float4
        const int a_ks = A_KC ? BK * 4 : BK * g.lda * 4, b_ks = B_KC ? BK * 4 : BK * g.ldb * 4;   // bytes per K tile
        const int b_next = B_KC ? BN * g.ldb * 4 : BN * 4;                                  // bytes to the next N tile of B
        float* const aw = smem + TA::soff(tid);
        float* const bw = smem + 2 * TA::FLOATS + TB::soff(tid);
        // one read base per 32-row group: every fragment offset of a contraction-major tile is then a multiple of 256 B below
        // 64 KB, which ds_read2st64_b32 encodes (two bases short, hipcc re-derives addresses with v_add_u32 inside the loop)
        const float* ar[TM];
        const float* br[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) ar[i] = smem + TA::roff((wm * TM + i) * 32 + l31, h);
#pragma unroll
        for (int j = 0; j < TN; ++j) br[j] = smem + 2 * TA::FLOATS + TB::roff((wn * TN + j) * 32 + l31, h);
        auto load = [&](int t, int set) __attribute__((always_inline)) {
            v4f (&xa)[TA::NV] = xas[set];
            v4f (&xb)[TB::NV] = xbs[set];
            // K tiles past the end: the NEXT tile of the run (same A panel from its start, B one N tile further), or - no next
            // tile - zeros: through the empty descriptor (K-contiguous operand) or past the end of the operand's own
            // descriptor (contraction-major operand); an odd nk runs one iteration on such a zero tile
            const bool in = t < nk, wrap = !in && has_next;
            const int tw = wrap ? t - nk : t;
            const __amdgpu_buffer_rsrc_t ua = (A_KC && !in && !wrap) ? dza : da, ub = (B_KC && !in && !wrap) ? dzb : db;
            const int oa = tw * a_ks, ob = tw * b_ks + (wrap ? b_next : 0);
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) xa[i] = __builtin_amdgcn_raw_buffer_load_b128(ua, va, oa + i * a_is, 0);
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) xb[i] = __builtin_amdgcn_raw_buffer_load_b128(ub, vb, ob + i * b_is, 0);
        };
        auto store = [&](int c, int set) __attribute__((always_inline)) {
            v4f (&xa)[TA::NV] = xas[set];
            v4f (&xb)[TB::NV] = xbs[set];
            if constexpr (COLSUM) {
#pragma unroll
                for (int i = 0; i < TA::NV; ++i) {
                    // (as asm: hipcc merges the two halves into one 4-wide add and then splits THAT into four v_add_f32)
                    const v2f lo = __builtin_shufflevector(xa[i], xa[i], 0, 1), hi = __builtin_shufflevector(xa[i], xa[i], 2, 3);
                    asm("v_pk_add_f32 %0, %0, %1" : "+v"(col_lo) : "v"(lo));
                    asm("v_pk_add_f32 %0, %0, %1" : "+v"(col_hi) : "v"(hi));
                }
            }
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) *reinterpret_cast<v4f*>(aw + c * TA::FLOATS + i * TA::SSTEP) = xa[i];
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) *reinterpret_cast<v4f*>(bw + c * TB::FLOATS + i * TB::SSTEP) = xb[i];
        };
        auto ldf = [&](float (&a)[TM][4], float (&b)[TN][4], int c, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) TA::frag_at(a[i], ar[i] + c * TA::FLOATS, 0, s);
#pragma unroll
            for (int j = 0; j < TN; ++j) TB::frag_at(b[j], br[j] + c * TB::FLOATS, 0, s);
        };
        constexpr int SS = NCH / 2 - 1;
        // cur is a literal at every call site (the lambda is always inlined), so cur * FLOATS folds into the immediates.
        // There are NO peeled tail iterations: the last two iterations run the same code - their loads fetch the next tile of
        // the run (or zeros), their LDS writes go to the buffer nobody reads any more (or hold the next tile's first K
        // tile) and the fragments read behind the last barrier are dropped (or are the next tile's first).  One copy of the
        // iteration per buffer parity keeps the code small and the register allocation tight (seven inlined tail copies
        // cost 60+ registers and spills).
        auto iter = [&](int kt, int cur) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NCH; ++s) {
                if (s + 1 < NCH) ldf(ffa[(s + 1) & 1], ffb[(s + 1) & 1], cur, s + 1);
                if (s == NCH - 1) {
                    __syncthreads();
                    ldf(ffa[0], ffb[0], cur ^ 1, 0);
                }
                VLG_SCHED_FENCE();
                if (s == SS) {                               // tile kt + 1 -> LDS, its registers take tile kt + 1 + DEPTH
                    const int set = DEPTH == 2 ? (cur ^ 1) : 0;
                    store(cur ^ 1, set);
                    load(kt + 1 + DEPTH, set);
                }
                mma(ffa[s & 1], ffb[s & 1]);
                if (s == SS) {
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
        if (first) {
            load(0, 0);
            store(0, 0);
            // (fenced: hipcc's wait-count pass merges the prologue's load order into the loop's; loads reordered HERE make it
            // wait for the youngest tile inside the loop)
            VLG_SCHED_FENCE();
            if constexpr (DEPTH == 2) { load(1, 1); VLG_SCHED_FENCE(); load(2, 0); }
            else load(1, 0);
            VLG_SCHED_FENCE();
            __syncthreads();
            ldf(ffa[0], ffb[0], 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        for (int kt = 0; kt < nk; kt += 2) { iter(kt, 0); iter(kt + 1, 1); }
        __builtin_amdgcn_s_setprio(2);
    };
    // interior blocks (every tile fully inside both operands) take the unguarded instantiation
    const bool interior = (m0 + BM <= g.M) && (n0 + run * BN <= g.N) && (((kend - kbeg) % BK) == 0);
#ifdef VLG_TIMELINE
    tl_loop0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (FAST) {
        // byte offsets from the tile origins are 32-bit in the fast path
        const int64_t ext = kend - kbeg;
        const int64_t span_a = A_KC ? (int64_t)(BM - 1) * g.lda + ext : ext * g.lda + BM;
        const int64_t span_b = B_KC ? (int64_t)(2 * BN - 1) * g.ldb + ext : ext * g.ldb + 2 * BN;   // (+ the next N tile of a run)
        const bool fits = span_a < (1ll << 28) && span_b < (1ll << 28) && (int64_t)(BM - 1) * g.ldc + BN < (1ll << 28);
        fast_tile = interior && fits;
        if (!fast_tile) mainloop(std::true_type{});
    } else {
        if (interior) mainloop(std::false_type{});
        else mainloop(std::true_type{});
    }
#ifdef VLG_TIMELINE
    tl_loop1 = __builtin_amdgcn_s_memrealtime();
#endif

#ifndef VLG_TIMELINE
    if (g.clock_probe && tid == 0) {
        g.clock_probe[2 * bid] = __builtin_amdgcn_s_memtime() - t0;
        g.clock_probe[2 * bid + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h.
    // 32 lanes of a half write one 128-B row segment per store.
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
                    float v = acc[i][j][r];
                    if constexpr ((EPI & VLG_EPI_GELU) != 0) {
                        if constexpr ((EPI & VLG_EPI_GELU_GRAD) != 0) {
                            float c, pd;
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
    // fast epilogue: no address arithmetic on the vector ALU (buffer stores: one loop-constant byte offset per thread, the
    // row in the scalar offset, the 32-column group in the immediate), GELU / dGELU on packed instructions, two rows at a time
    auto emit_fast = [&](int n0v) __attribute__((always_inline)) {
        const int64_t corner = m0 * g.ldc + n0v;
        const __amdgpu_buffer_rsrc_t dc = vlg_rsrc(Cs + corner);
        const __amdgpu_buffer_rsrc_t dxin = vlg_rsrc(gAuxIn ? gAuxIn + corner : gA);
        const __amdgpu_buffer_rsrc_t dxout = vlg_rsrc(gAuxOut ? gAuxOut + corner : Cs + corner);
        const int vc = ((wm * TM * 32 + 4 * h) * g.ldc + wn * TN * 32 + l31) * 4;
        const int rowb = g.ldc * 4;
        constexpr bool AUX = (EPI & (VLG_EPI_RESID | VLG_EPI_DGELU | VLG_EPI_MUL)) != 0;
        // the auxiliary operand: ALL of it is requested before the first store where the registers allow (BK = 32: two blocks
        // per CU, 256 registers) - loads and stores retire through one in-order counter, so a load issued behind a tile's
        // stores is not usable before those stores have landed; with three blocks per CU (BK = 16) one tile ahead
        constexpr bool AUX_ALL = BK == 32;
        constexpr int NAUX = AUX_ALL ? TM * TN : 2;
        float auxb[NAUX][16];
        auto fetch = [&](float (&a)[16], int i, int j) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dxin, vc + j * 128, (i * 32 + (r & 3) + 8 * (r >> 2)) * rowb, 0));
                vlg_epi_pace(r + 1);
            }
        };
        if constexpr (AUX) {
            if constexpr (AUX_ALL) {
#pragma unroll
                for (int t = 0; t < TM * TN; ++t) fetch(auxb[t], t / TN, t % TN);
            } else {
                fetch(auxb[0], 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int tix = i * TN + j;
                float (&aux)[16] = auxb[AUX_ALL ? tix : (tix & 1)];
                if constexpr (AUX && !AUX_ALL) {
                    if (tix + 1 < TM * TN) fetch(auxb[(tix + 1) & 1], (tix + 1) / TN, (tix + 1) % TN);
                }
                const auto soff = [&](int r) { return (i * 32 + (r & 3) + 8 * (r >> 2)) * rowb; };       // row of register r
                v2f v[8];                                                // the tile's 16 registers as 8 row pairs (r, r + 1)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = v2f{acc[i][j][2 * q], acc[i][j][2 * q + 1]};
                if constexpr ((EPI & VLG_EPI_GELU) != 0 && (EPI & VLG_EPI_GELU_GRAD) != 0) {
                    v2f dv[8];
                    gelu_both2n<8>(v, dv);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv[q].x), dxout, vc + j * 128, soff(2 * q), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv[q].y), dxout, vc + j * 128, soff(2 * q + 1), 0);
                        vlg_epi_pace(2 * q + 2);
                    }
                } else if constexpr ((EPI & VLG_EPI_GELU) != 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].x), dxout, vc + j * 128, soff(2 * q), 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].y), dxout, vc + j * 128, soff(2 * q + 1), 0);
                        vlg_epi_pace(2 * q + 2);
                    }
                    gelu2n<8>(v);
                }
                if constexpr ((EPI & VLG_EPI_RESID) != 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += v2f{aux[2 * q], aux[2 * q + 1]};
                }
                if constexpr ((EPI & VLG_EPI_MUL) != 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= v2f{aux[2 * q], aux[2 * q + 1]};
                }
                if constexpr ((EPI & VLG_EPI_DGELU) != 0) {
                    v2f u[8], d[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) u[q] = v2f{aux[2 * q], aux[2 * q + 1]};
                    dgelu2n<8>(u, d);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= d[q];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].x), dc, vc + j * 128, soff(2 * q), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q].y), dc, vc + j * 128, soff(2 * q + 1), 0);
                    vlg_epi_pace(2 * q + 2);
                }
                if constexpr (PP) __syncthreads();          // one epilogue slice per slot of the other group's main loop
            }
    };
    if constexpr (PP) {
        if (!fast_tile) return;                    // (never: the host launches this variant only where every block is fast)
        if (grp == 1) for (int i = 0; i < nk / 2; ++i) __syncthreads();
    }
    if (fast_tile) {
        // the tiles of the run: [main loop, epilogue] per tile; the next tile's bias is fetched ahead of the main loop
        for (int rt = 0; rt < run; ++rt) {
            const int n0v = n0 + rt * BN;
            const bool has_next = rt + 1 < run;
            mainloop_fast(n0v, rt == 0, has_next);
            if (has_next) load_bias(n0v + BN);
            emit_fast(n0v);
            if (has_next) init_acc();
        }
        if constexpr (PP) {
            if (grp == 0) for (int i = 0; i < nk / 2; ++i) __syncthreads();
        }
    } else if (full) emit(std::false_type{});
    else emit(std::true_type{});
#ifdef VLG_TIMELINE
    if (g.clock_probe) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tl_exit = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* o = g.clock_probe + 8 * (size_t)bid;
            o[0] = tl_entry; o[1] = tl_loop0; o[2] = tl_loop1; o[3] = tl_exit;
            o[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));        // HW_REG_HW_ID
            o[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));       // HW_REG_XCC_ID
            o[6] = (unsigned long long)tile; o[7] = (unsigned long long)split;
        }
    }
#endif
    if constexpr (COLSUM) {
        static_assert(!A_KC, "column sums are taken from a contraction-major A tile");
        if (tn == 0) {                                  // block-uniform; the tiles are dead (the main loop ended on a barrier)
            constexpr int PR = TA::PER_ROW, NG = GEMM_THREADS / PR;       // threads per k-row, groups sharing the same rows
            float* red = smem;
            st4(red + (tid / PR) * BM + 4 * (tid % PR), make_float4(col_lo.x, col_lo.y, col_hi.x, col_hi.y));
            __syncthreads();
            if (tid < BM && m0 + tid < g.M) {
                float s = 0.f;
#pragma unroll 8
                for (int t = 0; t < NG; ++t) s += red[t * BM + tid];
                st1(Cs + g.colsum_off + m0 + tid, s);
            }
        }
    }
}

template <int BM, int BN, int BK, bool A_KC, bool B_KC, int EPI, bool COLSUM, bool PP = false>
__global__ __launch_bounds__(PP ? 2 * GEMM_THREADS : GEMM_THREADS, (BM == 64 && BN == 64) ? 4 : (BK == 16 ? 3 : 2)) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem_all[(PP ? 2 : 1) * gemm_smem_floats<BM, BN, BK, A_KC, B_KC>()];
    gemm_f32_body<BM, BN, BK, A_KC, B_KC, EPI, COLSUM, PP>(g, smem_all, (int)blockIdx.x, (int)gridDim.x);
}

// A projection's data gradient dX = dY . W and weight gradient dW = dY^T . X need nothing from each other: ONE launch runs
// both (blocks [0, nd_pad) the data gradient - nd of them real, padded to a multiple of 8 so that a block's number mod 8
// still names its XCD for the second problem - the rest the weight gradient).  Each problem's blocks execute exactly the code
// of its own launch (same tiles, same split plan, same sums: bit for bit the two launches), but the chip is dealt both at
// once: launches that cannot fill 256 CUs on their own (few tokens per GPU: the strong-scaling shard) share it, and one
// launch ramp / drain is paid instead of two.
// Blocks past both problems are RIDERS: rows of a slab-reduction table (the previous gradient bucket's partial sums, complete
// since the launches that wrote them precede this one on the stream; this launch's weight gradient writes the OTHER arena).
// The reduction is pure bandwidth work on 256-thread blocks that need 4 KB of LDS: it runs in the slots the GEMM blocks leave
// towards the end of the launch instead of as a launch of its own.
template <int BT, int EPI_D>
__global__ __launch_bounds__(GEMM_THREADS, BT == 64 ? 4 : 2) void gemm_pair_kernel(const GemmArgs gd, const GemmArgs gw, const int nd, const int nd_pad,
                                                                                   const int nw, const int64_t* __restrict__ rider, const int rider_bpr) {
    constexpr int FD = gemm_smem_floats<BT, BT, 32, true, false>(), FW = gemm_smem_floats<BT, BT, 32, false, false>();
    __shared__ __attribute__((aligned(16))) float smem_all[FD > FW ? FD : FW];
    const int b = (int)blockIdx.x;
    if (b < nd_pad) {
        if (b < nd) gemm_f32_body<BT, BT, 32, true, false, EPI_D, false>(gd, smem_all, b, nd);
    } else if (b < nd_pad + nw) {
        gemm_f32_body<BT, BT, 32, false, false, VLG_EPI_NONE, true>(gw, smem_all, b - nd_pad, nw);
    } else {
        const int rb = b - nd_pad - nw;
        reduce_table_row(rider, rb / rider_bpr, rb % rider_bpr, rider_bpr, reinterpret_cast<float4(*)[16]>(smem_all));
    }
}

// The library keeps NO mutable process-wide state (include/vlg_hip.h).  The development switches below exist only in the
// diagnostic build (`make diag` -> libvlg_hip_diag.so, -DVLG_DIAG; loaded by tools/diag, tools/ab through VLG_HIP_LIB); in
// the product build they are constants and the vlg_debug_set_* entry points do not exist.
#ifdef VLG_DIAG
// a device buffer of 2 * blocks uint64 set through vlg_debug_set_clock_probe (NULL = off)
static unsigned long long* vlg_gemm_clock_probe = nullptr;
extern "C" void vlg_debug_set_clock_probe(unsigned long long* p) { vlg_gemm_clock_probe = p; }
#else
static constexpr unsigned long long* vlg_gemm_clock_probe = nullptr;
#endif

// Contraction depth per tile.  Measured on MI355X at the metric shape (tools/kernel_bench.py):
// BK = 32 (2 blocks / CU) is 3-5 % faster for plain epilogues, BK = 16 (41 KB LDS, 4 blocks / CU)
// is 8-11 % faster when the epilogue is heavy (GELU / dGELU: two extra 134 MB streams), because more
// resident blocks de-synchronise the store bursts from the other blocks' MFMA phases.
// VLG_GEMM_BK=16|32 forces one value for A/B runs.
#ifdef VLG_DIAG
static int vlg_gemm_bk_forced = -1;
static int gemm_bk_override() {
    if (vlg_gemm_bk_forced < 0) {
        const char* e = getenv("VLG_GEMM_BK");
        vlg_gemm_bk_forced = e ? atoi(e) : 0;
    }
    return vlg_gemm_bk_forced;
}
extern "C" void vlg_debug_set_gemm_bk(int bk) { vlg_gemm_bk_forced = (bk == 16 || bk == 32) ? bk : 0; }
#else
static int gemm_bk_override() {                  // the environment variable, read once: configuration, not state
    static const int forced = [] { const char* e = getenv("VLG_GEMM_BK"); return e ? atoi(e) : 0; }();
    return forced;
}
#endif

// Consecutive N tiles per block (GemmArgs::run) for a launch that would otherwise take several rounds of blocks: the
// largest divisor of the N tile count that still leaves one block per slot.  Only where EVERY block takes the fast path
// (no edge tiles, an even number of K tiles, 32-bit spans): a block that does not computes one tile only.
#ifdef VLG_DIAG
static int vlg_gemm_run_forced = -1;
static int vlg_gemm_pingpong = 0;
// run: 0 = never chain, -1 = the library's choice, > 0 = that many; bit 16 set: chained launches as ping-pong pairs (two
// four-wave groups per workgroup, see the kernel) instead of independent 256-thread workgroups - MEASURED 25-30 % SLOWER
// (tools/ab/gemm_ab.py: fwd qkv 95 -> 120 us, FFN1 + GELU 139 -> 188 us), so it stays a diagnostic switch
extern "C" void vlg_debug_set_gemm_run(int run) {
    vlg_gemm_pingpong = (run >= 0 && (run & 0x10000)) ? 1 : 0;
    vlg_gemm_run_forced = run < 0 ? -1 : (run & 0xffff) == 0xffff ? -1 : (run & 0xffff);
}
#else
static constexpr int vlg_gemm_run_forced = -1;
#endif
template <int BM, int BN, int BK, bool A_KC, bool B_KC>
static int gemm_run(const GemmArgs& g, int slots) {
    if (BM != 128 || BN != 128 || g.splits != 1) return 1;
    if (g.M % BM != 0 || g.N % BN != 0 || g.Kc % (2 * BK) != 0) return 1;
    const int64_t span_a = A_KC ? (int64_t)(BM - 1) * g.lda + g.Kc : g.Kc * g.lda + BM;
    const int64_t span_b = B_KC ? (int64_t)(2 * BN - 1) * g.ldb + g.Kc : g.Kc * g.ldb + 2 * BN;
    if (span_a >= (1ll << 28) || span_b >= (1ll << 28) || (int64_t)(BM - 1) * g.ldc + BN >= (1ll << 28)) return 1;
    if (vlg_gemm_run_forced == 0) return 1;
    int best = 1;
    for (int r = 2; r <= g.tiles_n; ++r)
        if (g.tiles_n % r == 0 && (int64_t)g.tiles_m * (g.tiles_n / r) >= slots) best = r;
    if (vlg_gemm_run_forced > 0 && g.tiles_n % vlg_gemm_run_forced == 0) best = vlg_gemm_run_forced;
    return best;
}

// 64x64 tiles (four 32x32 waves, four blocks per CU) for launches whose 128x128 tiles would leave CUs without a block: the
// per-GPU share of a GLOBAL batch (reference src/trainer.py:148: 32 // 8 = 4 clips, M = 4 096 tokens) gives the N = 256
// products 64 tiles for 256 CUs.  VLG_GEMM_SMALL=0 (read once) keeps 128x128 everywhere (A/B runs).
static bool gemm_small_tiles() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("VLG_GEMM_SMALL");
        on = (e && atoi(e) == 0) ? 0 : 1;
    }
    return on == 1;
}
// Measured (tools/shard_bench.py, interleaved repeated runs on one box): 64x64 tiles pay while the 128x128 tiles would not fill
// the 512 block slots of the chip (2 per CU) - four resident 64x64 blocks overlap one block's prologue / epilogue with the
// others' MFMAs: B = 4: 1.75 -> 1.05 ms, B = 8: 1.79 -> 1.76, B = 16: 3.11 -> 3.05.  A launch of exactly 512 blocks (the d x d
// products of the headline step) is FASTER on 128x128 tiles (their higher arithmetic intensity: K = 4096 asymptote 150 vs 140
// TFLOP/s; the step 5.55 vs 5.63 ms with a threshold of 1 025 - a first single-run sweep had suggested the opposite).
static int gemm_small_below() {                  // VLG_GEMM_SMALL_BELOW (read once): 128x128 block count under which 64x64 tiles are taken
    static int thr = -1;
    if (thr < 0) {
        const char* e = getenv("VLG_GEMM_SMALL_BELOW");
        thr = e ? atoi(e) : 512;
    }
    return thr;
}
static int gemm_small_below_wgrad() {            // the same for the weight-gradient plan (VLG_GEMM_SMALL_BELOW_WGRAD)
    static int thr = -1;
    if (thr < 0) {
        const char* e = getenv("VLG_GEMM_SMALL_BELOW_WGRAD");
        thr = e ? atoi(e) : 385;                     // under 3/4 of the 512 slots of the 128x128 plan (B = 32 keeps that plan: 504-512 blocks)
    }
    return thr;
}
static bool gemm_wants_small(int64_t M, int N, int splits) {
    return gemm_small_tiles() && ((M + 127) / 128) * (int64_t)((N + 127) / 128) * splits < gemm_small_below();
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, bool COLSUM>
static int launch_gemm(GemmArgs g, hipStream_t s) {
    if constexpr (BM == 128 && BN == 128 && !COLSUM && (EPI & (GEMM_A_GELU | GEMM_B_GELU)) == 0) {
        if (gemm_wants_small(g.M, g.N, g.splits)) return launch_gemm<64, 64, A_KC, B_KC, EPI, COLSUM>(g, s);
    }
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (g.N + BN - 1) / BN;
    g.run = 1;
    g.clock_probe = vlg_gemm_clock_probe;
    const dim3 block(GEMM_THREADS);
    constexpr bool CAN_RUN = BM == 128 && BN == 128 && !COLSUM && (EPI & (GEMM_A_GELU | GEMM_B_GELU)) == 0;
    if constexpr (BM == 128 && BN == 128) {
        const int forced = gemm_bk_override();
        const bool heavy_epilogue = (EPI & (VLG_EPI_GELU | VLG_EPI_DGELU)) != 0;
        int run32 = 1;
        if constexpr (CAN_RUN) run32 = gemm_run<BM, BN, 32, A_KC, B_KC>(g, 512);
        // heavy epilogues: three blocks per CU (BK = 16) hide more of the store phase - unless the tiles chain (run > 1)
        if (forced == 16 || (forced != 32 && heavy_epilogue && run32 == 1)) {
            const int64_t blocks = (int64_t)g.tiles_m * (g.tiles_n / g.run) * g.splits;
            if (blocks < 1 || blocks > 0x7fffffff) return VLG_ERR_SHAPE;
            hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 16, A_KC, B_KC, EPI, COLSUM>), dim3((unsigned)blocks), block, 0, s, g);
            return vlg_last_error();
        }
        g.run = run32;
    }
    const int64_t blocks = (int64_t)g.tiles_m * (g.tiles_n / g.run) * g.splits;
    if (blocks < 1 || blocks > 0x7fffffff) return VLG_ERR_SHAPE;
    if constexpr (CAN_RUN) {
        // chained launches: two four-wave groups per workgroup taking the matrix pipe in turns (see the kernel); the virtual
        // block ids of a group must keep their XCD (multiple of 8 workgroups)
#ifdef VLG_DIAG
        if (g.run > 1 && vlg_gemm_pingpong && (blocks % 16) == 0) {
            hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 32, A_KC, B_KC, EPI, COLSUM, true>), dim3((unsigned)(blocks / 2)), dim3(2 * GEMM_THREADS), 0, s, g);
            return vlg_last_error();
        }
#endif
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, 32, A_KC, B_KC, EPI, COLSUM>), dim3((unsigned)blocks), block, 0, s, g);
    return vlg_last_error();
}

// bf16-MFMA kernels with bf16 LDS tiles (gemm_bf16.hip)
int vlg_gemm16_fwd(GemmArgs g, int epilogue, int io, hipStream_t s);
int vlg_gemm16_dgrad(GemmArgs g, int epilogue, int io, hipStream_t s);
int vlg_gemm16_wgrad(GemmArgs g, int io, hipStream_t s);
int vlg_gemm16_pair(GemmArgs gd, GemmArgs gw, int epilogue, bool dy_bf16, const int64_t* rider, int rider_rows, int rider_bpr, hipStream_t s);
// fp32 operands split into three bf16 terms, six bf16 MFMAs per product block (gemm_split.hip)
int vlg_gemm_split_fwd(GemmArgs g, int epilogue, hipStream_t s);
int vlg_gemm_split_dgrad(GemmArgs g, int epilogue, hipStream_t s);
int vlg_gemm_split_wgrad(GemmArgs g, hipStream_t s);
// storage bits of the epilogue / flags word -> IO template value (bit 0 A, bit 1 B, bit 2 C + aux)
static int gemm_io_bits(int flags) {
    return ((flags & VLG_EPI_A_BF16) ? 1 : 0) | ((flags & VLG_EPI_B_BF16) ? 2 : 0) | ((flags & VLG_EPI_OUT_BF16) ? 4 : 0);
}
#define VLG_EPI_STORAGE (VLG_EPI_A_BF16 | VLG_EPI_B_BF16 | VLG_EPI_OUT_BF16)

static bool gemm_ptr_ok(const void* p, int ld) { return vlg_aligned16(p) && (ld & 3) == 0; }

extern "C" int vlg_linear_fwd(const void* A, int lda, const void* W, int ldw, const float* bias,
                              void* C, int ldc, const void* aux_in, void* aux_out,
                              int64_t M, int N, int K, int epilogue, void* stream) {
    if (M < 1 || N < 1 || K < 4 || (K & 3) || lda < K || ldw < K || ldc < N) return VLG_ERR_SHAPE;
    if (!gemm_ptr_ok(A, lda) || !gemm_ptr_ok(W, ldw) || !C) return VLG_ERR_ALIGN;
    GemmArgs g{};
    g.A = A; g.B = W; g.C = C; g.bias = bias; g.aux_in = aux_in; g.aux_out = aux_out;
    g.M = M; g.N = N; g.Kc = K; g.lda = lda; g.ldb = ldw; g.ldc = ldc;
    g.splits = 1; g.kc_per_split = K; g.slab_stride = 0; g.colsum_off = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool bf16 = (epilogue & VLG_EPI_BF16) != 0, split3 = (epilogue & VLG_EPI_SPLIT3) != 0;
    const int io = gemm_io_bits(epilogue);
    epilogue &= ~(VLG_EPI_BF16 | VLG_EPI_STORAGE | VLG_EPI_SPLIT3);
    if (split3 && (bf16 || io != 0)) return VLG_ERR_SHAPE;
    if ((epilogue & VLG_EPI_BIAS) && !bias) return VLG_ERR_SHAPE;
    if ((epilogue & (VLG_EPI_RESID | VLG_EPI_DGELU)) && !aux_in) return VLG_ERR_SHAPE;
    if ((epilogue & VLG_EPI_GELU) && !aux_out) return VLG_ERR_SHAPE;
    if ((epilogue & VLG_EPI_MUL) != 0) return VLG_ERR_SHAPE;                                          // a dgrad epilogue
    if ((epilogue & VLG_EPI_GELU_GRAD) && (split3 || epilogue != (VLG_EPI_BIAS | VLG_EPI_GELU | VLG_EPI_GELU_GRAD)))
        return VLG_ERR_SHAPE;                                                                         // native fp32 and bf16 paths
    const bool narrow = N <= 32;
    const bool act_gelu = (epilogue & VLG_EPI_ACT_GELU) != 0;
    epilogue &= ~VLG_EPI_ACT_GELU;
    if (act_gelu && (bf16 || split3)) return VLG_ERR_SHAPE;      // native fp32 path only
    if (io != 0 && !bf16) return VLG_ERR_SHAPE;      // bf16 activation storage exists for the bf16 MFMA mode only
    if (((io & 1) && (lda & 7)) || ((io & 2) && (ldw & 7))) return VLG_ERR_ALIGN;
    if (bf16) return vlg_gemm16_fwd(g, epilogue, io, s);
    if (split3) return vlg_gemm_split_fwd(g, epilogue, s);
    if (act_gelu) {                                   // A = gelu(stored pre-activation): the FFN's second projection
        if (epilogue != (VLG_EPI_BIAS | VLG_EPI_RESID) || narrow) return VLG_ERR_SHAPE;
        return launch_gemm<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_RESID | GEMM_A_GELU, false>(g, s);
    }
    switch (epilogue) {
        case VLG_EPI_BIAS:
            return narrow ? launch_gemm<128, 32, true, true, VLG_EPI_BIAS, false>(g, s)
                          : launch_gemm<128, 128, true, true, VLG_EPI_BIAS, false>(g, s);
        case VLG_EPI_BIAS | VLG_EPI_GELU:
            return launch_gemm<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU, false>(g, s);
        case VLG_EPI_BIAS | VLG_EPI_GELU | VLG_EPI_GELU_GRAD:
            return launch_gemm<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_GELU | VLG_EPI_GELU_GRAD, false>(g, s);
        case VLG_EPI_BIAS | VLG_EPI_RESID:
            return launch_gemm<128, 128, true, true, VLG_EPI_BIAS | VLG_EPI_RESID, false>(g, s);
        default:
            return VLG_ERR_SHAPE;
    }
}

extern "C" int vlg_linear_dgrad(const void* dY, int ldy, const void* W, int ldw, void* dX, int ldx,
                                const void* aux_in, int64_t M, int N, int K, int epilogue, void* stream) {
    // dX[M,K] = dY[M,N] . W[N,K]  : contraction over N, W is contraction-major
    if (M < 1 || N < 4 || (N & 3) || K < 4 || (K & 3) || ldy < N || ldw < K || ldx < K) return VLG_ERR_SHAPE;
    if (!gemm_ptr_ok(dY, ldy) || !gemm_ptr_ok(W, ldw) || !dX) return VLG_ERR_ALIGN;
    GemmArgs g{};
    g.A = dY; g.B = W; g.C = dX; g.aux_in = aux_in;
    g.M = M; g.N = K; g.Kc = N; g.lda = ldy; g.ldb = ldw; g.ldc = ldx;
    g.splits = 1; g.kc_per_split = N;
    hipStream_t s = (hipStream_t)stream;
    const bool bf16 = (epilogue & VLG_EPI_BF16) != 0, split3 = (epilogue & VLG_EPI_SPLIT3) != 0;
    const int io = gemm_io_bits(epilogue);
    epilogue &= ~(VLG_EPI_BF16 | VLG_EPI_STORAGE | VLG_EPI_SPLIT3);
    if (split3 && (bf16 || io != 0)) return VLG_ERR_SHAPE;
    if (io != 0 && !bf16) return VLG_ERR_SHAPE;
    if (((io & 1) && (ldy & 7)) || ((io & 2) && (ldw & 7))) return VLG_ERR_ALIGN;
    if ((epilogue == VLG_EPI_DGELU || epilogue == VLG_EPI_MUL) && !aux_in) return VLG_ERR_SHAPE;
    if (epilogue == VLG_EPI_MUL && split3) return VLG_ERR_SHAPE;                                      // native fp32 and bf16 paths
    if (bf16) return vlg_gemm16_dgrad(g, epilogue, io, s);
    if (split3) return vlg_gemm_split_dgrad(g, epilogue, s);
    switch (epilogue) {
        case VLG_EPI_NONE:
            return launch_gemm<128, 128, true, false, VLG_EPI_NONE, false>(g, s);
        case VLG_EPI_DGELU:
            if (!aux_in) return VLG_ERR_SHAPE;
            return launch_gemm<128, 128, true, false, VLG_EPI_DGELU, false>(g, s);
        case VLG_EPI_MUL:
            return launch_gemm<128, 128, true, false, VLG_EPI_MUL, false>(g, s);
        default:
            return VLG_ERR_SHAPE;
    }
}

// split plan for the weight gradient: enough blocks to fill 256 CUs x 2 blocks, each split a
// multiple of BK token rows
// *small (native fp32 kernel only): 64x64 tiles, four resident blocks per CU, token ranges down to 128 rows - taken when the
// 128x128 plan would leave CUs without a block (few tokens: the strong-scaling shard)
static void wgrad_plan(int64_t M, int N, int K, int* splits, int64_t* per, bool bf16 = false, bool* small = nullptr) {
    const int bm = N <= 32 ? 32 : 128;
    int64_t tiles = ((N + bm - 1) / bm) * (int64_t)((K + 127) / 128);
    int64_t want = 512 / tiles;                      // blocks <= 512 = 256 CUs x 2 resident blocks: one full wave, no tail
    int64_t max_splits = (M + 255) / 256;
    if (small) *small = false;
    if (small && bm == 128 && gemm_small_tiles() && tiles * (want < max_splits ? (want < 1 ? 1 : want) : max_splits) < gemm_small_below_wgrad()) {
        *small = true;
        tiles = ((N + 63) / 64) * (int64_t)((K + 63) / 64);
        static int slots = -1;                       // VLG_WGRAD_SMALL_SLOTS (read once; A/B runs)
        if (slots < 0) { const char* e = getenv("VLG_WGRAD_SMALL_SLOTS"); slots = e ? atoi(e) : 1024; }
        want = slots / tiles;
        max_splits = (M + 127) / 128;
    }
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    if (small && *small) {
        // equal token ranges where a slightly smaller count allows them (a ragged last range costs a whole block round)
        for (int64_t c = want; c >= 1 && 4 * c >= 3 * want; --c)
            if (M % (c * 64) == 0) { want = c; break; }
    }
    int64_t p = (M + want - 1) / want;
    const int kt = 64;                               // whole K tiles, and an even number of the fp32 kernel's 32-row tiles (its
                                                     // loop runs them in pairs: an odd count costs one iteration on zeros)
    p = (p + kt - 1) / kt * kt;
    *per = p;
    *splits = (int)((M + p - 1) / p);
}

// (the small-tile plan exists for the native fp32 kernel only: flags without the bf16 / split bits)
static bool wgrad_native(int flags) { return (flags & (VLG_EPI_BF16 | VLG_EPI_SPLIT3 | VLG_EPI_ACT_GELU)) == 0; }
extern "C" int vlg_linear_wgrad_slabs(int64_t M, int N, int K) {
    int splits; int64_t per; bool small;
    wgrad_plan(M, N, K, &splits, &per, false, &small);
    return splits;
}
extern "C" int vlg_linear_wgrad_slabs_for(int64_t M, int N, int K, int flags) {
    int splits; int64_t per; bool small;
    wgrad_plan(M, N, K, &splits, &per, (flags & VLG_EPI_BF16) != 0, wgrad_native(flags) ? &small : nullptr);
    return splits;
}

extern "C" int vlg_linear_wgrad(const void* dY, int ldy, const void* X, int ldx, float* slabs,
                                int64_t slab_stride, int64_t slab_capacity, int64_t M, int N, int K, int flags, void* stream) {
    // slab[s][n*K + k] = sum_{m in split s} dY[m,n] X[m,k] ;  slab[s][N*K + n] = sum_m dY[m,n]
    if (M < 1 || N < 4 || (N & 3) || K < 4 || (K & 3) || ldy < N || ldx < K) return VLG_ERR_SHAPE;
    if (slab_stride < (int64_t)N * K + N) return VLG_ERR_SHAPE;
    if (!gemm_ptr_ok(dY, ldy) || !gemm_ptr_ok(X, ldx) || !slabs) return VLG_ERR_ALIGN;
    GemmArgs g{};
    g.A = dY; g.B = X; g.C = slabs;
    g.M = N; g.N = K; g.Kc = M; g.lda = ldy; g.ldb = ldx; g.ldc = K;
    bool small = false;
    wgrad_plan(M, N, K, &g.splits, &g.kc_per_split, (flags & VLG_EPI_BF16) != 0, wgrad_native(flags) ? &small : nullptr);
    if (slab_capacity < (int64_t)g.splits * slab_stride) return VLG_ERR_SHAPE;      // the caller's buffer must hold every slab
    g.slab_stride = slab_stride; g.colsum_off = (int64_t)N * K;
    hipStream_t s = (hipStream_t)stream;
    const bool bf16 = (flags & VLG_EPI_BF16) != 0;
    const int io = gemm_io_bits(flags);
    if (io != 0 && !bf16) return VLG_ERR_SHAPE;
    if (((io & 1) && (ldy & 7)) || ((io & 2) && (ldx & 7))) return VLG_ERR_ALIGN;
    if (bf16) return (flags & VLG_EPI_ACT_GELU) ? VLG_ERR_SHAPE : vlg_gemm16_wgrad(g, io, s);
    if (flags & VLG_EPI_SPLIT3) return (io == 0 && !(flags & VLG_EPI_ACT_GELU)) ? vlg_gemm_split_wgrad(g, s) : VLG_ERR_SHAPE;
    if (flags & VLG_EPI_ACT_GELU)                     // X = gelu(stored pre-activation): weight gradient of the FFN's second projection
        return N <= 32 ? VLG_ERR_SHAPE : launch_gemm<128, 128, false, false, GEMM_B_GELU, true>(g, s);
    if (small) return launch_gemm<64, 64, false, false, VLG_EPI_NONE, true>(g, s);
    return N <= 32 ? launch_gemm<32, 128, false, false, VLG_EPI_NONE, true>(g, s)
                   : launch_gemm<128, 128, false, false, VLG_EPI_NONE, true>(g, s);
}

// ---- data gradient + weight gradient of one projection in ONE launch (gemm_pair_kernel)
// VLG_GEMM_PAIR (read once): 0 = always two launches, 1 = one launch only where both problems take the 64x64 tiles (few
// tokens: neither fills the chip alone), 2 (default) = also on 128x128 tiles.  Measured at the metric shape (interleaved
// repeated runs, one box): 5.553 -> 5.500 ms per step (-1.0 %, three of three pairs of runs) - each launch pays one ramp
// and one drain for two problems, and the second problem's blocks fill the slots the first one's stragglers leave.
static int gemm_pair_mode() {
    static const int mode = [] { const char* e = getenv("VLG_GEMM_PAIR"); return e ? atoi(e) : 2; }();
    return mode;
}
#define VLG_RIDER_BPR 128          /* blocks per table row: what vlg_reduce_slabs_table launches get from the engine */
template <int BT, int EPI_D>
static int launch_pair(GemmArgs gd, GemmArgs gw, const int64_t* rider, int rider_rows, hipStream_t s) {
    gd.tiles_m = (int)((gd.M + BT - 1) / BT); gd.tiles_n = (gd.N + BT - 1) / BT; gd.run = 1; gd.clock_probe = nullptr;
    gw.tiles_m = (int)((gw.M + BT - 1) / BT); gw.tiles_n = (gw.N + BT - 1) / BT; gw.run = 1; gw.clock_probe = nullptr;
    if constexpr (BT == 128) gd.run = gemm_run<128, 128, 32, true, false>(gd, 512);
    const int64_t nd = (int64_t)gd.tiles_m * (gd.tiles_n / gd.run), nw = (int64_t)gw.tiles_m * gw.tiles_n * gw.splits;
    const int64_t nd_pad = (nd + 7) / 8 * 8;
    const int64_t nr = rider ? (int64_t)rider_rows * VLG_RIDER_BPR : 0;
    if (nd < 1 || nw < 1 || nd_pad + nw + nr > 0x7fffffff) return VLG_ERR_SHAPE;
    hipLaunchKernelGGL((gemm_pair_kernel<BT, EPI_D>), dim3((unsigned)(nd_pad + nw + nr)), dim3(GEMM_THREADS), 0, s, gd, gw, (int)nd, (int)nd_pad,
                       (int)nw, rider, VLG_RIDER_BPR);
    return vlg_last_error();
}

extern "C" int vlg_reduce_slabs_table(const int64_t* table, int n_rows, int blocks_per_row, void* stream);

extern "C" int vlg_linear_dgrad_wgrad(const void* dY, int ldy, const void* W, int ldw, void* dX, int ldx, const void* aux_in,
                                      const void* X, int ldxx, float* slabs, int64_t slab_stride, int64_t slab_capacity,
                                      int64_t M, int N, int K, int epilogue, const int64_t* rider_table, int rider_rows,
                                      void* stream) {
    // dX[M,K] = dY[M,N] . W[N,K] (x aux_in with VLG_EPI_MUL)   and   slab[s] = dY^T . X, column sums of dY  - the results of
    // vlg_linear_wgrad followed by vlg_linear_dgrad with the same arguments, bit for bit.  rider_table (may be NULL): rows of
    // a slab-reduction table (vlg_reduce_slabs_table) over OTHER buffers than this call writes, reduced by extra blocks of the
    // same launch where the two products are fused, by a launch of their own otherwise - same sums either way.
    if (rider_table != nullptr && (rider_rows < 1 || rider_rows > 4096)) return VLG_ERR_SHAPE;
    auto rider_alone = [&]() -> int { return rider_table ? vlg_reduce_slabs_table(rider_table, rider_rows, VLG_RIDER_BPR, stream) : 0; };
    // the bf16-storage step (bf16 W / X / dX, dY bf16 or fp32): one launch of the bf16-tile kernels at every shape
    const int st_bits = epilogue & (VLG_EPI_A_BF16 | VLG_EPI_B_BF16 | VLG_EPI_OUT_BF16);
    if ((epilogue & VLG_EPI_BF16) && (st_bits & ~VLG_EPI_A_BF16) == (VLG_EPI_B_BF16 | VLG_EPI_OUT_BF16) && gemm_pair_mode() > 1 && N > 32 && K > 32) {
        const int epi = epilogue & ~(VLG_EPI_BF16 | VLG_EPI_A_BF16 | VLG_EPI_B_BF16 | VLG_EPI_OUT_BF16);
        if (epi != VLG_EPI_NONE && epi != VLG_EPI_MUL) return VLG_ERR_SHAPE;
        if (M < 1 || (N & 7) || (K & 7) || ldy < N || ldw < K || ldx < K || ldxx < K) return VLG_ERR_SHAPE;
        const bool a16 = (st_bits & VLG_EPI_A_BF16) != 0;
        if (!vlg_aligned16(dY) || !vlg_aligned16(W) || !vlg_aligned16(X) || !dX || !slabs || (a16 && (ldy & 7)) || (ldw & 7) || (ldxx & 7) || (ldx & 7))
            return VLG_ERR_ALIGN;
        if (epi == VLG_EPI_MUL && !aux_in) return VLG_ERR_SHAPE;
        GemmArgs gd{}, gw{};
        gd.A = dY; gd.B = W; gd.C = dX; gd.aux_in = aux_in;
        gd.M = M; gd.N = K; gd.Kc = N; gd.lda = ldy; gd.ldb = ldw; gd.ldc = ldx;
        gd.splits = 1; gd.kc_per_split = N;
        gw.A = dY; gw.B = X; gw.C = slabs;
        gw.M = N; gw.N = K; gw.Kc = M; gw.lda = ldy; gw.ldb = ldxx; gw.ldc = K;
        wgrad_plan(M, N, K, &gw.splits, &gw.kc_per_split, true);
        if (slab_stride < (int64_t)N * K + N || slab_capacity < (int64_t)gw.splits * slab_stride) return VLG_ERR_SHAPE;
        gw.slab_stride = slab_stride; gw.colsum_off = (int64_t)N * K;
        return vlg_gemm16_pair(gd, gw, epi, a16, rider_table, rider_rows, VLG_RIDER_BPR, (hipStream_t)stream);
    }
    const bool native = (epilogue & ~VLG_EPI_MUL) == 0;
    bool small_w = false;
    int splits = 1; int64_t per = 0;
    if (native && gemm_pair_mode() > 0 && M >= 1 && N >= 4 && K >= 4) wgrad_plan(M, N, K, &splits, &per, false, &small_w);
    const bool small_d = gemm_wants_small(M, K, 1);
    const bool pair = native && gemm_pair_mode() > 0 && small_w == small_d && (small_d || gemm_pair_mode() > 1) && N > 32 && K > 32;
    if (!pair) {
        // (bf16 storage: A = the shared dY, B = W of the data gradient AND X of the weight gradient, OUT = dX)
        const int wflags = epilogue & (VLG_EPI_BF16 | VLG_EPI_SPLIT3 | VLG_EPI_A_BF16 | VLG_EPI_B_BF16);
        if (const int rc0 = rider_alone()) return rc0;
        const int rc = vlg_linear_wgrad(dY, ldy, X, ldxx, slabs, slab_stride, slab_capacity, M, N, K, wflags, stream);
        if (rc != 0) return rc;
        return vlg_linear_dgrad(dY, ldy, W, ldw, dX, ldx, aux_in, M, N, K, epilogue, stream);
    }
    if ((N & 3) || (K & 3) || ldy < N || ldw < K || ldx < K || ldxx < K) return VLG_ERR_SHAPE;
    if (!gemm_ptr_ok(dY, ldy) || !gemm_ptr_ok(W, ldw) || !gemm_ptr_ok(X, ldxx) || !dX || !slabs) return VLG_ERR_ALIGN;
    if ((epilogue & VLG_EPI_MUL) && !aux_in) return VLG_ERR_SHAPE;
    if (slab_stride < (int64_t)N * K + N || slab_capacity < (int64_t)splits * slab_stride) return VLG_ERR_SHAPE;
    GemmArgs gd{}, gw{};
    gd.A = dY; gd.B = W; gd.C = dX; gd.aux_in = aux_in;
    gd.M = M; gd.N = K; gd.Kc = N; gd.lda = ldy; gd.ldb = ldw; gd.ldc = ldx;
    gd.splits = 1; gd.kc_per_split = N;
    gw.A = dY; gw.B = X; gw.C = slabs;
    gw.M = N; gw.N = K; gw.Kc = M; gw.lda = ldy; gw.ldb = ldxx; gw.ldc = K;
    gw.splits = splits; gw.kc_per_split = per; gw.slab_stride = slab_stride; gw.colsum_off = (int64_t)N * K;
    hipStream_t s = (hipStream_t)stream;
    const bool mul = (epilogue & VLG_EPI_MUL) != 0;
    if (small_d) return mul ? launch_pair<64, VLG_EPI_MUL>(gd, gw, rider_table, rider_rows, s) : launch_pair<64, VLG_EPI_NONE>(gd, gw, rider_table, rider_rows, s);
    return mul ? launch_pair<128, VLG_EPI_MUL>(gd, gw, rider_table, rider_rows, s) : launch_pair<128, VLG_EPI_NONE>(gd, gw, rider_table, rider_rows, s);
}
