// Operand-tile machinery shared by the dense GEMMs (gemm.hip) and the implicit-GEMM convolutions
// (conv.hip): LDS tile layouts, global->register->LDS staging, MFMA fragment reads.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define GEMM_THREADS 256

struct GemmArgs {
    const void* A; const void* B; void* C;       // element types are template parameters of the kernel (float or bf16_t)
    const float* bias; const void* aux_in; void* aux_out;
    int64_t M;               // rows of C (non-contraction extent of A)
    int N;                   // cols of C (non-contraction extent of B)
    int64_t Kc;              // contraction extent
    int lda, ldb, ldc;
    int tiles_m, tiles_n, splits;
    int run;                 // consecutive N tiles one block computes back to back (fp32 fast path; 1 everywhere else)
    int64_t kc_per_split, slab_stride, colsum_off;
    unsigned long long* clock_probe;   // diagnostic only (VLG_GEMM_CLOCK_PROBE): {shader ticks, 100 MHz ticks} per block
};


// One operand tile: BR rows (non-contraction) x BK contraction steps.
// NT = threads of the block that stages the tile (256 everywhere but the 8-wave persistent GEMM)
template <int BR, bool KC, int BK, int NT = GEMM_THREADS>
struct Tile {
    static constexpr int LDK = BK + 4;                                  // padded row stride of a KC tile
    static constexpr int F4 = BR * BK / 4;                              // float4 per tile
    static constexpr int NV = (F4 + NT - 1) / NT;   // float4 per thread
    static constexpr int FLOATS = KC ? BR * LDK : BK * BR;
    static constexpr int PER_ROW = KC ? BK / 4 : BR / 4;                // float4 per memory row

    // GUARD = false: interior tile, plain loads that stay in flight until the LDS write.
    // GUARD = true : edge tile; address clamped into the operand and the value zeroed by a select
    //                (no branches).  Requires R >= 4 and kend - k0 >= 4 when anything is in range.
    // E = element type in memory (float, or bf16_t widened on load: the LDS tile is fp32 either way)
    template <bool GUARD, typename E>
    __device__ static __forceinline__ void gload(float4 (&r)[NV], const E* __restrict__ P, int ld,
                                                 int64_t r0, int64_t R, int64_t k0, int64_t kend, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            if (F4 % NT != 0 && idx >= F4) { r[i] = f4_zero(); continue; }
            const int major = idx / PER_ROW, minor = (idx % PER_ROW) << 2;
            // KC: memory row = operand row, column = k.   MC: memory row = k, column = operand row.
            const int64_t grow = KC ? r0 + major : k0 + major;
            const int64_t gcol = KC ? k0 + minor : r0 + minor;
            const int64_t rlim = KC ? R : kend, clim = KC ? kend : R;
            if constexpr (GUARD) {
                const bool ok = grow < rlim && gcol < clim;
                const float4 v = ld4(P + (grow < rlim ? grow : rlim - 1) * ld + (gcol < clim ? gcol : clim - 4));
                r[i] = ok ? v : f4_zero();
            } else {
                r[i] = ld4(P + grow * ld + gcol);
            }
        }
    }
    __device__ static __forceinline__ void sstore(const float4 (&r)[NV], float* S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + NT * i;
            if (F4 % NT != 0 && idx >= F4) continue;
            if constexpr (KC) st4(S + (idx / PER_ROW) * LDK + ((idx % PER_ROW) << 2), r[i]);
            else st4(S + (idx << 2), r[i]);
        }
    }
    // ---- address pieces for loops that keep every per-iteration address out of the vector ALU (gemm.hip, fast path):
    // a thread's float4 number i sits ISTEP memory rows below its float4 0, in memory and (KC) in LDS alike, so the
    // per-thread part of every address is a loop constant and the rest is a scalar or an instruction immediate.
    static constexpr int ISTEP = NT / PER_ROW;                          // memory rows between a thread's consecutive float4
    static constexpr int SSTEP = KC ? ISTEP * LDK : NT * 4;             // the same step in the LDS tile, floats
    __device__ static __forceinline__ int voff_bytes(int ld, int tid) { return ((tid / PER_ROW) * ld + ((tid % PER_ROW) << 2)) * 4; }
    __device__ static __forceinline__ int soff(int tid) { return KC ? (tid / PER_ROW) * LDK + ((tid % PER_ROW) << 2) : (tid << 2); }
    __device__ static __forceinline__ int roff(int row, int h) { return KC ? row * LDK + 4 * h : 4 * h * BR + row; }
    // fragment of the 32-row group `grp` for chunk s, from base = tile + roff(first row of the wave, h): all offsets immediates
    // (grp and s are compile-time constants once the caller's loops are unrolled)
    __device__ static __forceinline__ void frag_at(float (&f)[4], const float* base, int grp, int s) {
        if constexpr (KC) {
            const float4 t = ld4(base + grp * 32 * LDK + 8 * s);
            f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = base[(8 * s + j) * BR + grp * 32];
        }
    }
    // 4 fragment values of tile row `row` for chunk s, lane half h:  k = 8s + 4h + j
    __device__ static __forceinline__ void frag(float (&f)[4], const float* S, int row, int s, int h) {
        if constexpr (KC) {
            const float4 t = ld4(S + row * LDK + 8 * s + 4 * h);
            f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = S[(8 * s + 4 * h + j) * BR + row];
        }
    }
};

