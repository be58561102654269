// bf16 operand tiles in LDS for the bf16-MFMA GEMMs (gemm_bf16.hip, gemm_split.hip): images, staging, fragment reads.
//   KC image  [row][k]   rows of BK bf16 + 8 pad: per-lane 16-byte fragment reads (ds_read_b128, lane = row, half-wave h
//                        picks k = 16s + 8h ..+7) are bank-conflict-free for BK = 64 (36-dword rows) and 16 (12-dword rows)
//   MC image  [k][row]   the operand is contraction-major in memory: stored as it arrives and read with gfx950's
//                        TRANSPOSING LDS read ds_read_b64_tr_b16 - each 16-lane group fetches a 4(k) x 16(row) block and
//                        every lane receives the 4 k-values of ITS row; two reads make the 8-deep MFMA operand.  Row stride
//                        = BR*2 + 64 bytes: the four k-rows of a block start 64 B apart modulo the 256-B bank window.
// Both images give lane (row = l & 31, h = l >> 5) the k-values 16s + 8h + j, j = 0..7.
#pragma once
#include "gemm_tile.h"
#include <type_traits>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;


// pack 8 fp32 values (two float4) into 8 bf16 (round to nearest even)
__device__ __forceinline__ uint4 pack_bf16x8(float4 a, float4 b) {
    const bf16x8 t = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w, (bf16_t)b.x, (bf16_t)b.y, (bf16_t)b.z, (bf16_t)b.w};
    return __builtin_bit_cast(uint4, t);
}

// Register stage of one operand tile: raw loads, converted only when written to LDS (so the loads stay in flight).
template <typename E, int NV> struct Stage16;
template <int NV> struct Stage16<bf16_t, NV> { uint4 v[NV]; };
template <int NV> struct Stage16<float, NV> { float4 lo[NV], hi[NV]; };

template <int BR, bool KC, int BK16>
struct Tile16 {
    static constexpr int RS = KC ? BK16 + 8 : BR + 32;                   // row stride of the LDS image, elements
    static constexpr int ELEMS = KC ? BR * RS : BK16 * RS;
    static constexpr int SLOTS = BR * BK16 / 8;                          // 16-byte (8-element) slots per tile
    static constexpr int NV = (SLOTS + GEMM_THREADS - 1) / GEMM_THREADS;
    static constexpr int PER_ROW = KC ? BK16 / 8 : BR / 8;               // slots per memory row

    template <bool GUARD, typename E>
    __device__ static __forceinline__ void gload(Stage16<E, NV>& st, const E* __restrict__ P, int ld,
                                                 int64_t r0, int64_t R, int64_t k0, int64_t kend, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            const bool live = (SLOTS % GEMM_THREADS == 0) || idx < SLOTS;
            const int major = idx / PER_ROW, minor = (idx % PER_ROW) << 3;
            int64_t grow = KC ? r0 + major : k0 + major;
            int64_t gcol = KC ? k0 + minor : r0 + minor;
            bool ok = live;
            if constexpr (GUARD) {
                const int64_t rlim = KC ? R : kend, clim = KC ? kend : R;     // clim is a multiple of 8 (checked on the host)
                ok = live && grow < rlim && gcol < clim;
                grow = grow < rlim ? grow : rlim - 1;
                gcol = gcol < clim ? gcol : clim - 8;
            } else if (!live) { grow = KC ? r0 : k0; gcol = KC ? k0 : r0; }
            const E* src = P + grow * ld + gcol;
            if constexpr (std::is_same<E, float>::value) {
                const float4 a = ld4(src), b = ld4(src + 4);
                st.lo[i] = ok ? a : f4_zero();
                st.hi[i] = ok ? b : f4_zero();
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(src);
                st.v[i] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    template <typename E>
    __device__ static __forceinline__ void sstore(const Stage16<E, NV>& st, bf16_t* S, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GEMM_THREADS * i;
            if (SLOTS % GEMM_THREADS != 0 && idx >= SLOTS) continue;
            uint4 v;
            if constexpr (std::is_same<E, float>::value) v = pack_bf16x8(st.lo[i], st.hi[i]);
            else v = st.v[i];
            *reinterpret_cast<uint4*>(S + (idx / PER_ROW) * RS + ((idx % PER_ROW) << 3)) = v;
        }
    }
    // MFMA operand of the 32-row sub-tile starting at tile row rb for k16-step s: lane (l31, h) gets k = 16s + 8h + 0..7
    __device__ static __forceinline__ bf16x8 frag(const bf16_t* S, int rb, int s, int lane) {
        if constexpr (KC) {
            return *reinterpret_cast<const bf16x8*>(S + (rb + (lane & 31)) * RS + 16 * s + 8 * (lane >> 5));
        } else {
            // group g = lane>>4 covers rows rb + 16(g&1) .. +15 and k = 16s + 8(g>>1) .. +7; lane 4q+p of the group supplies
            // the address of k-row q, rows 4p..4p+3 and receives the 4 k-values of row (lane & 15)
            const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
            const bf16_t* a = S + (16 * s + 8 * (g >> 1) + q) * RS + rb + 16 * (g & 1) + 4 * p;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * RS));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, t);
        }
    }
    // column sums of the tile this thread staged (weight gradient: bias gradient), 8 rows per thread
    template <typename E>
    __device__ static __forceinline__ void colsum_add(float (&acc)[8], const Stage16<E, NV>& st) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if constexpr (std::is_same<E, float>::value) {
                acc[0] += st.lo[i].x; acc[1] += st.lo[i].y; acc[2] += st.lo[i].z; acc[3] += st.lo[i].w;
                acc[4] += st.hi[i].x; acc[5] += st.hi[i].y; acc[6] += st.hi[i].z; acc[7] += st.hi[i].w;
            } else {
                const unsigned w[4] = {st.v[i].x, st.v[i].y, st.v[i].z, st.v[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * j] += __uint_as_float(w[j] << 16);
                    acc[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
                }
            }
        }
    }
};

