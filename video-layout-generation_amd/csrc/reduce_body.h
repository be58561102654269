// One row of a slab-reduction table, as a device function of (table, row, block number within the row, blocks per row, 4 KB
// of LDS): reduce_slabs_table_kernel runs it as a launch of its own, gemm_pair_kernel as extra blocks ("riders") of a GEMM
// launch.  table[i] = {slabs pointer, slab stride, slab count, destination pointer, length}, int64 each, in device memory.
// Every row is reduced exactly as vlg_reduce_slabs would reduce it (same summation order) wherever it runs.
#pragma once
#include "common.h"

__device__ __forceinline__ void reduce_table_row(const int64_t* __restrict__ table, const int row, const int blk, const int nblk,
                                                 float4 (*part)[16]) {
    const int64_t* t = table + 5 * (int64_t)row;
    const float* __restrict__ slabs = reinterpret_cast<const float*>(t[0]);
    const int64_t stride = t[1];
    const int n_slabs = (int)t[2];
    float* __restrict__ dst = reinterpret_cast<float*>(t[3]);
    const int64_t len4 = t[4] / 4;
    if (len4 < 32768 && n_slabs >= 16) {                       // as reduce_slabs_tall_kernel
        const int c = threadIdx.x & 15, sg = threadIdx.x >> 4;
        for (int64_t bk = blk; bk * 16 < len4; bk += nblk) {
            const int64_t col = bk * 16 + c;
            float4 acc = f4_zero();
            if (col < len4) {
                const float* p = slabs + col * 4;
                int s = sg;
                for (; s + 48 < n_slabs; s += 64) {
                    const float4 a = ld4(p + s * stride), b2 = ld4(p + (s + 16) * stride);
                    const float4 c2 = ld4(p + (s + 32) * stride), d = ld4(p + (s + 48) * stride);
                    acc = f4_add(acc, f4_add(f4_add(a, b2), f4_add(c2, d)));
                }
                for (; s < n_slabs; s += 16) acc = f4_add(acc, ld4(p + s * stride));
            }
            part[sg][c] = acc;
            __syncthreads();
            if (sg == 0 && col < len4) {
                float4 v = part[0][c];
#pragma unroll
                for (int k = 1; k < 16; ++k) v = f4_add(v, part[k][c]);
                st4(dst + col * 4, v);
            }
            __syncthreads();
        }
    } else {                                                   // as reduce_slabs_kernel
        for (int64_t e = (int64_t)blk * blockDim.x + threadIdx.x; e < len4; e += (int64_t)nblk * blockDim.x) {
            float4 acc = f4_zero();
            const float* p = slabs + e * 4;
            int s = 0;
            for (; s + 4 <= n_slabs; s += 4) {
                const float4 a = ld4(p), b2 = ld4(p + stride), c = ld4(p + 2 * stride), d = ld4(p + 3 * stride);
                acc = f4_add(acc, f4_add(f4_add(a, b2), f4_add(c, d)));
                p += 4 * stride;
            }
            for (; s < n_slabs; ++s) { acc = f4_add(acc, ld4(p)); p += stride; }
            st4(dst + e * 4, acc);
        }
    }
}
