// Development micro-benchmark: what besides the MFMAs costs the fp32 GEMM loop its clock / issue slots?
// Variants of a register-resident fp32 MFMA loop (2 waves / SIMD, 64 MFMAs = one 128x128x32 k-tile per iteration):
//   0 bare   1 + LDS fragment reads (16 ds_read_b128 per tile)   2 + global loads of a k-tile (8 float4 / thread, L2-resident)
//   3 + both   4 + both + LDS writes + barrier (the full staging pattern)   5 = 4 with the loads STREAMING from HBM
//   (groups of 4 blocks share a stream: ~1.2 TB/s of HBM reads chip-wide, the weight-gradient GEMM's rate)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_mix.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void fill_random(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x400000);
    }
}
template <int V, bool RANDOM>
__global__ __launch_bounds__(256, 2) void kern(float* out, const float* __restrict__ src, int iters, unsigned long long* ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __shared__ __attribute__((aligned(16))) float lds[2 * 9216];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 9216; i += 256) { unsigned h = (i + 977u * blockIdx.x) * 2654435761u; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        lds[i] = RANDOM ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x400000) : i * 1e-4f; }
    __syncthreads();
    float4 g[8];
    for (int i = 0; i < 8; ++i) g[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    const float* p = src + (size_t)(blockIdx.x % 64) * 65536 + tid * 4;
    const float* ps = src + (size_t)(blockIdx.x / 4) * (size_t)iters * 8192 + tid * 4;      // variant 5: 32 KB per iteration per group
    float a = 1.0f + tid * 1e-3f, b = 2.0f + tid * 2e-3f;
    if (RANDOM) { a = lds[tid * 7 % 9216]; b = lds[(tid * 13 + 5) % 9216]; }
    for (int it = 0; it < iters; ++it) {
        const float* L = lds + (it & 1) * 9216;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                        // 4 chunks of 16 MFMAs
            float4 fa0, fa1, fb0, fb1;
            if (V == 1 || V >= 3) {
                fa0 = *(const float4*)(L + (lane & 31) * 36 + 8 * c + 4 * (lane >> 5));
                fa1 = *(const float4*)(L + (32 + (lane & 31)) * 36 + 8 * c + 4 * (lane >> 5));
                fb0 = *(const float4*)(L + 4608 + (lane & 31) * 36 + 8 * c + 4 * (lane >> 5));
                fb1 = *(const float4*)(L + 4608 + (32 + (lane & 31)) * 36 + 8 * c + 4 * (lane >> 5));
            } else { fa0 = fa1 = make_float4(a, a, a, a); fb0 = fb1 = make_float4(b, b, b, b); }
            const float av[2][4] = {{fa0.x, fa0.y, fa0.z, fa0.w}, {fa1.x, fa1.y, fa1.z, fa1.w}};
            const float bv[2][4] = {{fb0.x, fb0.y, fb0.z, fb0.w}, {fb1.x, fb1.y, fb1.z, fb1.w}};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][k], bv[j][k], acc[i * 2 + j], 0, 0, 0);
            if (c == 1) {
                if (V >= 4) {
                    float* W = lds + ((it & 1) ^ 1) * 9216;
#pragma unroll
                    for (int i = 0; i < 8; ++i) *(float4*)(W + ((tid + 256 * i) / 8) * 36 + ((tid + 256 * i) % 8) * 4) = g[i];
                }
                if (V >= 2 && V != 5) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = *(const float4*)(p + ((it * 8 + i) & 63) * 1024);
                }
                if (V == 5) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = *(const float4*)(ps + ((size_t)it * 8 + i) * 1024);
                }
            }
        }
        if (V >= 4) __syncthreads();
        a += 1e-6f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += g[i].x;
    out[blockIdx.x * 256 + tid] = s;
    if (ticks != nullptr && tid == 0 && blockIdx.x == 0) { ticks[0] = __builtin_amdgcn_s_memtime() - t0; ticks[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int V, bool RANDOM> void run(float* out, const float* src, hipEvent_t e0, hipEvent_t e1) {
    static unsigned long long* ticks = nullptr;
    if (!ticks) hipHostMalloc(&ticks, 16);
    const int iters = V == 5 ? 500 : 2000;
    hipEventRecord(e0);
    for (int l = 0; l < 5; ++l) hipLaunchKernelGGL((kern<V, RANDOM>), dim3(512), dim3(256), 0, 0, out, src, iters, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 5.0 * 512 * 4 * (double)iters * 64 * 4096.0;
    if (V == 5) printf("  (HBM stream %.2f TB/s)\n", 5.0 * 128 * iters * 32768.0 / ms / 1e9);
    printf("%s variant %d: %.2f ms  %.1f TFLOP/s   s_memtime/s_memrealtime = %.3f (x100 MHz), MFMA cycles/iter-pair %.0f of %.0f ticks\n", RANDOM ? "random data" : "smooth data", V, ms, flop / ms / 1e9,
           (double)ticks[0] / ticks[1], 2.0 * 64 * 64, (double)ticks[0] / iters);
}
int main() {
    float *out, *src; hipMalloc(&out, 512 * 256 * 4); const size_t nb = (size_t)128 * 500 * 32768 + (1 << 22);
    hipMalloc(&src, nb);
    hipMemset(src, 0, nb);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, nb / 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // sustained load: the same variants after ~3 s of continuous launches (DVFS settles on a power budget)
    for (int w = 0; w < 300; ++w) hipLaunchKernelGGL((kern<4, true>), dim3(512), dim3(256), 0, 0, out, src, 2000, (unsigned long long*)nullptr);
    printf("after 300 back-to-back launches (~2.2 s):\n");
    run<4, true>(out, src, e0, e1); run<5, true>(out, src, e0, e1); run<0, true>(out, src, e0, e1);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 1; ++rep) {
        run<0, false>(out, src, e0, e1); run<0, true>(out, src, e0, e1); run<1, true>(out, src, e0, e1);
        run<4, false>(out, src, e0, e1); run<4, true>(out, src, e0, e1); run<5, true>(out, src, e0, e1);
    }
    return 0;
}
