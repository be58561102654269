// Development micro-benchmark: does v_mfma_f32_32x32x2_f32 share its SIMD with the vector ALU?
// (The instruction runs at exactly the fp32 VALU rate, 64 FLOP/clk/SIMD.  If the two use the same lanes, every vector
// instruction issued on a SIMD - by the MFMA wave itself or by a co-resident wave - costs matrix throughput.)
//   part A: ONE wave per SIMD, 4 independent accumulators, N filler instructions behind every MFMA (same wave)
//   part B: TWO waves per SIMD: waves 0-3 issue only MFMAs, waves 4-7 only fillers (other wave)
// filler kinds: 0 v_fma_f32   1 v_add_u32   2 v_exp_f32   3 v_lshl_add_u64   4 ds_read_b128   5 v_mov_b32   6 v_pk_fma_f32   7 v_pk_mul_f32   8 v_rcp_f32   9 v_cndmask_b32
// 10 global_store_dword   11 global_load_dword   12 global_load_dwordx4   13 ds_write_b128   14 global_store_dwordx4 (each lane its own 16-B slot)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_valu_share.hip -o tools/micro/mfma_valu_share && tools/micro/mfma_valu_share
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

template <int KIND>
__device__ __forceinline__ void filler(float& x, unsigned& u, unsigned long long& w, v4f& f4, const float* lds, float a, float b, float* gp = nullptr) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
    else if constexpr (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(u));
    else if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(a));
    else if constexpr (KIND == 3) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w) : "v"(w));
    else if constexpr (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(f4) : "v"((unsigned)(size_t)lds));
    else if constexpr (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(a));
    else if constexpr (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(w) : "v"(w));
    else if constexpr (KIND == 7) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(w) : "v"(w));
    else if constexpr (KIND == 8) asm volatile("v_rcp_f32 %0, %1" : "=v"(x) : "v"(a));
    else if constexpr (KIND == 9) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(a), "v"(b));
    else if constexpr (KIND == 10) asm volatile("global_store_dword %0, %1, off" :: "v"(gp), "v"(a) : "memory");
    else if constexpr (KIND == 11) asm volatile("global_load_dword %0, %1, off" : "=v"(x) : "v"(gp) : "memory");
    else if constexpr (KIND == 12) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(f4) : "v"(gp) : "memory");
    else if constexpr (KIND == 13) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(size_t)lds), "v"(f4) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gp), "v"(f4) : "memory");
}

template <int KIND, int NF>
__global__ __launch_bounds__(256, 1) void same_wave(float* out, int iters, unsigned long long* ticks, float* mem) {
    float* gp = mem + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
    extern __shared__ float lds[];                          // 120 KB requested: one block per CU
    const int tid = threadIdx.x;
    lds[tid] = tid;
    __syncthreads();
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    float a = 1.0f + tid * 1e-3f, b = 0.5f - tid * 2e-3f, x = 0.f;
    unsigned u = tid; unsigned long long w = tid; v4f f4 = {0.f, 0.f, 0.f, 0.f};
    const float* lp = lds + (tid & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            MFMA(acc0);
#pragma unroll
            for (int n = 0; n < NF; ++n) filler<KIND>(x, u, w, f4, lp, a, b, gp);
            MFMA(acc1);
#pragma unroll
            for (int n = 0; n < NF; ++n) filler<KIND>(x, u, w, f4, lp, a, b, gp);
            MFMA(acc2);
#pragma unroll
            for (int n = 0; n < NF; ++n) filler<KIND>(x, u, w, f4, lp, a, b, gp);
            MFMA(acc3);
#pragma unroll
            for (int n = 0; n < NF; ++n) filler<KIND>(x, u, w, f4, lp, a, b, gp);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = x + u + (float)w + f4.x;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

// waves 0-3: MFMAs only; waves 4-7: FPM fillers per MFMA of the other waves (fixed counts: no flag, no spin)
template <int KIND, int FPM, int PRIO = 0>
__global__ __launch_bounds__(512, 1) void other_wave(float* out, int iters, unsigned long long* ticks, unsigned long long* fticks, float* mem) {
    float* gp = mem + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    lds[tid] = tid;
    __syncthreads();
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    float a = 1.0f + tid * 1e-3f, b = 0.5f - tid * 2e-3f, x = 0.f;
    unsigned u = tid; unsigned long long w = tid; v4f f4 = {0.f, 0.f, 0.f, 0.f};
    const float* lp = lds + (tid & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        }
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    } else {
        if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16 * FPM; ++q) filler<KIND>(x, u, w, f4, lp, a, b, gp);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (tid == 256) fticks[blockIdx.x] = t1 - t0;
    }
    float s = x + u + (float)w + f4.x;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 512 + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at line %d\n", (int)e_, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static unsigned long long h_ticks[256], h_fc[256];
static float* g_mem = nullptr;
// two MFMA streams per SIMD: waves 0-3 (older) and waves 4-7 (younger, at s_setprio PRIO): who gets the matrix pipe?
template <int PRIO>
__global__ __launch_bounds__(512, 1) void two_streams(float* out, int iters, unsigned long long* ticks, unsigned long long* fticks) {
    const int tid = threadIdx.x, wave = tid >> 6;
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    float a = 1.0f + tid * 1e-3f, b = 0.5f - tid * 2e-3f;
    __syncthreads();
    if (wave >= 4 && PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (tid == 256) fticks[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 512 + tid] = s;
}
static double median(unsigned long long* p, int n) { std::vector<unsigned long long> v(p, p + n); std::sort(v.begin(), v.end()); return (double)v[n / 2]; }

template <int KIND, int NF> void runA(float* out, unsigned long long* ticks) {
    const int iters = 4000;
    CK(hipFuncSetAttribute((const void*)same_wave<KIND, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    for (int l = 0; l < 3; ++l) hipLaunchKernelGGL((same_wave<KIND, NF>), dim3(256), dim3(256), 120 * 1024, 0, out, iters, ticks, g_mem);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_ticks, ticks, sizeof h_ticks, hipMemcpyDeviceToHost));
    printf("  same wave, kind %d, %d fillers per MFMA: %.1f cycles per MFMA\n", KIND, NF, median(h_ticks, 256) / (iters * 16.0));
}
template <int KIND, int FPM, int PRIO = 0> void runB(float* out, unsigned long long* ticks, unsigned long long* fc) {
    const int iters = 4000;
    CK(hipFuncSetAttribute((const void*)other_wave<KIND, FPM, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    for (int l = 0; l < 3; ++l) hipLaunchKernelGGL((other_wave<KIND, FPM, PRIO>), dim3(256), dim3(512), 120 * 1024, 0, out, iters, ticks, fc, g_mem);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_ticks, ticks, sizeof h_ticks, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_fc, fc, sizeof h_fc, hipMemcpyDeviceToHost));
    const double cyc = median(h_ticks, 256);
    if (PRIO) printf("  [filler waves at s_setprio %d]", PRIO);
    printf("  other wave, kind %d, %d fillers per MFMA: MFMA waves %.1f cycles per MFMA, filler waves %.1f cycles per filler (in all %.1f per MFMA)\n", KIND, FPM, cyc / (iters * 16.0), median(h_fc, 256) / (iters * 16.0 * FPM), median(h_fc, 256) / (iters * 16.0));
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* out; CK(hipMalloc(&out, 256 * 512 * 4));
    unsigned long long *ticks, *fc; CK(hipMalloc(&ticks, 256 * 8)); CK(hipMalloc(&fc, 256 * 8));
    CK(hipFuncSetAttribute((const void*)same_wave<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CK(hipMalloc(&g_mem, (size_t)256 * 512 * 16));
    CK(hipMemset(g_mem, 0, (size_t)256 * 512 * 16));
    printf("warm-up\n");
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL((same_wave<0, 0>), dim3(256), dim3(256), 120 * 1024, 0, out, 20000, ticks, g_mem);   // ~1 s: settle the clock
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (argc > 1 && argv[1][0] == 't') {
        printf("two MFMA streams per SIMD (older waves 0-3, younger waves 4-7 at the given priority): cycles until each finishes, per MFMA of ONE stream\n");
        const int iters = 4000;
        auto run2 = [&](auto kern, int prio) {
            for (int l = 0; l < 3; ++l) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters, ticks, fc);
            CK(hipGetLastError()); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h_ticks, ticks, sizeof h_ticks, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_fc, fc, sizeof h_fc, hipMemcpyDeviceToHost));
            printf("  younger at prio %d: older stream done after %.1f, younger after %.1f cycles per MFMA\n", prio, median(h_ticks, 256) / (iters * 16.0), median(h_fc, 256) / (iters * 16.0));
        };
        run2(two_streams<0>, 0); run2(two_streams<1>, 1); run2(two_streams<3>, 3);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {
        printf("younger vector-ALU waves beside older MFMA waves, with and without raised priority\n");
        runB<0, 1>(out, ticks, fc); runB<0, 1, 3>(out, ticks, fc); runB<0, 4>(out, ticks, fc); runB<0, 4, 3>(out, ticks, fc); runB<0, 4, 1>(out, ticks, fc);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'm') {
        printf("memory instructions as fillers (same wave, then other wave)\n");
        runA<0, 0>(out, ticks);
        runA<10, 1>(out, ticks); runA<10, 2>(out, ticks); runA<10, 4>(out, ticks);
        runA<11, 1>(out, ticks); runA<11, 2>(out, ticks); runA<11, 4>(out, ticks);
        runA<12, 1>(out, ticks); runA<12, 2>(out, ticks);
        runA<13, 1>(out, ticks); runA<13, 2>(out, ticks);
        runA<14, 1>(out, ticks); runA<14, 2>(out, ticks);
        runB<10, 1>(out, ticks, fc); runB<10, 2>(out, ticks, fc); runB<11, 2>(out, ticks, fc); runB<12, 1>(out, ticks, fc); runB<13, 1>(out, ticks, fc); runB<14, 1>(out, ticks, fc);
        return 0;
    }
    printf("A: one wave per SIMD\n");
    runA<0, 0>(out, ticks);
    runA<0, 1>(out, ticks); runA<0, 2>(out, ticks); runA<0, 4>(out, ticks); runA<0, 8>(out, ticks); runA<0, 12>(out, ticks);
    runA<1, 2>(out, ticks); runA<1, 4>(out, ticks); runA<1, 8>(out, ticks);
    runA<2, 1>(out, ticks); runA<2, 2>(out, ticks); runA<2, 4>(out, ticks);
    runA<3, 2>(out, ticks); runA<3, 4>(out, ticks);
    runA<4, 1>(out, ticks); runA<4, 2>(out, ticks);
    runA<5, 4>(out, ticks); runA<5, 8>(out, ticks);
    runA<6, 2>(out, ticks); runA<6, 4>(out, ticks); runA<7, 2>(out, ticks); runA<7, 4>(out, ticks); runA<8, 2>(out, ticks); runA<9, 2>(out, ticks); runA<9, 4>(out, ticks);
    if (argc > 1) return 0;
    printf("B: MFMA waves beside filler waves on the same SIMDs\n");
    runB<0, 1>(out, ticks, fc); runB<0, 2>(out, ticks, fc); runB<0, 4>(out, ticks, fc); runB<0, 8>(out, ticks, fc);
    runB<2, 2>(out, ticks, fc); runB<6, 2>(out, ticks, fc); runB<6, 4>(out, ticks, fc); runB<4, 2>(out, ticks, fc);
    return 0;
}
