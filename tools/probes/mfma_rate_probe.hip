// Probe: sustained v_mfma_f32_32x32x2_f32 rate with W waves per SIMD and C independent accumulator chains per wave,
// with and without a workgroup barrier every 32 MFMAs (the k_gemm loop shape).  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, bool BARRIER> __global__ void k(float *out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 32 / CHAINS; i++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, bool BARRIER> void run(int threads, int blocks_per_cu, float *out) {
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, BARRIER>), dim3(blocks), dim3(threads), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, BARRIER>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 32.0 * (2.0 * 32 * 32 * 2);
    printf("threads/WG %4d  WG/CU %d  chains %d  barrier %d : %7.1f TFLOP/s\n", threads, blocks_per_cu, CHAINS, (int)BARRIER,
           flop / (ms * 1e-3) / 1e12);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS> __global__ void k4(float *out, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 128 / CHAINS; i++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 3, 5, 1);
        __syncthreads();
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++)
        for (int r = 0; r < 4; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS> void run4(float *out) {
    const int iters = 2000, blocks = 256, threads = 512;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k4<CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k4<CHAINS>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * 128.0 * (2.0 * 16 * 16);
    printf("4x4x1, 512 threads/WG, chains %d : %7.1f TFLOP/s (peak 155)\n", CHAINS, flop / (ms * 1e-3) / 1e12);
}
int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 4 * 1024 * sizeof(float));
    run<4, false>(256, 1, out);  // 1 wave per SIMD
    run<4, false>(512, 1, out);  // 2 waves per SIMD
    run<4, true>(512, 1, out);   // ... with the barrier
    run<4, false>(1024, 1, out); // 4 waves per SIMD
    run<4, true>(1024, 1, out);
    run<2, false>(512, 1, out);
    run<1, false>(512, 1, out);
    run<4, true>(512, 2, out);   // 2 WGs of 8 waves per CU
    run4<1>(out);
    run4<2>(out);
    run4<4>(out);
    run4<8>(out);
    return 0;
}
