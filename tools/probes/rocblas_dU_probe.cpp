// Timing probe: the dU product (4N x N x T = 2048 x 512 x 6336, C = A * B^T, fp32) through rocBLAS and hipBLASLt-backed
// rocBLAS paths, to compare with k_gemm (118 us).  hipcc -O2 rocblas_dU_probe.cpp -lrocblas -o rocblas_dU_probe
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
int main() {
    const int M = 2048, N = 512, K = 6336;
    float *A, *B, *C;
    hipMalloc(&A, sizeof(float) * M * K);
    hipMalloc(&B, sizeof(float) * N * K);
    hipMalloc(&C, sizeof(float) * (size_t)256 * K * 2); // >= max(M*N, 256*K) floats: both products write into it
    std::vector<float> h((size_t)M * K, 0.01f);
    hipMemcpy(A, h.data(), sizeof(float) * M * K, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), sizeof(float) * N * K, hipMemcpyHostToDevice);
    rocblas_handle hd;
    rocblas_create_handle(&hd);
    rocblas_set_atomics_mode(hd, rocblas_atomics_not_allowed); // deterministic reductions only
    const float one = 1.0f, zero = 0.0f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 5; i++)
            rocblas_sgemm(hd, rocblas_operation_none, rocblas_operation_transpose, M, N, K, &one, A, M, B, N, &zero, C, M);
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++)
            rocblas_sgemm(hd, rocblas_operation_none, rocblas_operation_transpose, M, N, K, &one, A, M, B, N, &zero, C, M);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        fflush(stdout);
        printf("rocblas_sgemm NT %dx%dx%d: %.1f us, %.1f TFLOP/s\n", M, N, K, ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
    }
    // Y = Why * H: 256 x 6336 x 512 (NN)
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++)
            rocblas_sgemm(hd, rocblas_operation_none, rocblas_operation_none, 256, K, 512, &one, A, 256, B, 512, &zero, C, 256);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("rocblas_sgemm NN 256x%dx512: %.1f us\n", K, ms / 20 * 1e3);
    }
    // run-to-run bit identity of the dU product (random-ish data)
    {
        std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
        unsigned x = 12345;
        for (auto &v : ha) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
        for (auto &v : hb) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
        hipMemcpy(A, ha.data(), sizeof(float) * M * K, hipMemcpyHostToDevice);
        hipMemcpy(B, hb.data(), sizeof(float) * N * K, hipMemcpyHostToDevice);
        std::vector<float> c1((size_t)M * N), c2((size_t)M * N);
        rocblas_sgemm(hd, rocblas_operation_none, rocblas_operation_transpose, M, N, K, &one, A, M, B, N, &zero, C, M);
        hipMemcpy(c1.data(), C, sizeof(float) * M * N, hipMemcpyDeviceToHost);
        int diff = 0;
        for (int rep = 0; rep < 5; rep++) {
            rocblas_sgemm(hd, rocblas_operation_none, rocblas_operation_transpose, M, N, K, &one, A, M, B, N, &zero, C, M);
            hipMemcpy(c2.data(), C, sizeof(float) * M * N, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < c1.size(); i++) diff += c1[i] != c2[i];
        }
        printf("bitwise differences over 5 repeats: %d\n", diff);
    }
    return 0;
}
