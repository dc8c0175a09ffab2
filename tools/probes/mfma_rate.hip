// Diagnostic: issue cost of the MFMA shapes the recurrences can use, one wave per SIMD and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Each wave runs REP x 64 instructions on 8 independent accumulators (no dependent issue closer than 8 apart) and reports
// s_memtime cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int REP = 64;
template <int KIND> __global__ void k(unsigned long long *out, float *sink) {
    f32x4 c[8];
    f32x16 d[4];
    for (int i = 0; i < 8; i++) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) d[i][j] = 0.f;
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {4, 5, (short)threadIdx.x, 7};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; i++) a8[i] = (__bf16)(float)(threadIdx.x + i), b8[i] = (__bf16)(float)(i);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (KIND == 0) c[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[u], 4, 3, 0);
                if (KIND == 1) c[u] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, b4, c[u], 4, 3, 0);
                if (KIND == 2) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c[u], 0, 0, 0);
                if (KIND == 3) c[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[u], 0, 0, 0);
                if (KIND == 4) d[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, d[u & 3], 0, 0, 0);
                if (KIND == 5) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[u], 0, 0, 0);
                if (KIND == 6) d[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d[u & 3], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) s += d[i][j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (s == 12345.678f) *sink = s;
}
// 64 instructions on 64 distinct weight registers (the recurrences' pattern), 4 accumulators; waves >= 8 spin on an LDS word
template <int NCH, bool DIST> __global__ __launch_bounds__(768) void k_distinct(unsigned long long *out, float *sink, const s16x4 *wsrc) {
    __shared__ unsigned flag;
    if (threadIdx.x == 0) flag = 0;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    if (w >= 8) { // a polling wave per SIMD, as the elementwise / gating waves are most of the time
        __builtin_amdgcn_s_setprio(3);
        while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 8u) __builtin_amdgcn_s_sleep(1);
        return;
    }
    s16x4 wq[64];
#pragma unroll
    for (int i = 0; i < 64; i++) wq[i] = wsrc[i * 64 + (threadIdx.x & 63)];
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("" ::"v"(wq[i]));
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3};
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
#define WQ(i) wq[DIST ? (i) : 0]
#define Q8(i)                                                                   \
    c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 0), c0, 4, (i + 0) & 15, 0); \
    c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 1), c1, 4, (i + 1) & 15, 0); \
    c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 2), c2, 4, (i + 2) & 15, 0); \
    c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 3), c3, 4, (i + 3) & 15, 0); \
    c4 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 4), c4, 4, (i + 4) & 15, 0); \
    c5 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 5), c5, 4, (i + 5) & 15, 0); \
    c6 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 6), c6, 4, (i + 6) & 15, 0); \
    c7 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 7), c7, 4, (i + 7) & 15, 0);
        if (NCH == 8) { Q8(0) Q8(8) Q8(16) Q8(24) Q8(32) Q8(40) Q8(48) Q8(56) } else {
#define Q(i)                                                                   \
    c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 0), c0, 4, (i + 0) & 15, 0); \
    c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 1), c1, 4, (i + 1) & 15, 0); \
    c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 2), c2, 4, (i + 2) & 15, 0); \
    c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a4, WQ(i + 3), c3, 4, (i + 3) & 15, 0);
        Q(0) Q(4) Q(8) Q(12) Q(16) Q(20) Q(24) Q(28) Q(32) Q(36) Q(40) Q(44) Q(48) Q(52) Q(56) Q(60)
        }
#undef Q
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[0] + c6[0] + c7[0];
    if ((threadIdx.x & 63) == 0) {
        out[w] = t1 - t0;
        __hip_atomic_fetch_add(&flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (s == 12345.678f) *sink = s;
}
template <int KIND> void run(const char *name, unsigned long long *dout, float *dsink) {
    for (int waves : {4, 8, 12, 16}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * waves), 0, 0, dout, dsink);
        unsigned long long h[16];
        hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < waves; i++) m += (double)h[i];
        printf("%-28s %d waves/CU: %.1f memtime ticks per instruction per wave\n", name, waves, m / waves / (REP * 64.0));
    }
}
int main() {
    unsigned long long *dout;
    float *dsink;
    hipMalloc(&dout, 16 * 8);
    hipMalloc(&dsink, 4);
    run<0>("f32 4x4x1 (16 blocks)", dout, dsink);
    run<1>("bf16 4x4x4 (16 blocks)", dout, dsink);
    run<2>("bf16 16x16x32", dout, dsink);
    run<3>("bf16 16x16x16 (1k)", dout, dsink);
    run<4>("bf16 32x32x16", dout, dsink);
    run<5>("f32 16x16x4", dout, dsink);
    run<6>("f32 32x32x2", dout, dsink);
    s16x4 *wsrc;
    hipMalloc(&wsrc, 64 * 64 * 8);
    hipMemset(wsrc, 0, 64 * 64 * 8);
    for (int variant = 0; variant < 4; variant++)
    for (int waves : {4, 8, 12}) {
        if (variant == 0) hipLaunchKernelGGL((k_distinct<4, true>), dim3(1), dim3(64 * waves), 0, 0, dout, dsink, wsrc);
        if (variant == 1) hipLaunchKernelGGL((k_distinct<4, false>), dim3(1), dim3(64 * waves), 0, 0, dout, dsink, wsrc);
        if (variant == 2) hipLaunchKernelGGL((k_distinct<8, true>), dim3(1), dim3(64 * waves), 0, 0, dout, dsink, wsrc);
        if (variant == 3) hipLaunchKernelGGL((k_distinct<8, false>), dim3(1), dim3(64 * waves), 0, 0, dout, dsink, wsrc);
        unsigned long long h[16];
        hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        const int nw = waves < 8 ? waves : 8;
        for (int i = 0; i < nw; i++) m += (double)h[i];
        printf("bf16 4x4x4, %s weight registers, %d chains; %2d waves (8+ poll LDS): %.1f ticks per instruction per wave\n", (variant & 1) ? "one     " : "distinct", variant < 2 ? 4 : 8, waves, m / nw / (REP * 64.0));
    }
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("(ticks are s_memtime counts, about one per shader cycle at the stamped launches; shader clock attribute %d kHz)\n", clk);
    return 0;
}
