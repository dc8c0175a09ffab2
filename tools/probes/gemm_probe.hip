// gemm_probe.hip -- correctness and timing of the register-streamed fp32 products (csrc/gemm.hip) at the window's shapes,
// with rocBLAS beside them as a yardstick only (the library itself does not link or load it).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm_probe.hip -lrocblas -o tools/probes/gemm_probe
//   tools/probes/gemm_probe [N S B]      (default 512 100 64)
#include "../../eigen-lstm_amd/csrc/gemm.hip"

#include <rocblas/rocblas.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

using namespace lstmk;
// the library's ordered slab fold lives in kernels.hip; the probe sums slabs on the host and never calls gemm()
void lstmk::gemm_fold(const float *, int, int, int, float *, int, hipStream_t, size_t) { abort(); }

static std::vector<float> rnd(size_t n, unsigned seed) {
    std::vector<float> v(n);
    unsigned x = seed;
    for (auto &e : v) {
        x = x * 1664525u + 1013904223u;
        e = ((x >> 8) & 0xffff) / 65536.0f - 0.5f;
    }
    return v;
}

struct Dev {
    float *p = nullptr;
    explicit Dev(size_t n) { hipMalloc(&p, n * sizeof(float)); }
    ~Dev() { hipFree(p); }
};

static float time_us(const std::function<void()> &fn, int reps = 30) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) fn();
    float best = 1e30f, sum = 0;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        for (int i = 0; i < reps; i++) fn();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms / reps * 1e3f);
        sum += ms / reps * 1e3f;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best;
}

// opA(m,k): akf ? A[m*lda + k] : A[k*lda + m]
static double check(bool akf, bool bkf, int M, int Nn, int K, const std::vector<float> &A, int lda, const std::vector<float> &B,
                    int ldb, const std::vector<float> &C, int ldc, int samples) {
    double worst = 0;
    unsigned x = 99;
    const bool all = (size_t)M * Nn <= (size_t)samples;
    const size_t cnt = all ? (size_t)M * Nn : samples;
    for (size_t s = 0; s < cnt; s++) {
        int m, n;
        if (all) {
            m = s % M;
            n = s / M;
        } else {
            x = x * 1664525u + 1013904223u;
            m = (x >> 8) % M;
            x = x * 1664525u + 1013904223u;
            n = (x >> 8) % Nn;
            if (s < 8) { m = (s & 1) ? M - 1 : 0; n = (s & 2) ? Nn - 1 : 0; }
        }
        double ref = 0, mag = 0;
        for (int k = 0; k < K; k++) {
            const double a = akf ? A[(size_t)m * lda + k] : A[(size_t)k * lda + m];
            const double b = bkf ? B[(size_t)n * ldb + k] : B[(size_t)k * ldb + n];
            ref += a * b;
            mag += std::fabs(a * b);
        }
        const double err = std::fabs(C[(size_t)n * ldc + m] - ref) / (mag + 1e-30);
        worst = std::max(worst, err);
    }
    return worst;
}

template <bool AKF, bool BKF, int VA, int VB, int NW, int DEPTH>
static void variant(const char *name, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                    int splits, float *slabs) {
    int kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + 7) / 8 * 8;
    splits = (K + kchunk - 1) / kchunk;
    float *out = splits > 1 ? slabs : C;
    const int ldo = splits > 1 ? M : ldc;
    const size_t stride = splits > 1 ? (size_t)M * Nn : 0;
    const float us = time_us([&]() {
        launch_regs<AKF, BKF, VA, VB, NW, DEPTH>(M, Nn, K, A, lda, B, ldb, out, ldo, splits, kchunk, stride, nullptr);
    });
    double mhz = 0;
#ifdef GEMM_CLOCK_STAMPS
    {
        std::vector<unsigned long long> st(2 * 4096);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_gemm_stamps), st.size() * 8);
        const int nb = std::min(4096, ((M + 32 * VA - 1) / (32 * VA)) * ((Nn + 32 * VB - 1) / (32 * VB)));
        std::vector<double> f, cyc;
        for (int b = 0; b < nb; b++)
            if (st[2 * b + 1]) f.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0), cyc.push_back((double)st[2 * b]);
        std::sort(f.begin(), f.end());
        std::sort(cyc.begin(), cyc.end());
        if (!f.empty()) mhz = f[f.size() / 2];
        if (!cyc.empty())
            printf("    main loop cycles of wave 0: min %.0f median %.0f max %.0f; MFMA-issue floor %.0f (x%d waves per SIMD)\n", cyc.front(),
                   cyc[cyc.size() / 2], cyc.back(), (double)((kchunk / 8 + NW - 1) / NW) * 4.0 * VA * VB * 64.0, NW > 4 ? NW / 4 : 1);
    }
#endif
#ifdef GEMM_CLOCK_STAMPS
    if (getenv("GEMM_PROBE_TIMELINE")) {
        std::vector<unsigned long long> tl(512 * 8 * 4);
        hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_gemm_timeline), tl.size() * 8);
        const int nb = std::min(512, ((M + 32 * VA - 1) / (32 * VA)) * ((Nn + 32 * VB - 1) / (32 * VB)));
        unsigned long long t0 = ~0ull, tend = 0;
        for (int b = 0; b < nb; b++)
            for (int w = 0; w < NW; w++) t0 = std::min(t0, tl[(b * 8 + w) * 4]), tend = std::max(tend, tl[(b * 8 + w) * 4 + 2]);
        printf("    timeline (us after the first wave's start; last reduce end %.2f)\n", (tend - t0) * 0.01);
        for (int b : {0, 1, 8, 100, nb - 1}) {
            printf("      block %3d:", b);
            for (int w = 0; w < NW; w++)
                printf(" w%d %.1f/%.1f/%.1f", w, (tl[(b * 8 + w) * 4] - t0) * 0.01, (tl[(b * 8 + w) * 4 + 1] - t0) * 0.01,
                       (tl[(b * 8 + w) * 4 + 2] - t0) * 0.01);
            printf("\n");
        }
    }
#endif
    printf("  %-34s tile %3dx%-3d waves %d depth %d splits %d: %7.1f us  %6.1f TFLOP/s  clock %.0f MHz\n", name, 32 * VA, 32 * VB, NW, DEPTH,
           splits, us, 2.0 * M * Nn * K / us / 1e6, mhz);
    fflush(stdout);
}

static void one_product(const char *what, bool akf, bool bkf, int M, int Nn, int K, rocblas_handle hd) {
    if (getenv("GEMM_PROBE_ONLY") && !strstr(what, getenv("GEMM_PROBE_ONLY"))) return;
    const int pad = getenv("GEMM_PROBE_PAD") ? atoi(getenv("GEMM_PROBE_PAD")) : 0; // leading dimensions off the powers of two
    const int lda = (akf ? K : M) + pad, ldb = (bkf ? K : Nn) + pad, ldc = M;
    const size_t nA = (size_t)lda * (akf ? M : K), nB = (size_t)ldb * (bkf ? Nn : K);
    const std::vector<float> hA = rnd(nA, 1), hB = rnd(nB, 2);
    Dev dA(nA), dB(nB), dC((size_t)M * Nn), dS((size_t)32 * M * Nn); // room for any split the shape rule picks
    hipMemcpy(dA.p, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB.p, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    printf("%s: C[%d x %d] = opA * opB, K = %d (A %s, B %s)\n", what, M, Nn, K, akf ? "k fast" : "k slow", bkf ? "k fast" : "k slow");
    // the library's own choice: correctness
    hipMemset(dC.p, 0xff, (size_t)M * Nn * 4);
    const int splits = gemm_regs_splits(akf, bkf, M, Nn, K, 256);
    if (splits > 32) { printf("  split %d > 32 slabs allocated\n", splits); return; }
    const int used = gemm_regs(akf, bkf, M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, splits, dS.p, nullptr);
    hipDeviceSynchronize();
    std::vector<float> hC((size_t)M * Nn);
    if (used > 1) {
        std::vector<float> hS((size_t)used * M * Nn);
        hipMemcpy(hS.data(), dS.p, hS.size() * 4, hipMemcpyDeviceToHost);
        for (size_t e = 0; e < hC.size(); e++) {
            float s = hS[e];
            for (int z = 1; z < used; z++) s += hS[(size_t)z * M * Nn + e];
            hC[e] = s;
        }
    } else
        hipMemcpy(hC.data(), dC.p, hC.size() * 4, hipMemcpyDeviceToHost);
    const double err = check(akf, bkf, M, Nn, K, hA, lda, hB, ldb, hC, ldc, 4096);
    printf("  gemm_regs (splits %d): max |err| / sum|a b| = %.2e  %s\n", used, err, err < 3e-6 ? "ok" : "WRONG");
    const float us = time_us([&]() { gemm_regs(akf, bkf, M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, splits, dS.p, nullptr); });
    printf("  %-34s %7.1f us  %6.1f TFLOP/s\n", "gemm_regs (library rule)", us, 2.0 * M * Nn * K / us / 1e6);
    if (hd) {
        const float one = 1.0f, zero = 0.0f;
        const float rb = time_us([&]() {
            rocblas_sgemm(hd, akf ? rocblas_operation_transpose : rocblas_operation_none,
                          bkf ? rocblas_operation_none : rocblas_operation_transpose, M, Nn, K, &one, dA.p, lda, dB.p, ldb, &zero, dC.p,
                          ldc);
        });
        printf("  %-34s %7.1f us  %6.1f TFLOP/s\n", "rocblas_sgemm (yardstick)", rb, 2.0 * M * Nn * K / rb / 1e6);
    }
    if (getenv("GEMM_PROBE_VARIANTS")) {
        if (!akf && !bkf) {
            variant<false, false, 2, 2, 4, 4>("kslow x kslow", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, false, 4, 2, 4, 2>("kslow x kslow", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 2, dS.p);
            variant<false, false, 4, 2, 4, 3>("kslow x kslow", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 2, dS.p);
            variant<false, false, 2, 4, 4, 2>("kslow x kslow", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 2, dS.p);
        } else if (!akf && bkf) {
            variant<false, true, 2, 2, 4, 2>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 2, 2, 4, 3>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 2, 2, 4, 4>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 2, 2, 8, 2>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 2, 2, 8, 4>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 2, 1, 4, 4>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 2, 1, 2, 4>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 4, 1, 4, 3>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
            variant<false, true, 4, 2, 4, 2>("kslow x kfast", M, Nn, K, dA.p, lda, dB.p, ldb, dC.p, ldc, 1, dS.p);
        }
    }
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const int N = argc > 3 ? atoi(argv[1]) : 512, S = argc > 3 ? atoi(argv[2]) : 100, B = argc > 3 ? atoi(argv[3]) : 64;
    const int T = (S - 1) * B;
    rocblas_handle hd = nullptr;
    if (!getenv("GEMM_PROBE_NO_ROCBLAS")) {
        rocblas_create_handle(&hd);
        rocblas_set_atomics_mode(hd, rocblas_atomics_not_allowed);
    }
    printf("window N=%d S=%d B=%d (T=%d)\n", N, S, B, T);
    one_product("dU   = DG * H^T   (R/lstm.cc:250)", false, false, 4 * N, N, T, hd);
    one_product("Y    = Why * H    (R/lstm.cc:195)", false, true, 256, T, N, hd);
    one_product("dWhy = dY * H^T   (R/lstm.cc:226)", false, false, 256, N, T, hd);
    one_product("DHy  = Why^T * dY (R/lstm.cc:228)", true, true, N, T, 256, hd);
    // ragged shapes: rows and k tails that no tile divides
    one_product("ragged kslow x kslow", false, false, 64, 16, 37, nullptr);
    one_product("ragged kslow x kslow", false, false, 192, 48, 5, nullptr);
    one_product("ragged kslow x kfast", false, true, 256, 3, 48, nullptr);
    one_product("ragged kslow x kfast", false, true, 256, 77, 16, nullptr);
    one_product("ragged kfast x kfast", true, true, 48, 9, 256, nullptr);
    one_product("ragged kfast x kfast", true, true, 16, 100, 256, nullptr);
    return 0;
}
