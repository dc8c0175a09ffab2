// gemm.hip -- the time-batched fp32 products of a window (R/lstm.cc:195,226,228,250) on v_mfma_f32_32x32x2_f32.
//
//   C[m + ldc*n] = sum_k opA(m,k) * opB(n,k)         column-major C, exact fp32 (an fma chain per output)
//
// Operand layouts (the four products of the window use three of the four combinations):
//   "k slow": X[k*ld + r]  -- rows of the product contiguous, one line of `ld` floats per k
//             (Why as stored; H, DG, dY as the contraction runs over the window's columns: dU, dWhy)
//   "k fast": X[r*ld + k]  -- k contiguous (H as the B operand of Y = Why*H; Why and dY in DHy = Why^T*dY)
//
// Structure: NO LDS in the main loop.  fp32 MFMA runs at 1/16 of the bf16 rate, so a 64 x 64 output tile per compute unit
// needs only ~16 bytes per clock of operands, which the L2 delivers straight into registers.  A workgroup owns one output
// tile; its waves split K among themselves (wave w takes the 8-deep k-groups w, w+NW, ...), each keeps the whole tile in
// accumulators and streams its own operand fragments with plain global loads DEPTH groups ahead -- no barrier, no staging
// pass, no cross-wave dependency until the end, where the NW partial tiles are summed through LDS in wave order
// (deterministic) and stored.  A lane's 32x32x2 fragment element is (row i = lane & 31, k-slot h = lane >> 5); within a
// k-group of 8 the four instructions use k = 4h + j (j = 0..3), the same on both operands, so
//   a k-slow operand is fetched as V consecutive rows per lane (one 4V-byte load per j: rows r0 + V*i + v),
//   a k-fast operand as 4 consecutive k per lane (one 16-byte load per 32-row subtile v: rows r0 + 32*v + i).
// Out-of-range rows are clamped to the last valid address (their products only reach outputs that are never stored);
// the k tail (K % 8, k-slow operands only) is one predicated group.
#include <cstdlib>
#include "kernels.h"

namespace lstmk {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef GEMM_CLOCK_STAMPS // diagnostic builds only (tools/probes/gemm_probe.hip): shader clock held during the main loop
__device__ unsigned long long g_gemm_stamps[2 * 4096];
__device__ unsigned long long g_gemm_timeline[512 * 8 * 4]; // [block][wave][start, loop end, kernel end] s_memrealtime (100 MHz)
#endif

template <int V> struct VecOf;
template <> struct VecOf<1> { using T = float; };
template <> struct VecOf<2> { using T = f32x2; };
template <> struct VecOf<4> { using T = f32x4; };

// One operand's fragments of one k-group: v[j][s] = element for instruction j (k = k0 + 4h + j), subtile s.
// X: the operand's base advanced to the group (wave-uniform); off: this lane's element offsets, rows already clamped --
// one per subtile for a k-fast operand (row (r0 + 32 s + i), k = 4h), one in all for a k-slow operand (row r0 + V i, k = 4h).
template <bool KFAST, int V> struct Frag {
    static constexpr int NOFF = KFAST ? V : 1;
    float v[4][V];
    __device__ __forceinline__ void load(const float *__restrict__ X, const unsigned (&off)[NOFF], int ld) {
        if constexpr (KFAST) {
#pragma unroll
            for (int s = 0; s < V; s++) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(X + off[s]);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j][s] = x[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const typename VecOf<V>::T x = *reinterpret_cast<const typename VecOf<V>::T *>(X + (size_t)j * ld + off[0]);
                if constexpr (V == 1) v[j][0] = x;
                else {
#pragma unroll
                    for (int s = 0; s < V; s++) v[j][s] = x[s];
                }
            }
        }
    }
    // one load unit: row j of a k-slow operand (all V subtiles), subtile u of a k-fast operand (all four j)
    __device__ __forceinline__ void load_unit(int u, const float *__restrict__ X, const unsigned (&off)[NOFF], int ld) {
        if constexpr (KFAST) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(X + off[u]);
#pragma unroll
            for (int j = 0; j < 4; j++) v[j][u] = x[j];
        } else {
            const typename VecOf<V>::T x = *reinterpret_cast<const typename VecOf<V>::T *>(X + (size_t)u * ld + off[0]);
            if constexpr (V == 1) v[u][0] = x;
            else {
#pragma unroll
                for (int s = 0; s < V; s++) v[u][s] = x[s];
            }
        }
    }
    // k tail of a k-slow operand: rows k0 + 4h + j >= K read nothing and contribute zero
    __device__ __forceinline__ void load_tail(const float *__restrict__ X, const unsigned (&off)[NOFF], int ld, int kfirst, int K) {
        static_assert(!KFAST, "k-fast operands need K % 8 == 0");
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int s = 0; s < V; s++) v[j][s] = (kfirst + j < K) ? X[(size_t)j * ld + off[0] + s] : 0.0f;
    }
};

template <bool KFAST, int V> struct Rows {
    static constexpr int TILE = 32 * V;
    static constexpr int NOFF = KFAST ? V : 1;
    // row of the product held by (fragment row i, subtile s)
    __device__ static __forceinline__ int row(int r0, int i, int s) { return KFAST ? r0 + 32 * s + i : r0 + V * i + s; }
    // lane offsets within a k-group (floats): k-slot h selects k = 4h; rows past the end are clamped to the last valid ones
    __device__ static __forceinline__ void offsets(unsigned (&off)[NOFF], int ld, int r0, int R, int i, int h) {
        if constexpr (KFAST) {
#pragma unroll
            for (int s = 0; s < V; s++) {
                int r = r0 + 32 * s + i;
                r = r < R ? r : R - 1;
                off[s] = (unsigned)r * (unsigned)ld + 4u * h;
            }
        } else {
            int r = r0 + V * i;
            r = r + V <= R ? r : R - V;
            off[0] = 4u * h * (unsigned)ld + (unsigned)r;
        }
    }
    __device__ static __forceinline__ size_t group_stride(int ld) { return KFAST ? (size_t)8 : (size_t)8 * ld; }
};

// AKF / BKF: operand is k-fast; VA, VB: its width (tile = 32*VA x 32*VB); NW waves split K; DEPTH groups in flight per wave
template <bool AKF, bool BKF, int VA, int VB, int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW) void k_gemm_regs(int M, int Nn, int K, const float *__restrict__ A, int lda,
                                                       const float *__restrict__ Bm, int ldb, float *__restrict__ C, int ldc,
                                                       int tiles_m, int tiles_n, int kchunk, size_t slab_stride) {
    constexpr int TM = 32 * VA, TN = 32 * VB;
    extern __shared__ __attribute__((aligned(16))) float red[]; // [NW][VB*4][VA][64] float4 pieces of the partial tiles
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = l & 31, h = l >> 5;
    // XCD-aware tile order (speed only): blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of
    // tiles, the dimension with fewer tiles running fastest, so that its workgroups share operand panels in its L2.
    const int ntile = tiles_m * tiles_n;
    int lin = blockIdx.x;
    {
        const int q = ntile >> 3, r = ntile & 7, x = lin & 7;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
    }
    int tm, tn;
    if (tiles_n <= tiles_m) {
        tn = lin % tiles_n;
        tm = lin / tiles_n;
    } else {
        tm = lin % tiles_m;
        tn = lin / tiles_m;
    }
    const int m0 = tm * TM, n0 = tn * TN;
    // split-K over workgroups (blockIdx.y): k range [kbeg, kend), kbeg a multiple of 8; slab z of the output
    const int kbeg = blockIdx.y * kchunk;
    const int kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    C += (size_t)blockIdx.y * slab_stride;

    const float *pa = A + (AKF ? (size_t)kbeg : (size_t)kbeg * lda);
    const float *pb = Bm + (BKF ? (size_t)kbeg : (size_t)kbeg * ldb);
    unsigned oa[Rows<AKF, VA>::NOFF], ob[Rows<BKF, VB>::NOFF];
    Rows<AKF, VA>::offsets(oa, lda, m0, M, i, h);
    Rows<BKF, VB>::offsets(ob, ldb, n0, Nn, i, h);
    const size_t sa = Rows<AKF, VA>::group_stride(lda), sb = Rows<BKF, VB>::group_stride(ldb);

    f32x16 acc[VA][VB];
#pragma unroll
    for (int a = 0; a < VA; a++)
#pragma unroll
        for (int b = 0; b < VB; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    // The four instructions of a k-group run back to back on ONE accumulator before the next accumulator's: a chain of
    // dependent 32x32x2 instructions keeps its accumulator inside the matrix pipe (64 cycles each, 99 % of peak), while
    // interleaved independent accumulators pay the register file for C in and D out every time (73 cycles each measured,
    // tools/probes/mfma_rate_probe.hip: "chains").
    auto product = [&](const Frag<AKF, VA> &fa, const Frag<BKF, VB> &fb) {
#pragma unroll
        for (int a = 0; a < VA; a++)
#pragma unroll
            for (int b = 0; b < VB; b++) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    // operands swapped: D[row <- B's row (n)][col <- A's row (m)], so a lane owns one m and stores run along m
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.v[j][b], fa.v[j][a], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    };

#ifdef GEMM_CLOCK_STAMPS
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    if (l == 0 && blockIdx.x < 512 && blockIdx.y == 0) g_gemm_timeline[(blockIdx.x * 8 + w) * 4 + 0] = st_r0;
#endif
    const int nfull = (kend - kbeg) >> 3;       // complete k-groups of this k range
    const int mine = (nfull - w + NW - 1) / NW; // ... of which this wave takes groups w, w + NW, ...
    if (nfull > 0) {
        // The loop body has no branch around a load: the compiler's counted waits (vmcnt) then leave the DEPTH - 1 younger
        // groups in flight.  Groups past the wave's last one re-read the range's last group (valid memory, results unused).
        auto group_of = [&](int n) {
            const int g = w + n * NW;
            return (unsigned)(g < nfull ? g : nfull - 1);
        };
        Frag<AKF, VA> fa[DEPTH];
        Frag<BKF, VB> fb[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const unsigned g = group_of(d);
            fa[d].load(pa + (size_t)(g * (unsigned)sa), oa, lda);
            fb[d].load(pb + (size_t)(g * (unsigned)sb), ob, ldb);
            __builtin_amdgcn_sched_barrier(0); // keep the groups' loads in issue order: the counted waits depend on it
        }
        for (int it = 0; it < mine; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                if (it + d < mine) product(fa[d], fb[d]);
                const unsigned g = group_of(it + d + DEPTH);
                fa[d].load(pa + (size_t)(g * (unsigned)sa), oa, lda);
                fb[d].load(pb + (size_t)(g * (unsigned)sb), ob, ldb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef GEMM_CLOCK_STAMPS
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
        g_gemm_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
        g_gemm_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    }
    if (l == 0 && blockIdx.x < 512 && blockIdx.y == 0) g_gemm_timeline[(blockIdx.x * 8 + w) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (!AKF && !BKF) {
        // the k tail: one wave, predicated rows
        if (((kend - kbeg) & 7) != 0 && w == nfull % NW) {
            Frag<AKF, VA> ta;
            Frag<BKF, VB> tb;
            const int kfirst = kbeg + nfull * 8 + 4 * h;
            ta.load_tail(pa + (size_t)nfull * sa, oa, lda, kfirst, kend);
            tb.load_tail(pb + (size_t)nfull * sb, ob, ldb, kfirst, kend);
            product(ta, tb);
        }
    }

    // ---- sum of the NW partial tiles through LDS, in wave order; every wave folds and stores its share ----
    // The tile is cut into VB*4 pieces c = (b, r4): registers 4*r4 .. 4*r4+3 of acc[.][b]; piece c belongs to wave c % NW.
    // A wave writes the pieces it does not own (one float4 per lane and subtile a), and after the barrier sums its own
    // pieces over the waves in order 0, 1, ... (its own term from registers): deterministic, one barrier.
    constexpr int PIECES = VB * 4;
    f32x4 *part = reinterpret_cast<f32x4 *>(red); // [NW][PIECES][VA][64]
#pragma unroll
    for (int c = 0; c < PIECES; c++) {
        if (c % NW != w) {
            const int b = c >> 2, r4 = c & 3;
#pragma unroll
            for (int a = 0; a < VA; a++) {
                f32x4 x;
#pragma unroll
                for (int e = 0; e < 4; e++) x[e] = acc[a][b][4 * r4 + e];
                part[(((size_t)w * PIECES + c) * VA + a) * 64 + l] = x;
            }
        }
    }
    __syncthreads();
#ifdef GEMM_CLOCK_STAMPS
    if (l == 0 && blockIdx.x < 512 && blockIdx.y == 0) g_gemm_timeline[(blockIdx.x * 8 + w) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int c = 0; c < PIECES; c++) {
        if (c % NW != w) continue;
        const int b = c >> 2, r4 = c & 3;
        f32x4 s[VA];
#pragma unroll
        for (int a = 0; a < VA; a++) {
#pragma unroll
            for (int ww = 0; ww < NW; ww++) {
                f32x4 x;
                if (ww == w) {
#pragma unroll
                    for (int e = 0; e < 4; e++) x[e] = acc[a][b][4 * r4 + e];
                } else
                    x = part[(((size_t)ww * PIECES + c) * VA + a) * 64 + l];
                if (ww == 0) s[a] = x;
                else {
#pragma unroll
                    for (int e = 0; e < 4; e++) s[a][e] += x[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int nrow = e + 8 * r4 + 4 * h; // row of D = fragment row of the B operand
            const int n = Rows<BKF, VB>::row(n0, nrow, b);
            if (n >= Nn) continue;
            float *cp = C + (size_t)n * ldc;
            if constexpr (!AKF && VA > 1) { // a k-slow A: VA consecutive floats per lane
                const int m = m0 + VA * i;
                if (m + VA <= M) {
                    typename VecOf<VA>::T x;
#pragma unroll
                    for (int a = 0; a < VA; a++) x[a] = s[a][e];
                    *reinterpret_cast<typename VecOf<VA>::T *>(cp + m) = x;
                }
            } else {
#pragma unroll
                for (int a = 0; a < VA; a++) {
                    const int m = Rows<AKF, VA>::row(m0, i, a);
                    if (m < M) cp[m] = s[a][e];
                }
            }
        }
    }
}

template <bool AKF, bool BKF, int VA, int VB, int NW, int DEPTH>
void launch_regs(int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int splits,
                 int kchunk, size_t slab_stride, hipStream_t st) {
    constexpr int TM = 32 * VA, TN = 32 * VB;
    const int tiles_m = (M + TM - 1) / TM, tiles_n = (Nn + TN - 1) / TN;
    const size_t lds = sizeof(float) * 4 * 64 * (size_t)NW * VB * 4 * VA;
    auto kern = k_gemm_regs<AKF, BKF, VA, VB, NW, DEPTH>;
    static bool attr_done = false; // per instantiation
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, splits), dim3(64 * NW), lds, st, M, Nn, K, A, lda, B, ldb, C, ldc, tiles_m,
                       tiles_n, kchunk, slab_stride);
}

} // namespace

// Shape rules (static: the same shape always takes the same kernel, the same split and the same summation order).
// One wave per SIMD (four-wave workgroups, one per compute unit): measured at the headline dU shape (tools/probes/
// gemm_probe.hip), two waves per SIMD streaming operands share the matrix pipe badly (the older wave takes it, 60 % busy
// while both run), and a compute unit sustains ~14 bytes per clock of such loads, which a 64 x 64 tile (16 B/clk at the
// full matrix rate) exceeds and a 128 x 64 tile (12 B/clk) does not: 128 x 64, four waves, two groups in flight.
//   k slow x k slow (dU, dWhy; K = window columns): split K over workgroups until the grid covers the chip
//   k slow x k fast (Y) and k fast x k fast (DHy, unfused paths): K is the hidden size or 256: no split
int gemm_regs_splits(bool akf, bool bkf, int M, int Nn, int K, int n_cus) {
    if (akf || bkf) return 1;
    const int tiles = ((M + 127) / 128) * ((Nn + 63) / 64);
    int splits = 1;
    while (tiles * splits * 2 <= n_cus && K / (splits * 2) >= 256) splits *= 2;
    return splits;
}

int gemm_regs(bool akf, bool bkf, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
              int splits, float *slabs, hipStream_t st) {
    if (splits < 1) splits = 1;
    int kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + 7) / 8 * 8;
    splits = (K + kchunk - 1) / kchunk;
    float *out = splits > 1 ? slabs : C;
    const int ldo = splits > 1 ? M : ldc;
    const size_t stride = splits > 1 ? (size_t)M * Nn : 0;
    // 64 x 64 tiles for the small products of a narrow batch, where 128 x 64 tiles (times the split-K factor) would leave most
    // of the chip without one (LSTM_HIP_GEMM_SMALL_TILES=0: A/B)
    static const bool small_ok = !(getenv("LSTM_HIP_GEMM_SMALL_TILES") && atoi(getenv("LSTM_HIP_GEMM_SMALL_TILES")) == 0);
    const bool small = small_ok && ((M + 127) / 128) * ((Nn + 63) / 64) * splits < 128;
    if (!akf && !bkf && small) launch_regs<false, false, 2, 2, 4, 2>(M, Nn, K, A, lda, B, ldb, out, ldo, splits, kchunk, stride, st);
    else if (!akf && !bkf) launch_regs<false, false, 4, 2, 4, 2>(M, Nn, K, A, lda, B, ldb, out, ldo, splits, kchunk, stride, st);
    else if (!akf && bkf && small) launch_regs<false, true, 2, 2, 4, 2>(M, Nn, K, A, lda, B, ldb, out, ldo, splits, kchunk, stride, st);
    else if (!akf && bkf) launch_regs<false, true, 4, 2, 4, 2>(M, Nn, K, A, lda, B, ldb, out, ldo, splits, kchunk, stride, st);
    else if (akf && bkf) launch_regs<true, true, 2, 2, 4, 2>(M, Nn, K, A, lda, B, ldb, out, ldo, splits, kchunk, stride, st);
    else return -1; // k fast x k slow: no product of the window has this form
    return splits;
}

// timing probe only (LSTM_HIP_PROBE_OVERLAP): the Y-shaped product on 64 x 64 tiles -- 64 KB of LDS and ~100 registers, small
// enough to be co-resident with a workgroup of the forward recurrence on the same compute unit
void gemm_probe_small_kfast(int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, hipStream_t st) {
    launch_regs<false, true, 2, 2, 4, 2>(M, Nn, K, A, lda, B, ldb, C, ldc, 1, (K + 7) / 8 * 8, 0, st);
}

// the (TA, TB) form the rest of the library speaks: op(A) is k fast when TA, op(B) is k fast when !TB
void gemm(bool TA, bool TB, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int splits,
          float *slabs, hipStream_t st) {
    const int used = gemm_regs(TA, !TB, M, Nn, K, A, lda, B, ldb, C, ldc, splits, slabs, st);
    if (used > 1) gemm_fold(slabs, used, M, Nn, C, ldc, st, 0);
}
int gemm_pick_splits(bool TA, bool TB, int M, int Nn, int K, int n_cus) { return gemm_regs_splits(TA, !TB, M, Nn, K, n_cus); }
int gemm_slabs(bool TA, bool TB, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *slabs, int splits,
               hipStream_t st) {
    return gemm_regs(TA, !TB, M, Nn, K, A, lda, B, ldb, slabs, M, splits, slabs, st); // one slab: the product itself, ld = M
}

} // namespace lstmk
