// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the LSTM training window.
//
// Reference semantics restated by each kernel are cited as R/ (= /root/reference) file:line.
// All matrices are column-major fp32.  Gate row order is [i; o; f; u] (R/lstm.cc:77).
//
// MFMA fragment maps used below (cdna_hip_programming.md section 3):
//   v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+reg][col=l&15]
//   v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(reg&3)+8*(reg>>2)+4*(l>>5)][col=l&31]
#include "kernels.h"

namespace lstmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// scalar helpers (R/lstm.cc:30-48).  fp contraction is off so that i*u + f*c rounds like the
// reference's separate multiply and add.
// ------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
template <bool FAST> __device__ __forceinline__ float sigm(float x) {
    if (FAST) return __frcp_rn(1.0f + __expf(-x));
    return 1.0f / (1.0f + expf(-x));
}
template <bool FAST> __device__ __forceinline__ float tanh_(float x) {
    if (FAST) return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f);
    return tanhf(x);
}
__device__ __forceinline__ float tanh_prime(float x) { return 1.0f - x * x; }
__device__ __forceinline__ float logistic_prime(float x) { return x * (1.0f - x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// pack_U: build the MFMA A-fragment images of U (4N x N) for both recurrences.
//   Ufwd[jb][k4][l].i = U[(l&3)*N + 4*jb + ((l&15)>>2)][16*k4 + 4*(l>>4) + i]
//        tile rows are ordered (hidden unit, gate) so that one lane ends up holding i,o,f,u of ONE
//        hidden unit in its four accumulator registers (row = 4*(l>>4) + reg  ->  reg = gate).
//   Ubwd[kb][r4][l].i = U[16*r4 + 4*(l>>4) + i][16*kb + (l&15)]          (A = U^T, 16 hidden per tile)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_U(const float *__restrict__ U, float4 *__restrict__ Ufwd,
                                                float4 *__restrict__ Ubwd, int N) {
    const int G4 = 4 * N;
    const size_t nf4 = (size_t)N * N; // float4 count of each image (4N*N floats)
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < 2 * nf4; e += (size_t)gridDim.x * blockDim.x) {
        if (e < nf4) {
            int l = (int)(e & 63);
            size_t q = e >> 6;
            int k4 = (int)(q % (N / 16)), jb = (int)(q / (N / 16));
            int row = (l & 3) * N + 4 * jb + ((l & 15) >> 2);
            int k = 16 * k4 + 4 * (l >> 4);
            float4 v;
            v.x = U[(size_t)(k + 0) * G4 + row];
            v.y = U[(size_t)(k + 1) * G4 + row];
            v.z = U[(size_t)(k + 2) * G4 + row];
            v.w = U[(size_t)(k + 3) * G4 + row];
            Ufwd[e] = v;
        } else {
            size_t e2 = e - nf4;
            int l = (int)(e2 & 63);
            size_t q = e2 >> 6;
            int r4 = (int)(q % (N / 4)), kb = (int)(q / (N / 4));
            int r = 16 * r4 + 4 * (l >> 4);
            int k = 16 * kb + (l & 15);
            Ubwd[e2] = *reinterpret_cast<const float4 *>(U + (size_t)k * G4 + r);
        }
    }
}
void pack_U(const float *U, float4 *Ufwd, float4 *Ubwd, int N, hipStream_t st) {
    size_t n = 2 * (size_t)N * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_U, dim3(blocks), dim3(256), 0, st, U, Ufwd, Ubwd, N);
}

// ------------------------------------------------------------------------------------------------
// fwd_step (baseline engine): one timestep.  Workgroup jb owns hidden units 4*jb..4*jb+3, i.e. one
// 16-row MFMA tile holding their i,o,f,u rows; wave w takes batch-column tiles w, w+4, ...
//   g = W*x + U*h_prev + b          R/lstm.cc:176  (W*x is a column gather: x is one-hot or empty)
//   i,o,f = sigm ; u = tanh         R/lstm.cc:179-182
//   c = tanh(i*u + f*c_prev)        R/lstm.cc:185-189
//   h = o*c                         R/lstm.cc:192
// ------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256) void k_fwd_step(const float4 *__restrict__ Ufwd, const float *__restrict__ W,
                                                  const float *__restrict__ bias, const float *__restrict__ Hprev,
                                                  const float *__restrict__ Cprev, float *__restrict__ Hout,
                                                  float *__restrict__ Cout, float *__restrict__ Gout,
                                                  const int32_t *__restrict__ xi_t, int N, int B) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int jb = blockIdx.x;
    const int nk4 = N / 16, G4 = 4 * N;
    const int nct = (B + 15) / 16;
    const float4 *Ua = Ufwd + (size_t)jb * nk4 * 64 + l;
    constexpr int CH = 8; // k4-steps per operand chunk: 16 float4 in flight per lane
    for (int ct = w; ct < nct; ct += 4) {
        const int col = ct * 16 + (l & 15);
        const int colc = col < B ? col : B - 1;
        const float *hp = Hprev + (size_t)colc * N + 4 * (l >> 4);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float4 a[CH], b[CH], an[CH], bn[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int k4 = i < nk4 ? i : nk4 - 1;
            a[i] = Ua[(size_t)k4 * 64];
            b[i] = *reinterpret_cast<const float4 *>(hp + 16 * k4);
        }
        for (int c0 = 0; c0 < nk4; c0 += CH) {
            const bool more = c0 + CH < nk4;
            if (more) {
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    const int k4 = c0 + CH + i < nk4 ? c0 + CH + i : nk4 - 1;
                    an[i] = Ua[(size_t)k4 * 64];
                    bn[i] = *reinterpret_cast<const float4 *>(hp + 16 * k4);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (c0 + i < nk4) {
                    if (i & 1) {
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc1, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc0, 0, 0, 0);
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    a[i] = an[i];
                    b[i] = bn[i];
                }
            }
        }
        if (col < B) {
            const int j = 4 * jb + (l >> 4);
            const int x = xi_t[col];
            float pre[4];
#pragma unroll
            for (int gte = 0; gte < 4; gte++) {
                const int row = gte * N + j;
                float wx = x >= 0 ? W[(size_t)x * G4 + row] : 0.0f;
                pre[gte] = (wx + (acc0[gte] + acc1[gte])) + bias[row];
            }
            const float ig = sigm<FAST>(pre[0]), og = sigm<FAST>(pre[1]), fg = sigm<FAST>(pre[2]);
            const float ug = tanh_<FAST>(pre[3]);
            const float cp = Cprev[(size_t)col * N + j];
            const float c = tanh_<FAST>(ig * ug + fg * cp);
            const float hval = og * c;
            float *gc = Gout + (size_t)col * G4 + j;
            gc[0] = ig;
            gc[N] = og;
            gc[2 * N] = fg;
            gc[3 * N] = ug;
            Cout[(size_t)col * N + j] = c;
            Hout[(size_t)col * N + j] = hval;
        }
    }
}
void fwd_step(const float4 *Ufwd, const float *W, const float *bias, const float *Hprev, const float *Cprev, float *Hout,
              float *Cout, float *Gout, const int32_t *xi_t, int N, int B, bool fast, hipStream_t st) {
    if (fast)
        hipLaunchKernelGGL(k_fwd_step<true>, dim3(N / 4), dim3(256), 0, st, Ufwd, W, bias, Hprev, Cprev, Hout, Cout, Gout,
                           xi_t, N, B);
    else
        hipLaunchKernelGGL(k_fwd_step<false>, dim3(N / 4), dim3(256), 0, st, Ufwd, W, bias, Hprev, Cprev, Hout, Cout,
                           Gout, xi_t, N, B);
}

// ------------------------------------------------------------------------------------------------
// bwd_step (baseline engine): one BPTT step t.  Workgroup (kb, ct) owns hidden units 16*kb..+15 for
// batch columns 16*ct..+15; its four waves split the K = 4N contraction of dhnext = U^T * dg[t+1]
// (R/lstm.cc:255) and reduce through LDS; then one thread per (hidden, column):
//   dh = Why^T*dy + dhnext                           R/lstm.cc:228   (Why^T*dy arrives as DHy_t)
//   dc = (dh*o + dcnext) * (1 - c^2)                 R/lstm.cc:233-235
//   do,di,df,du and their nonlinearity derivatives   R/lstm.cc:238-247
//   dcnext = dc * f                                  R/lstm.cc:256
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bwd_step(const float4 *__restrict__ Ubwd, const float *__restrict__ DGnext,
                                                  const float *__restrict__ DHy_t, const float *__restrict__ G_t,
                                                  const float *__restrict__ C_t, const float *__restrict__ Cprev,
                                                  float *__restrict__ dcnext, float *__restrict__ DG_t, int N, int B) {
    __shared__ float red[4 * 4 * 64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int kb = blockIdx.x, ct = blockIdx.y;
    const int G4 = 4 * N, nr4 = N / 4; // 4N/16 k-steps of 16
    if (DGnext != nullptr) {
        const int col = ct * 16 + (l & 15);
        const int colc = col < B ? col : B - 1;
        const int per = nr4 / 4;
        const float4 *Ua = Ubwd + ((size_t)kb * nr4 + (size_t)w * per) * 64 + l;
        const float *dgp = DGnext + (size_t)colc * G4 + 16 * (w * per) + 4 * (l >> 4);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        constexpr int CH = 8;
        float4 a[CH], b[CH], an[CH], bn[CH];
#pragma unroll
        for (int i = 0; i < CH; i++) {
            const int r4 = i < per ? i : per - 1;
            a[i] = Ua[(size_t)r4 * 64];
            b[i] = *reinterpret_cast<const float4 *>(dgp + 16 * r4);
        }
        for (int c0 = 0; c0 < per; c0 += CH) {
            const bool more = c0 + CH < per;
            if (more) {
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    const int r4 = c0 + CH + i < per ? c0 + CH + i : per - 1;
                    an[i] = Ua[(size_t)r4 * 64];
                    bn[i] = *reinterpret_cast<const float4 *>(dgp + 16 * r4);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (c0 + i < per) {
                    if (i & 1) {
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc1, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc1, 0, 0, 0);
                    } else {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i].x, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i].y, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i].z, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i].w, acc0, 0, 0, 0);
                    }
                }
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < CH; i++) {
                    a[i] = an[i];
                    b[i] = bn[i];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) red[(w * 4 + r) * 64 + l] = acc0[r] + acc1[r];
    }
    __syncthreads();
    // thread e -> hidden jj = e&15 (contiguous in memory), column cc = e>>4
    const int e = threadIdx.x, jj = e & 15, cc = e >> 4;
    const int col = ct * 16 + cc, j = kb * 16 + jj;
    if (col >= B) return;
    float dhnext = 0.0f;
    if (DGnext != nullptr) {
        const int src = (jj >> 2) * 16 + cc, reg = jj & 3;
        dhnext = ((red[(0 * 4 + reg) * 64 + src] + red[(1 * 4 + reg) * 64 + src]) + red[(2 * 4 + reg) * 64 + src]) +
                 red[(3 * 4 + reg) * 64 + src];
    }
    const size_t o = (size_t)col * N + j;
    const float *gc = G_t + (size_t)col * G4 + j;
    const float ig = gc[0], og = gc[N], fg = gc[2 * N], ug = gc[3 * N];
    const float c = C_t[o], cp = Cprev[o];
    const float dh = DHy_t[o] + dhnext;
    float dcv = dh * og + dcnext[o];
    dcv = dcv * tanh_prime(c);
    float *dg = DG_t + (size_t)col * G4 + j;
    dg[N] = (dh * c) * logistic_prime(og);
    dg[0] = (dcv * ug) * logistic_prime(ig);
    dg[2 * N] = (dcv * cp) * logistic_prime(fg);
    dg[3 * N] = (dcv * ig) * tanh_prime(ug);
    dcnext[o] = dcv * fg;
}
void bwd_step(const float4 *Ubwd, const float *DGnext, const float *DHy_t, const float *G_t, const float *C_t,
              const float *Cprev, float *dcnext, float *DG_t, int N, int B, hipStream_t st) {
    hipLaunchKernelGGL(k_bwd_step, dim3(N / 16, (B + 15) / 16), dim3(256), 0, st, Ubwd, DGnext, DHy_t, G_t, C_t, Cprev,
                       dcnext, DG_t, N, B);
}

// ------------------------------------------------------------------------------------------------
// gemm: C = op(A) * op(B), fp32 MFMA 32x32x2, 128x128x16 LDS tiles, 4 waves each 64x64.
// The MFMA is issued with the operands swapped (A-operand <- op(B) column index, B-operand <- op(A)
// row index) so the accumulator holds C^T fragments: lanes then run along m, which is contiguous in
// column-major C, and the epilogue stores are coalesced.
// ------------------------------------------------------------------------------------------------
constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = 132;

template <bool TRANS> // TRANS=false: source is [rows contiguous] x K ; TRANS=true: source is K-contiguous
__device__ __forceinline__ void gemm_load(const float *__restrict__ src, int ld, int r0, int rmax, int k0, int kend,
                                          int tid, float4 (&reg)[2]) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (!TRANS) {
            const int r = r0 + (tid & 31) * 4, k = k0 + (tid >> 5) + q * 8;
            if (k < kend) {
                const float *p = src + (size_t)k * ld + r;
                if (r + 3 < rmax) v = *reinterpret_cast<const float4 *>(p);
                else {
                    if (r < rmax) v.x = p[0];
                    if (r + 1 < rmax) v.y = p[1];
                    if (r + 2 < rmax) v.z = p[2];
                }
            }
        } else {
            const int k = k0 + (tid & 3) * 4, r = r0 + (tid >> 2) + q * 64;
            if (r < rmax) {
                const float *p = src + (size_t)r * ld + k;
                if (k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);
                else {
                    if (k < kend) v.x = p[0];
                    if (k + 1 < kend) v.y = p[1];
                    if (k + 2 < kend) v.z = p[2];
                }
            }
        }
        reg[q] = v;
    }
}
template <bool TRANS> __device__ __forceinline__ void gemm_store(float *lds, int tid, const float4 (&reg)[2]) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (!TRANS) {
            const int r = (tid & 31) * 4, k = (tid >> 5) + q * 8;
            *reinterpret_cast<float4 *>(lds + k * GLD + r) = reg[q];
        } else {
            const int k = (tid & 3) * 4, r = (tid >> 2) + q * 64;
            lds[(k + 0) * GLD + r] = reg[q].x;
            lds[(k + 1) * GLD + r] = reg[q].y;
            lds[(k + 2) * GLD + r] = reg[q].z;
            lds[(k + 3) * GLD + r] = reg[q].w;
        }
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm(int M, int Nn, int K, const float *__restrict__ A, int lda,
                                              const float *__restrict__ Bm, int ldb, float *__restrict__ C, int ldc,
                                              int kchunk, size_t slab_stride) {
    __shared__ __attribute__((aligned(16))) float As[GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[GBK * GLD];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    C += (size_t)blockIdx.z * slab_stride;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;

    float4 ra[2], rb[2];
    // op(A): TA=false -> A is M x K, m contiguous (direct); TA=true -> A stored K x M, k contiguous
    // op(B): TB=true  -> B stored Nn x K, n contiguous (direct); TB=false -> B is K x Nn, k contiguous
    gemm_load<TA>(A, lda, m0, M, kbeg, kend, tid, ra);
    gemm_load<!TB>(Bm, ldb, n0, Nn, kbeg, kend, tid, rb);
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        gemm_store<TA>(As, tid, ra);
        gemm_store<!TB>(Bs, tid, rb);
        __syncthreads();
        if (k0 + GBK < kend) {
            gemm_load<TA>(A, lda, m0, M, k0 + GBK, kend, tid, ra);
            gemm_load<!TB>(Bm, ldb, n0, Nn, k0 + GBK, kend, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            const int k = kk + (l >> 5);
            float af[2], bf[2];
            af[0] = As[k * GLD + wm * 64 + (l & 31)];
            af[1] = As[k * GLD + wm * 64 + 32 + (l & 31)];
            bf[0] = Bs[k * GLD + wn * 64 + (l & 31)];
            bf[1] = Bs[k * GLD + wn * 64 + 32 + (l & 31)];
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[row][col] of the swapped product = C[m = col][n = row]
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int m = m0 + wm * 64 + mi * 32 + (l & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + wn * 64 + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (m < M && n < Nn) C[(size_t)n * ldc + m] = acc[mi][ni][r];
            }
        }
}

__global__ __launch_bounds__(256) void k_gemm_reduce(const float *__restrict__ slabs, int splits, int M, int Nn,
                                                     float *__restrict__ C, int ldc) {
    const size_t total = (size_t)M * Nn;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = slabs[e];
        for (int z = 1; z < splits; z++) s += slabs[(size_t)z * total + e];
        const size_t m = e % M, n = e / M;
        C[n * ldc + m] = s;
    }
}

int gemm_pick_splits(int M, int Nn, int K) {
    const int tiles = ((M + GBM - 1) / GBM) * ((Nn + GBN - 1) / GBN);
    int splits = 1;
    // aim for >= ~512 workgroups, keep >= 8 k-tiles per split
    while (tiles * splits < 512 && K / (splits * 2) >= 8 * GBK) splits *= 2;
    return splits;
}

void gemm(bool TA, bool TB, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
          int splits, float *slabs, hipStream_t st) {
    if (splits < 1) splits = 1;
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + GBK - 1) / GBK) * GBK;
    splits = (K + kchunk - 1) / kchunk;
    dim3 grid((M + GBM - 1) / GBM, (Nn + GBN - 1) / GBN, splits);
    float *out = splits > 1 ? slabs : C;
    const int ldo = splits > 1 ? M : ldc;
    const size_t stride = splits > 1 ? (size_t)M * Nn : 0;
#define GEMM_LAUNCH(ta, tb) \
    hipLaunchKernelGGL((k_gemm<ta, tb>), grid, dim3(256), 0, st, M, Nn, K, A, lda, B, ldb, out, ldo, kchunk, stride)
    if (!TA && !TB) GEMM_LAUNCH(false, false);
    else if (TA && !TB) GEMM_LAUNCH(true, false);
    else if (!TA && TB) GEMM_LAUNCH(false, true);
    else GEMM_LAUNCH(true, true);
#undef GEMM_LAUNCH
    if (splits > 1) {
        size_t total = (size_t)M * Nn;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_gemm_reduce, dim3(blocks), dim3(256), 0, st, slabs, splits, M, Nn, C, ldc);
    }
}

// ------------------------------------------------------------------------------------------------
// softmax_loss_dy: one wave per output column (M = 256 = 64 lanes x float4).
//   probs = exp(y + by) / sum  (no max shift)     R/lstm.cc:195-201
//   surprisal = -log2(probs[target])              R/lstm.cc:204
//   dy = probs - target                           R/lstm.cc:225
// ------------------------------------------------------------------------------------------------
constexpr int SM_COLS_PER_WAVE = 8;
__global__ __launch_bounds__(256) void k_softmax_loss_dy(float *__restrict__ Y, float *__restrict__ P,
                                                         const float *__restrict__ by, const int32_t *__restrict__ ti,
                                                         float *__restrict__ colloss, float *__restrict__ dby_part,
                                                         int T) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + w;
    const float4 b4 = reinterpret_cast<const float4 *>(by)[l];
    float4 dsum = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < SM_COLS_PER_WAVE; q++) {
        const int col = gw * SM_COLS_PER_WAVE + q;
        if (col >= T) break;
        float4 *yp = reinterpret_cast<float4 *>(Y + (size_t)col * 256) + l;
        float4 y = *yp;
        float4 e;
        e.x = expf(y.x + b4.x);
        e.y = expf(y.y + b4.y);
        e.z = expf(y.z + b4.z);
        e.w = expf(y.w + b4.w);
        const float s = wave_sum((e.x + e.y) + (e.z + e.w));
        float4 p;
        p.x = e.x / s;
        p.y = e.y / s;
        p.z = e.z / s;
        p.w = e.w / s;
        reinterpret_cast<float4 *>(P + (size_t)col * 256)[l] = p;
        const int tk = ti[col];
        float4 d = p;
        if (tk >= 0 && (tk >> 2) == l) {
            const int c = tk & 3;
            const float pt = c == 0 ? p.x : c == 1 ? p.y : c == 2 ? p.z : p.w;
            colloss[col] = -log2f(pt);
            if (c == 0) d.x -= 1.0f;
            else if (c == 1) d.y -= 1.0f;
            else if (c == 2) d.z -= 1.0f;
            else d.w -= 1.0f;
        }
        if (tk < 0 && l == 0) colloss[col] = 0.0f;
        *yp = d;
        dsum.x += d.x;
        dsum.y += d.y;
        dsum.z += d.z;
        dsum.w += d.w;
    }
    reinterpret_cast<float4 *>(dby_part + (size_t)gw * 256)[l] = dsum;
}
void softmax_loss_dy(float *Y, float *P, const float *by, const int32_t *ti, float *colloss, float *dby_part, int T,
                     int *n_parts_out, hipStream_t st) {
    const int waves = (T + SM_COLS_PER_WAVE - 1) / SM_COLS_PER_WAVE;
    const int blocks = (waves + 3) / 4;
    *n_parts_out = blocks * 4;
    hipLaunchKernelGGL(k_softmax_loss_dy, dim3(blocks), dim3(256), 0, st, Y, P, by, ti, colloss, dby_part, T);
}

// dby = rowsum(dY) (R/lstm.cc:227): fold the per-wave partials.  1024 threads = 64 float4 row groups
// x 16 phases; phase q sums partials q, q+16, ... in order, then the 16 phase sums are added in order.
__global__ __launch_bounds__(1024) void k_dby_finish(const float *__restrict__ part, int n_parts, float *__restrict__ dby) {
    __shared__ float4 red[16][64];
    const int m4 = threadIdx.x & 63, q = threadIdx.x >> 6;
    float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int p = q; p < n_parts; p += 16) {
        const float4 v = reinterpret_cast<const float4 *>(part + (size_t)p * 256)[m4];
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    red[q][m4] = s;
    __syncthreads();
    if (q == 0) {
        float4 t = red[0][m4];
        for (int i = 1; i < 16; i++) {
            t.x += red[i][m4].x;
            t.y += red[i][m4].y;
            t.z += red[i][m4].z;
            t.w += red[i][m4].w;
        }
        reinterpret_cast<float4 *>(dby)[m4] = t;
    }
}
void dby_finish(const float *dby_part, int n_parts, float *dby, hipStream_t st) {
    hipLaunchKernelGGL(k_dby_finish, dim3(1), dim3(1024), 0, st, dby_part, n_parts, dby);
}

// loss += surprisals.sum() / B per step (OV/lstm_eigen_opt/lstm.cc:249): float sum over the columns
// of a step, divided by the (global) batch, accumulated over steps in double.
__global__ __launch_bounds__(256) void k_loss_reduce(const float *__restrict__ colloss, int steps, int B, int Bg,
                                                     double *__restrict__ out) {
    __shared__ double part[256];
    double acc = 0.0;
    for (int t = threadIdx.x; t < steps; t += 256) {
        float s = 0.0f;
        for (int b = 0; b < B; b++) s += colloss[(size_t)t * B + b];
        acc += (double)(s / (float)Bg);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < 256; i++) tot += part[i];
        out[0] = tot;
    }
}
void loss_reduce(const float *colloss, int steps, int B, int B_global, double *out, hipStream_t st) {
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, st, colloss, steps, B, B_global, out);
}

// ------------------------------------------------------------------------------------------------
// dW_db: dW += dg * x^T with one-hot x (R/lstm.cc:251) = per-input-byte sums of DG columns;
// db += dg (R/lstm.cc:252) = sum over all buckets.  128 threads = 16 rows x 8 column phases.
// LDS table acc[257][128]: word (v, tid) is only ever touched by thread tid -> deterministic.
// ------------------------------------------------------------------------------------------------
constexpr int DW_ROWS = 16, DW_COPIES = 8, DW_THREADS = DW_ROWS * DW_COPIES;
__global__ __launch_bounds__(DW_THREADS) void k_dW_db(const float *__restrict__ DG, const int32_t *__restrict__ xi, int T,
                                                      int G4, float *__restrict__ dW, float *__restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float acc[]; // 257 * 128 floats
    const int tid = threadIdx.x, r = tid & (DW_ROWS - 1), q = tid / DW_ROWS;
    const int r0 = blockIdx.x * DW_ROWS;
    for (int i = tid; i < 257 * DW_THREADS; i += DW_THREADS) acc[i] = 0.0f;
    __syncthreads();
    const float *src = DG + r0 + r;
    constexpr int UN = 8; // columns in flight per thread
    int col = q;
    for (; col + (UN - 1) * DW_COPIES < T; col += UN * DW_COPIES) {
        int v[UN];
        float val[UN];
#pragma unroll
        for (int i = 0; i < UN; i++) {
            v[i] = xi[col + i * DW_COPIES];
            val[i] = src[(size_t)(col + i * DW_COPIES) * G4];
        }
#pragma unroll
        for (int i = 0; i < UN; i++) {
            const int vv = v[i] < 0 ? 256 : v[i];
            atomicAdd(&acc[vv * DW_THREADS + tid], val[i]); // ds_add_f32 on a word private to this thread
        }
    }
    for (; col < T; col += DW_COPIES) {
        int v = xi[col];
        v = v < 0 ? 256 : v;
        atomicAdd(&acc[v * DW_THREADS + tid], src[(size_t)col * G4]);
    }
    __syncthreads();
    // fold the 8 phases in fixed order; keep bucket totals in slot 0 for the db pass
    for (int i = tid; i < 257 * DW_ROWS; i += DW_THREADS) {
        const int v = i / DW_ROWS, rr = i % DW_ROWS;
        float s = acc[v * DW_THREADS + rr];
#pragma unroll
        for (int c = 1; c < DW_COPIES; c++) s += acc[v * DW_THREADS + c * DW_ROWS + rr];
        if (v < 256) dW[(size_t)v * G4 + r0 + rr] = s;
        acc[v * DW_THREADS + rr] = s;
    }
    __syncthreads();
    if (tid < DW_ROWS) {
        float s = 0.0f;
        for (int v = 0; v < 257; v++) s += acc[v * DW_THREADS + tid];
        db[r0 + tid] = s;
    }
}
void dW_db(const float *DG, const int32_t *xi, int T, int G4, float *dW, float *db, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = 257 * DW_THREADS * sizeof(float);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_dW_db), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_dW_db, dim3(G4 / DW_ROWS), dim3(DW_THREADS), lds, st, DG, xi, T, G4, dW, db);
}

// ------------------------------------------------------------------------------------------------
// adagrad: m += d.*d ; p -= lr * d ./ sqrt(m + eps)    R/lstm.cc:261-272.  eps = 1e-10 is a double
// literal there (R/lstm.cc:25,46-48): the add is done in double and narrowed before sqrtf.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float adagrad1(float p, float d, float &m, float lr) {
    m = m + d * d;
    const float den = sqrtf((float)((double)m + 1e-10));
    return p - lr * (d / den);
}
__global__ __launch_bounds__(256) void k_adagrad(float *__restrict__ P, const float *__restrict__ dP,
                                                 float *__restrict__ mem, size_t n4, float lr) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4 *>(P)[i];
        const float4 d = reinterpret_cast<const float4 *>(dP)[i];
        float4 m = reinterpret_cast<float4 *>(mem)[i];
        p.x = adagrad1(p.x, d.x, m.x, lr);
        p.y = adagrad1(p.y, d.y, m.y, lr);
        p.z = adagrad1(p.z, d.z, m.z, lr);
        p.w = adagrad1(p.w, d.w, m.w, lr);
        reinterpret_cast<float4 *>(P)[i] = p;
        reinterpret_cast<float4 *>(mem)[i] = m;
    }
}
void adagrad(float *P, const float *dP, float *mem, size_t n, float lr, hipStream_t st) {
    const size_t n4 = n / 4; // the flat block is a multiple of 4 floats (M = 256, N % 16 == 0)
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adagrad, dim3(blocks), dim3(256), 0, st, P, dP, mem, n4, lr);
}

// ------------------------------------------------------------------------------------------------
// slide_window: OV/lstm_eigen_opt/lstm.cc:190-213 on indices.  x and target are kept as rings of S
// rows (row s of the window lives in ring row (head+s)%S), so "shift every column left by one" is
// head++ and the new column overwrites the slot of the column that fell off -- exactly the
// reference's result, including row 0.  One workgroup:
//   event = text[pos]; pos++ (wrap to S)                              opt:192-197
//   target[S-1] = onehot(event); x[S-1] = target[S-2]                 opt:211-212
//   flat xi/ti (what the kernels read) are rewritten from the rings
//   h[0] <- h[1], c[0] <- c[1]                                        opt:205-206
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_slide_window(const uint8_t *__restrict__ text, uint64_t len,
                                                       uint64_t *__restrict__ pos, int32_t *__restrict__ Xr,
                                                       int32_t *__restrict__ Tr, int32_t *__restrict__ headp,
                                                       int32_t *__restrict__ xi, int32_t *__restrict__ ti,
                                                       float *__restrict__ H, float *__restrict__ C, int S, int B,
                                                       int NB4) {
    const int head = (*headp + 1) % S;
    const int last = (head + S - 1) % S, prev = (head + S - 2) % S;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        uint64_t p = pos[b];
        const int event = text[p];
        p++;
        if (p >= len) p = (uint64_t)S;
        pos[b] = p;
        Tr[last * B + b] = event;
        Xr[last * B + b] = Tr[prev * B + b];
    }
    // carry: column 0 of the next window is column 1 of this one
    for (int i = threadIdx.x; i < NB4; i += blockDim.x) {
        reinterpret_cast<float4 *>(H)[i] = reinterpret_cast<const float4 *>(H)[NB4 + i];
        reinterpret_cast<float4 *>(C)[i] = reinterpret_cast<const float4 *>(C)[NB4 + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S * B; i += blockDim.x) {
        const int t = i / B, b = i - t * B;
        const int row = (head + t) % S;
        xi[i] = Xr[row * B + b];
        ti[i] = Tr[row * B + b];
    }
    if (threadIdx.x == 0) *headp = head;
}
void slide_window(const uint8_t *text, uint64_t len, uint64_t *pos, int32_t *Xr, int32_t *Tr, int32_t *headp,
                  int32_t *xi, int32_t *ti, float *H, float *C, int S, int B, int N, hipStream_t st) {
    hipLaunchKernelGGL(k_slide_window, dim3(1), dim3(1024), 0, st, text, len, pos, Xr, Tr, headp, xi, ti, H, C, S, B,
                       N * B / 4);
}

// ------------------------------------------------------------------------------------------------
// B = 1 recurrence (evaluator, sampler): one 1024-thread workgroup; h, c, g live in LDS.
//   test():   OV/lstm_eigen_class_CUDA/lstm.cc:661-720      sample(): R/lstm.cc:293-356
// ------------------------------------------------------------------------------------------------
__device__ void b1_step(const float *__restrict__ W, const float *__restrict__ U, const float *__restrict__ bias, int N,
                        int x, float *hs, float *cs, float *gs) {
    const int G4 = 4 * N;
    for (int r = threadIdx.x; r < G4; r += blockDim.x) {
        float uh = 0.0f;
        for (int k = 0; k < N; k++) uh += U[(size_t)k * G4 + r] * hs[k];
        const float pre = (W[(size_t)x * G4 + r] + uh) + bias[r];
        gs[r] = r < 3 * N ? sigm<false>(pre) : tanh_<false>(pre);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const float c = tanh_<false>(gs[j] * gs[3 * N + j] + gs[2 * N + j] * cs[j]);
        cs[j] = c;
        hs[j] = gs[N + j] * c;
    }
    __syncthreads();
}
// probs (unnormalised exp) into ps[256]; returns the sum (computed by every thread identically)
__device__ float b1_output(const float *__restrict__ Why, const float *__restrict__ by, int N, const float *hs,
                           float *ps) {
    for (int m = threadIdx.x; m < 256; m += blockDim.x) {
        float y = 0.0f;
        for (int k = 0; k < N; k++) y += Why[(size_t)k * 256 + m] * hs[k];
        ps[m] = expf(y + by[m]);
    }
    __syncthreads();
    float s = 0.0f;
    for (int m = 0; m < 256; m++) s += ps[m];
    return s;
}
__global__ __launch_bounds__(1024) void k_eval_bits(const float *__restrict__ P, int N, const uint8_t *__restrict__ text,
                                                    uint64_t len, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *hs = sm, *cs = sm + N, *gs = sm + 2 * N, *ps = sm + 6 * N;
    const ParamLayout pl = ParamLayout::make(N, 256);
    for (int j = threadIdx.x; j < N; j += blockDim.x) hs[j] = cs[j] = 0.0f;
    __syncthreads();
    double err = 0.0;
    for (uint64_t ii = 0; ii + 1 < len; ii++) {
        b1_step(P + pl.W, P + pl.U, P + pl.b, N, text[ii], hs, cs, gs);
        const float s = b1_output(P + pl.Why, P + pl.by, N, hs, ps);
        err += -(double)log2f(ps[text[ii + 1]] / s);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = err;
}
void eval_bits(const float *P, int N, const uint8_t *text, uint64_t len, double *out_bits_sum, float *, hipStream_t st) {
    const size_t lds = (size_t)(6 * N + 256) * sizeof(float);
    hipLaunchKernelGGL(k_eval_bits, dim3(1), dim3(1024), lds, st, P, N, text, len, out_bits_sum);
}
__global__ __launch_bounds__(1024) void k_sample(const float *__restrict__ P, int N, float *__restrict__ hc,
                                                 const double *__restrict__ u, int count, uint8_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *hs = sm, *cs = sm + N, *gs = sm + 2 * N, *ps = sm + 6 * N;
    __shared__ int s_index;
    const ParamLayout pl = ParamLayout::make(N, 256);
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        hs[j] = hc[j];
        cs[j] = hc[N + j];
    }
    __syncthreads();
    for (int i = 0; i < count; i++) {
        const float s = b1_output(P + pl.Why, P + pl.by, N, hs, ps);
        if (threadIdx.x == 0) {
            // cumulative sum, first index with r < cdf (R/lstm.cc:321-338); index 0 if none
            const float r = (float)u[i];
            float cdf = 0.0f;
            int index = 0;
            for (int m = 0; m < 256; m++) {
                cdf += ps[m] / s;
                if (r < cdf) {
                    index = m;
                    break;
                }
            }
            s_index = index;
            out[i] = (uint8_t)index;
        }
        __syncthreads();
        b1_step(P + pl.W, P + pl.U, P + pl.b, N, s_index, hs, cs, gs);
    }
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        hc[j] = hs[j];
        hc[N + j] = cs[j];
    }
}
void sample(const float *P, int N, float *hc, const double *u, int count, uint8_t *out, float *, hipStream_t st) {
    const size_t lds = (size_t)(6 * N + 256) * sizeof(float);
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(1024), lds, st, P, N, hc, u, count, out);
}

} // namespace lstmk
