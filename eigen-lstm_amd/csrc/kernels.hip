// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the LSTM training window.
//
// Reference semantics restated by each kernel are cited as R/ (= /root/reference) file:line.
// All matrices are column-major fp32.  Gate row order is [i; o; f; u] (R/lstm.cc:77).
//
// MFMA fragment maps used below (cdna_hip_programming.md section 3):
//   v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)*4+reg][col=l&15]
//   v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(reg&3)+8*(reg>>2)+4*(l>>5)][col=l&31]
#include "kernels.h"

#include <cstdlib>

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
//   Ubwd4[kb][w][m][l].z' (optional; the 4x4x1 form of the backward recurrence, k_bwd_persistent<.., M4>): wave w of
//        workgroup kb owns gate rows [Kw*w, Kw*(w+1)), Kw = N/2; lane l = 32x' + 16y + 4z + j;
//        = U[Kw*w + 32*(m>>1) + 4*(2z' + y) + 2*(m&1) + x'][16*kb + 4z + j]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t ubwd4_index(int gk, int hr, int N) { // float index of U[gk][hr] in Ubwd4
    const int Kw = N / 2, w = gk / Kw, kk = gk % Kw, rem = kk & 31;
    const int m = 2 * (kk >> 5) + ((rem >> 1) & 1), xp = rem & 1, y = (rem >> 2) & 1, zp = rem >> 3;
    const int l = 32 * xp + 16 * y + (hr & 15);
    return ((((size_t)(hr >> 4) * 8 + w) * (N / 32) + m) * 64 + l) * 4 + zp;
}
//   Ufwd4[kb][w][pass][L][eh][sh][l].c (optional; the 8-column one-recurrence form of the forward recurrence, k_fwd_persistent4): wave w of
//        workgroup kb owns input indices [Kw*w, Kw*(w+1)), Kw = N/8; lane l = 32x' + 4u + j; slot s = 4*sh + c;
//        = U[gate j of unit 16*kb + 8*pass + u][Kw*w + 32*L + 4*s + 2*eh + x']
__device__ __forceinline__ size_t ufwd4_index(int row, int k, int N) { // float index of U[row][k] in Ufwd4
    const int gate = row / N, u = row % N, Kw = N / 8, w = k / Kw, kk = k % Kw, rem = kk & 31;
    const int L = kk >> 5, s = rem >> 2, eh = (rem >> 1) & 1, xp = rem & 1;
    const int l = 32 * xp + 4 * (u & 7) + gate;
    return (((((((size_t)(u >> 4) * 8 + w) * 2 + ((u >> 3) & 1)) * (Kw / 32) + L) * 2 + eh) * 2 + (s >> 2)) * 64 + l) * 4 + (s & 3);
}
//   Ufwd5[kb][w][ab][l].r (the 4-column-half form of the forward recurrence, k_fwd_persistent6; stored through the Ufwd4
//        pointer when `half_forms` is set): wave w of workgroup kb owns input indices [Kw*w, Kw*(w+1)), Kw = N/8; lane
//        l = 4*unit + gate;  = U[gate of unit 16*kb + unit][Kw*w + 4*ab + r]
__device__ __forceinline__ size_t ufwd5_index(int row, int k, int N) { // float index of U[row][k] in Ufwd5
    const int gate = row / N, u = row % N, Kw = N / 8, w = k / Kw, kk = k % Kw;
    const int l = 4 * (u & 15) + gate;
    return ((((size_t)(u >> 4) * 8 + w) * (Kw / 4) + (kk >> 2)) * 64 + l) * 4 + (kk & 3);
}
//   Ubwd6[kb][w][ab][l].r (the scatter form of the backward recurrence, k_bwd_scatter; stored through the Ubwd4 pointer when
//        bit 2 of `half_forms` is set): workgroup kb keeps the 64 gate rows of ITS units, k = gate*16 + (unit - 16*kb) = 4*ab + r,
//        for all N outputs: wave w owns outputs [64w, 64w+64), lane l = 4*block + j is output 64w + l;
//        = U[gate*N + 16*kb + unit][64*w + l]
__device__ __forceinline__ size_t ubwd6_index(int gk, int hr, int N) { // float index of U[gk][hr] in Ubwd6
    const int gate = gk / N, unit = gk % N, kb = unit >> 4, kk = gate * 16 + (unit & 15);
    const int w = hr >> 6, l = hr & 63;
    return ((((size_t)kb * (N / 64) + w) * 16 + (kk >> 2)) * 64 + l) * 4 + (kk & 3);
}
// half_forms: bit 0 = the forward image is Ufwd5, bit 2 = the backward image is Ubwd6 (bit 1: a removed form)
__device__ __forceinline__ size_t ufwd45_index(int row, int k, int N, int half_forms) {
    return (half_forms & 1) ? ufwd5_index(row, k, N) : ufwd4_index(row, k, N);
}
__device__ __forceinline__ size_t ubwd45_index(int gk, int hr, int N, int half_forms) {
    return (half_forms & 4) ? ubwd6_index(gk, hr, N) : ubwd4_index(gk, hr, N);
}
__global__ __launch_bounds__(256) void k_pack_U(const float *__restrict__ U, float4 *__restrict__ Ufwd,
                                                float4 *__restrict__ Ubwd, float4 *__restrict__ Ubwd4,
                                                float4 *__restrict__ Ufwd4, int N, int half_forms) {
    const int G4 = 4 * N;
    const size_t nf4 = (size_t)N * N; // float4 count of each image (4N*N floats)
    const size_t total = ((Ubwd4 || Ufwd4) ? 3 : 2) * nf4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        if (e >= 2 * nf4) { // one float4 of U (4 gate rows of one hidden column) -> four scalars of Ubwd4
            const size_t e3 = e - 2 * nf4;
            const int r = 4 * (int)(e3 % N), k = (int)(e3 / N);
            const float4 p = *reinterpret_cast<const float4 *>(U + (size_t)k * G4 + r);
            float *u4 = reinterpret_cast<float *>(Ubwd4);
            if (Ubwd4 != nullptr) {
                if (half_forms & 4) { // Ubwd6: four consecutive gate rows of one hidden column are one 16-byte piece of the image
                    *reinterpret_cast<float4 *>(u4 + ubwd6_index(r, k, N)) = p;
                } else {
                    u4[ubwd45_index(r + 0, k, N, half_forms)] = p.x;
                    u4[ubwd45_index(r + 1, k, N, half_forms)] = p.y;
                    u4[ubwd45_index(r + 2, k, N, half_forms)] = p.z;
                    u4[ubwd45_index(r + 3, k, N, half_forms)] = p.w;
                }
            }
            if (Ufwd4 != nullptr) {
                float *f4 = reinterpret_cast<float *>(Ufwd4);
                f4[ufwd45_index(r + 0, k, N, half_forms)] = p.x;
                f4[ufwd45_index(r + 1, k, N, half_forms)] = p.y;
                f4[ufwd45_index(r + 2, k, N, half_forms)] = p.z;
                f4[ufwd45_index(r + 3, k, N, half_forms)] = p.w;
            }
        } else if (e < nf4) {
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
            if (Ufwd != nullptr) Ufwd[e] = v;
        } else {
            size_t e2 = e - nf4;
            int l = (int)(e2 & 63);
            size_t q = e2 >> 6;
            int r4 = (int)(q % (N / 4)), kb = (int)(q / (N / 4));
            int r = 16 * r4 + 4 * (l >> 4);
            int k = 16 * kb + (l & 15);
            if (Ubwd != nullptr) Ubwd[e2] = *reinterpret_cast<const float4 *>(U + (size_t)k * G4 + r);
        }
    }
}
void pack_U(const float *U, float4 *Ufwd, float4 *Ubwd, int N, hipStream_t st, float4 *Ubwd4, float4 *Ufwd4, int half_forms) {
    size_t n = ((Ubwd4 || Ufwd4) ? 3 : 2) * (size_t)N * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_U, dim3(blocks), dim3(256), 0, st, U, Ufwd, Ubwd, Ubwd4, Ufwd4, N, half_forms);
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
// The fp32 time-batched products themselves live in gemm.hip (register-streamed 32x32x2 tiles); here only the ordered
// fold of split-K slabs / per-group partial blocks, which the bf16 products and the fused backward recurrence also use.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemm_reduce(const float *__restrict__ slabs, int splits, int M, int Nn,
                                                     float *__restrict__ C, int ldc, size_t slab_stride) {
    const size_t total = (size_t)M * Nn;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        float s = slabs[e];
        for (int z = 1; z < splits; z++) s += slabs[(size_t)z * slab_stride + e];
        const size_t m = e % M, n = e / M;
        C[n * ldc + m] = s;
    }
}

static void gemm_reduce_launch(const float *slabs, int splits, int M, int Nn, float *C, int ldc, hipStream_t st,
                               size_t slab_stride = 0) {
    size_t total = (size_t)M * Nn;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_gemm_reduce, dim3(blocks), dim3(256), 0, st, slabs, splits, M, Nn, C, ldc,
                       slab_stride ? slab_stride : total);
}
void gemm_fold(const float *slabs, int splits, int M, int Nn, float *C, int ldc, hipStream_t st, size_t slab_stride) {
    gemm_reduce_launch(slabs, splits, M, Nn, C, ldc, st, slab_stride);
}

// ------------------------------------------------------------------------------------------------
// bf16 time-batched products (LSTM_HIP_BF16_RECURRENCE, BASELINE configs[4]): C[m + ldc*n] = sum_k A[m][k] * B[n][k],
// fp32 accumulate, on v_mfma_f32_32x32x16_bf16.  Both operands are bfloat16 images with k CONTIGUOUS (k_transpose_pack_bf16
// builds them from the fp32 column-major activations), so a lane's MFMA fragment -- 8 consecutive k of one row -- is one
// 16-byte read.  Workgroup = 4 waves, tile 128 x 128 x 64 (each wave 64 x 64 = 2 x 2 MFMA tiles), LDS double-buffered
// (rows padded to 144 bytes: conflict-free 16-byte fragment reads), next k-tile's global loads in flight during the
// MFMAs.  The operands are swapped in the instruction so that the accumulator holds C^T fragments and the stores run
// along m.  K must be a multiple of 64 (the packed images are zero-padded); split-K writes slabs like k_gemm.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int HBK_ = 64, HLD_ = 72; // k-tile and the padded LDS row (elements)
// TM x TN tile: 128 x 128 where that still gives the chip enough workgroups, 64 x 64 (one 32 x 32 instruction tile per wave)
// for the small products of a narrow batch -- configs[4]'s Y = Why * H is 256 x 1 584: 26 tiles of 128 x 128, 100 of 64 x 64
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm_bf16(int M, int Nn, int K, const unsigned short *__restrict__ A, int lda,
                                                   const unsigned short *__restrict__ Bm, int ldb, float *__restrict__ C,
                                                   int ldc, int kchunk, size_t slab_stride) {
    constexpr int MI = TM / 64, NI = TN / 64, PA = TM / 32, PB = TN / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned short hl[]; // As[2][TM*72] | Bs[2][TN*72]
    unsigned short *As = hl, *Bs = hl + 2 * TM * HLD_;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int kbeg = blockIdx.z * kchunk, kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    C += (size_t)blockIdx.z * slab_stride;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int a = 0; a < MI; a++)
#pragma unroll
        for (int b = 0; b < NI; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
    const int srow = tid >> 3, sch = (tid & 7) * 8; // staging: 32 rows x 8 chunks of 8 elements per pass
    uint4 ra[PA], rb[PB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < PA; p++) {
            const int row = p * 32 + srow;
            ra[p] = (m0 + row < M) ? *reinterpret_cast<const uint4 *>(A + (size_t)(m0 + row) * lda + k0 + sch) : uint4{0, 0, 0, 0};
        }
#pragma unroll
        for (int p = 0; p < PB; p++) {
            const int row = p * 32 + srow;
            rb[p] = (n0 + row < Nn) ? *reinterpret_cast<const uint4 *>(Bm + (size_t)(n0 + row) * ldb + k0 + sch) : uint4{0, 0, 0, 0};
        }
    };
    auto lstore = [&](int stage) {
#pragma unroll
        for (int p = 0; p < PA; p++) *reinterpret_cast<uint4 *>(As + (size_t)stage * TM * HLD_ + (p * 32 + srow) * HLD_ + sch) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; p++) *reinterpret_cast<uint4 *>(Bs + (size_t)stage * TN * HLD_ + (p * 32 + srow) * HLD_ + sch) = rb[p];
    };
    const int ntiles = (kend - kbeg) / HBK_;
    if (ntiles <= 0) return;
    gload(kbeg);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < ntiles) gload(kbeg + (kt + 1) * HBK_);
        const unsigned short *Ac = As + (size_t)cur * TM * HLD_ + (wm * (TM / 2) + (l & 31)) * HLD_ + 8 * (l >> 5);
        const unsigned short *Bc = Bs + (size_t)cur * TN * HLD_ + (wn * (TN / 2) + (l & 31)) * HLD_ + 8 * (l >> 5);
#pragma unroll
        for (int ks = 0; ks < HBK_ / 16; ks++) {
            bf16x8_t af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; i++) af[i] = *reinterpret_cast<const bf16x8_t *>(Ac + i * 32 * HLD_ + ks * 16);
#pragma unroll
            for (int i = 0; i < NI; i++) bf[i] = *reinterpret_cast<const bf16x8_t *>(Bc + i * 32 * HLD_ + ks * 16);
#pragma unroll
            for (int mi = 0; mi < MI; mi++)
#pragma unroll
                for (int ni = 0; ni < NI; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < ntiles) lstore(cur ^ 1); // the other stage: its readers finished before the previous barrier
        __syncthreads();
    }
    // D[row][col] of the swapped product = C[m = col][n = row]
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
            const int m = m0 + wm * (TM / 2) + mi * 32 + (l & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int n = n0 + wn * (TN / 2) + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (m < M && n < Nn) C[(size_t)n * ldc + m] = acc[mi][ni][r];
            }
        }
}
// tile of a product: 64 x 64 while 128 x 128 tiles (times the split-K factor) would leave most of the chip without one
static int gemm_bf16_tile(int M, int Nn, int splits) {
    static const int force = getenv("LSTM_HIP_BF16_GEMM_TILE") ? atoi(getenv("LSTM_HIP_BF16_GEMM_TILE")) : 0; // A/B: 64 or 128
    if (force == 64 || force == 128) return force;
    return ((M + 127) / 128) * ((Nn + 127) / 128) * splits < 128 ? 64 : 128;
}
// splits > 1: slabs of M*Nn floats (ld = M) + ordered fold, as gemm().  K: multiple of 64.
void gemm_bf16(int M, int Nn, int K, const unsigned short *A, int lda, const unsigned short *B, int ldb, float *C, int ldc,
               int splits, float *slabs, hipStream_t st) {
    if (splits < 1) splits = 1;
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + HBK_ - 1) / HBK_) * HBK_;
    splits = (K + kchunk - 1) / kchunk;
    float *out = splits > 1 ? slabs : C;
    const int ldo = splits > 1 ? M : ldc;
    const size_t stride = splits > 1 ? (size_t)M * Nn : 0;
    const int T = gemm_bf16_tile(M, Nn, splits);
    const size_t lds = sizeof(unsigned short) * 2 * (T + T) * HLD_;
    if (T == 64) {
        hipLaunchKernelGGL((k_gemm_bf16<64, 64>), dim3((M + 63) / 64, (Nn + 63) / 64, splits), dim3(256), lds, st, M, Nn, K, A, lda, B, ldb,
                           out, ldo, kchunk, stride);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_bf16<128, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_gemm_bf16<128, 128>), dim3((M + 127) / 128, (Nn + 127) / 128, splits), dim3(256), lds, st, M, Nn, K, A, lda, B,
                           ldb, out, ldo, kchunk, stride);
    }
    if (splits > 1) gemm_reduce_launch(slabs, splits, M, Nn, C, ldc, st);
}
int gemm_bf16_pick_splits(int M, int Nn, int K) {
    const int tiles = ((M + 127) / 128) * ((Nn + 127) / 128);
    int splits = 1;
    while (tiles * splits < 256 && K / (splits * 2) >= 4 * HBK_) splits *= 2; // fill the CUs, keep >= 4 k-tiles per split
    return splits;
}
// dst[r][k] = bf16(src[k*ld + r]) for k < K, 0 for K <= k < Kpad: the k-contiguous image of a column-major fp32 matrix whose
// columns are the contraction index (activations [t][b][rows]).  64 x 64 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void k_transpose_pack_bf16(const float *__restrict__ src, int K, int R, int ld,
                                                             unsigned short *__restrict__ dst, int Kpad) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int k = k0 + ty * 16 + i, r = r0 + tx;
        tile[ty * 16 + i][tx] = (k < K && r < R) ? src[(size_t)k * ld + r] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = r0 + ty * 16 + i, k = k0 + tx;
        if (r < R && k < Kpad) dst[(size_t)r * Kpad + k] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][ty * 16 + i]);
    }
}
void transpose_pack_bf16(const float *src, int K, int R, int ld, unsigned short *dst, int Kpad, hipStream_t st) {
    hipLaunchKernelGGL(k_transpose_pack_bf16, dim3((R + 63) / 64, (Kpad + 63) / 64), dim3(256), 0, st, src, K, R, ld, dst, Kpad);
}
// dst[i] = bf16(src[i]) (same layout)
__global__ __launch_bounds__(256) void k_pack_bf16(const float *__restrict__ src, size_t n, unsigned short *__restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = __builtin_bit_cast(unsigned short, (__bf16)src[i]);
}
void pack_bf16(const float *src, size_t n, unsigned short *dst, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_bf16, dim3(blocks), dim3(256), 0, st, src, n, dst);
}

// ------------------------------------------------------------------------------------------------
// softmax_loss_dy: one wave per output column (M = 256 = 64 lanes x float4), two columns per wave, four waves per workgroup;
// the workgroup leaves ONE row of dby partial sums (its eight columns in order).  (Eight columns per wave: 13 us -- too few
// waves in flight; two per wave with a partial row per wave: 8.8 us but the fold over 4x the rows cost 8 us more.)
//   probs = exp(y + by) / sum  (no max shift)     R/lstm.cc:195-201
//   surprisal = -log2(probs[target])              R/lstm.cc:204
//   dy = probs - target                           R/lstm.cc:225
// ------------------------------------------------------------------------------------------------
constexpr int SM_COLS_PER_WAVE = 2;
__global__ __launch_bounds__(256) void k_softmax_loss_dy(float *__restrict__ Y, float *__restrict__ P,
                                                         const float *__restrict__ by, const int32_t *__restrict__ ti,
                                                         float *__restrict__ colloss, float *__restrict__ dby_part,
                                                         int col0, int T) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ float4 part[4][64];
    const int gw = col0 / SM_COLS_PER_WAVE + blockIdx.x * 4 + w; // global wave index: columns 2*gw, 2*gw+1
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
    part[w][l] = dsum;
    __syncthreads();
    if (w == 0) {
        float4 s = part[0][l];
#pragma unroll
        for (int i = 1; i < 4; i++) {
            s.x += part[i][l].x;
            s.y += part[i][l].y;
            s.z += part[i][l].z;
            s.w += part[i][l].w;
        }
        reinterpret_cast<float4 *>(dby_part + (size_t)(col0 / (4 * SM_COLS_PER_WAVE) + blockIdx.x) * 256)[l] = s;
    }
}
// columns [col0, col1) of a T-column problem; col0 must be a multiple of 8.  dby_part needs
// softmax_parts(T) rows of 256 floats; rows of waves past col1 are written as zeros.
int softmax_parts(int T) { return (T + 4 * SM_COLS_PER_WAVE - 1) / (4 * SM_COLS_PER_WAVE) + 4; } // one row per workgroup
void softmax_loss_dy(float *Y, float *P, const float *by, const int32_t *ti, float *colloss, float *dby_part, int col0,
                     int col1, hipStream_t st) {
    const int waves = (col1 - col0 + SM_COLS_PER_WAVE - 1) / SM_COLS_PER_WAVE;
    const int blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(k_softmax_loss_dy, dim3(blocks), dim3(256), 0, st, Y, P, by, ti, colloss, dby_part, col0, col1);
}

// dby = rowsum(dY) (R/lstm.cc:227): fold the per-wave partials.  1024 threads = 64 float4 row groups
// x 16 phases; phase q sums partials q, q+16, ... in order, then the 16 phase sums are added in order.
// loss += surprisals.sum() / B per step (OV/lstm_eigen_opt/lstm.cc:249): float sum over the columns
// of a step, divided by the (global) batch, accumulated over steps in double.
// block 0: window loss; blocks 1..4 (when dby != null): dby = rowsum(dY) from the per-wave partials, 64 rows each
__global__ __launch_bounds__(1024) void k_loss_dby(const float *__restrict__ colloss, int steps, int B, int Bg,
                                                   double *__restrict__ out, const float *__restrict__ dby_part,
                                                   int n_parts, float *__restrict__ dby, float scale) {
    __shared__ float4 red[16][64];
    if (blockIdx.x >= 1) { // 16 float4 row groups x 64 phases; phases folded in order
        const int m4 = (blockIdx.x - 1) * 16 + (threadIdx.x & 15), ph = threadIdx.x >> 4;
        float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int p = ph; p < n_parts; p += 64) {
            const float4 v = reinterpret_cast<const float4 *>(dby_part + (size_t)p * 256)[m4];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        float4 *r = &red[0][0];
        r[ph * 16 + (threadIdx.x & 15)] = s;
        __syncthreads();
        if (ph == 0) {
            float4 t = r[threadIdx.x & 15];
            for (int i = 1; i < 64; i++) {
                const float4 v = r[i * 16 + (threadIdx.x & 15)];
                t.x += v.x;
                t.y += v.y;
                t.z += v.z;
                t.w += v.w;
            }
            reinterpret_cast<float4 *>(dby)[m4] = t;
        }
        return;
    }
    // loss: one thread per timestep walks its B columns in order (the reference's float sum over a step's
    // columns, OV/lstm_eigen_opt/lstm.cc:249) with 8 loads in flight; steps are then added in order in double
    double *part = reinterpret_cast<double *>(&red[0][0]);
    double acc = 0.0;
    for (int t = threadIdx.x; t < steps; t += blockDim.x) {
        const float *cl = colloss + (size_t)t * B;
        float s = 0.0f;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            const float v0 = cl[b], v1 = cl[b + 1], v2 = cl[b + 2], v3 = cl[b + 3], v4 = cl[b + 4], v5 = cl[b + 5],
                        v6 = cl[b + 6], v7 = cl[b + 7];
            s = (((((((s + v0) + v1) + v2) + v3) + v4) + v5) + v6) + v7;
        }
        for (; b < B; b++) s += cl[b];
        acc += (double)((s * scale) / (float)Bg); // scale: 1 (bits) or ln 2 (nats, last-step mode)
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        const int n = steps < (int)blockDim.x ? steps : (int)blockDim.x;
        for (int i = 0; i < n; i++) tot += part[i];
        out[0] = tot;
    }
}
void loss_reduce(const float *colloss, int steps, int B, int B_global, double *out, const float *dby_part, int n_parts,
                 float *dby, hipStream_t st, float scale) {
    hipLaunchKernelGGL(k_loss_dby, dim3(dby ? 5 : 1), dim3(1024), 0, st, colloss, steps, B, B_global, out, dby_part,
                       n_parts, dby, scale);
}

// ------------------------------------------------------------------------------------------------
// dW, db: dW += dg * x^T with one-hot x (R/lstm.cc:251) = per-input-byte sums of DG columns;
// db += dg (R/lstm.cc:252) = sum over all buckets.  Three deterministic passes:
//   k_bucket_columns : stable counting sort of the T column ids by input byte (bucket 256 = empty
//                      column), cut into chunks of <= DW_CHUNK columns                 (one workgroup)
//   k_dW_segsum      : one workgroup per chunk sums its columns in order; each column is a contiguous
//                      4N-float run, read with float4 loads                           (HBM/L2 bound)
//   k_dW_finish      : per (bucket, 1024 rows) adds the bucket's chunk partials in order -> dW;
//                      k_db_finish adds the 257 bucket totals per row -> db
// ------------------------------------------------------------------------------------------------
constexpr int SORT_NCH = 64; // column ranges processed sequentially by one thread each
__global__ __launch_bounds__(1024) void k_bucket_columns(const int32_t *__restrict__ xi, int T,
                                                         int32_t *__restrict__ perm, int32_t *__restrict__ chunk_start,
                                                         int32_t *__restrict__ bucket_chunk, int32_t *__restrict__ n_chunks) {
    __shared__ int hist[SORT_NCH][257];
    __shared__ int bstart[258];
    const int tid = threadIdx.x;
    for (int i = tid; i < SORT_NCH * 257; i += blockDim.x) (&hist[0][0])[i] = 0;
    __syncthreads();
    const int per = (T + SORT_NCH - 1) / SORT_NCH;
    const int c0 = tid * per, c1 = (c0 + per < T) ? c0 + per : T;
    if (tid < SORT_NCH)
        for (int c = c0; c < c1; c++) {
            int v = xi[c];
            v = v < 0 ? 256 : v;
            hist[tid][v]++;
        }
    __syncthreads();
    if (tid < 257) { // exclusive scan over the ranges, per bucket
        int run = 0;
        for (int ch = 0; ch < SORT_NCH; ch++) {
            const int n = hist[ch][tid];
            hist[ch][tid] = run;
            run += n;
        }
        bstart[tid + 1] = run; // bucket size for now
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0, chunks = 0;
        bstart[0] = 0;
        for (int v = 0; v < 257; v++) {
            const int n = bstart[v + 1];
            bucket_chunk[v] = chunks;
            for (int o = 0; o < n; o += DW_CHUNK) chunk_start[chunks++] = run + o;
            run += n;
            bstart[v + 1] = run;
        }
        bucket_chunk[257] = chunks;
        chunk_start[chunks] = T;
        n_chunks[0] = chunks;
    }
    __syncthreads();
    if (tid < SORT_NCH)
        for (int c = c0; c < c1; c++) {
            int v = xi[c];
            v = v < 0 ? 256 : v;
            perm[bstart[v] + hist[tid][v]++] = c;
        }
}

// Fast path for T <= 16384: rank-based stable counting sort in one workgroup.  Columns are taken in
// chunks of 64 (one wave); inside a chunk a lane's rank among equal bytes comes from a 64-step
// readlane sweep, per-chunk counts go to an LDS table, one thread per byte turns them into chunk
// offsets, and every column then knows its slot: bucket start + chunk offset + rank.  Order inside a
// bucket is ascending column, so every float sum downstream has a fixed order.
constexpr int RANK_SLOTS = 16; // chunks per wave: 16 waves x 16 slots x 64 columns = 16384 columns
__global__ __launch_bounds__(1024) void k_bucket_columns_rank(const int32_t *__restrict__ xi, int T,
                                                              int32_t *__restrict__ perm,
                                                              int32_t *__restrict__ chunk_start,
                                                              int32_t *__restrict__ bucket_chunk,
                                                              int32_t *__restrict__ n_chunks) {
    extern __shared__ unsigned short hist[]; // [nch][257]
    __shared__ int bstart[258];
    __shared__ int bchunk[258];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nch = (T + 63) / 64;
    for (int i = tid; i < nch * 257; i += 1024) hist[i] = 0;
    __syncthreads();
    int keys[RANK_SLOTS], ranks[RANK_SLOTS];
#pragma unroll
    for (int s = 0; s < RANK_SLOTS; s++) {
        const int c = w + 16 * s;
        keys[s] = 257;
        ranks[s] = 0;
        if (c < nch) { // wave-uniform
            const int col = c * 64 + lane;
            int key = 257;
            if (col < T) {
                key = xi[col];
                key = key < 0 ? 256 : key;
            }
            int rank = 0, later = 0;
            for (int k = 0; k < 64; k++) {
                const int ok = __shfl(key, k, 64);
                const int same = ok == key;
                rank += same & (k < lane);
                later |= same & (k > lane);
            }
            keys[s] = key;
            ranks[s] = rank;
            if (key < 257 && !later) hist[c * 257 + key] = (unsigned short)(rank + 1);
        }
    }
    __syncthreads();
    if (tid < 257) { // chunk counts -> chunk offsets inside the bucket; bucket size
        int run = 0;
        for (int c = 0; c < nch; c++) {
            const int n = hist[c * 257 + tid];
            hist[c * 257 + tid] = (unsigned short)run;
            run += n;
        }
        bstart[tid + 1] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0, chunks = 0;
        bstart[0] = 0;
        for (int v = 0; v < 257; v++) {
            const int n = bstart[v + 1];
            bchunk[v] = chunks;
            chunks += (n + DW_CHUNK - 1) / DW_CHUNK;
            run += n;
            bstart[v + 1] = run;
        }
        bchunk[257] = chunks;
        chunk_start[chunks] = T;
        n_chunks[0] = chunks;
    }
    __syncthreads();
    if (tid < 258) bucket_chunk[tid] = bchunk[tid];
    if (tid < 257) {
        int q = bchunk[tid];
        for (int o = bstart[tid]; o < bstart[tid + 1]; o += DW_CHUNK) chunk_start[q++] = o;
    }
#pragma unroll
    for (int s = 0; s < RANK_SLOTS; s++) {
        const int c = w + 16 * s;
        if (c < nch && keys[s] < 257) perm[bstart[keys[s]] + hist[c * 257 + keys[s]] + ranks[s]] = c * 64 + lane;
    }
}

__global__ __launch_bounds__(256) void k_dW_segsum(const float *__restrict__ DG, int G4, const int32_t *__restrict__ perm,
                                                   const int32_t *__restrict__ chunk_start,
                                                   const int32_t *__restrict__ bucket_chunk,
                                                   const int32_t *__restrict__ n_chunks, float *__restrict__ part) {
    const int chunk = blockIdx.x;
    if (chunk >= n_chunks[0]) return;
    int c0 = chunk_start[chunk], c1 = chunk_start[chunk + 1];
    // a chunk never crosses a bucket: the next bucket's first chunk starts exactly at this bucket's end
    if (c1 - c0 > DW_CHUNK) c1 = c0 + DW_CHUNK;
    (void)bucket_chunk;
    for (int r4 = threadIdx.x; r4 < G4 / 4; r4 += blockDim.x) {
        float4 s = {0.f, 0.f, 0.f, 0.f};
        int i = c0;
        for (; i + 4 <= c1; i += 4) {
            const float4 v0 = reinterpret_cast<const float4 *>(DG + (size_t)perm[i] * G4)[r4];
            const float4 v1 = reinterpret_cast<const float4 *>(DG + (size_t)perm[i + 1] * G4)[r4];
            const float4 v2 = reinterpret_cast<const float4 *>(DG + (size_t)perm[i + 2] * G4)[r4];
            const float4 v3 = reinterpret_cast<const float4 *>(DG + (size_t)perm[i + 3] * G4)[r4];
            s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x;
            s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
            s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z;
            s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; i < c1; i++) {
            const float4 v = reinterpret_cast<const float4 *>(DG + (size_t)perm[i] * G4)[r4];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        reinterpret_cast<float4 *>(part + (size_t)chunk * G4)[r4] = s;
    }
}

// grid (257, G4/1024): bucket v, rows [1024*by, +1024); bucket 256 (empty columns) only feeds db
__global__ __launch_bounds__(256) void k_dW_finish(const float *__restrict__ part, const int32_t *__restrict__ bucket_chunk,
                                                   int G4, float *__restrict__ dW, float *__restrict__ dWnull) {
    const int v = blockIdx.x;
    const int r4 = blockIdx.y * 256 + threadIdx.x;
    if (r4 >= G4 / 4) return;
    const int q0 = bucket_chunk[v], q1 = bucket_chunk[v + 1];
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int q = q0;
    for (; q + 4 <= q1; q += 4) { // four partial rows in flight, added in order
        const float4 p0 = reinterpret_cast<const float4 *>(part + (size_t)q * G4)[r4];
        const float4 p1 = reinterpret_cast<const float4 *>(part + (size_t)(q + 1) * G4)[r4];
        const float4 p2 = reinterpret_cast<const float4 *>(part + (size_t)(q + 2) * G4)[r4];
        const float4 p3 = reinterpret_cast<const float4 *>(part + (size_t)(q + 3) * G4)[r4];
        s.x = (((s.x + p0.x) + p1.x) + p2.x) + p3.x;
        s.y = (((s.y + p0.y) + p1.y) + p2.y) + p3.y;
        s.z = (((s.z + p0.z) + p1.z) + p2.z) + p3.z;
        s.w = (((s.w + p0.w) + p1.w) + p2.w) + p3.w;
    }
    for (; q < q1; q++) {
        const float4 p = reinterpret_cast<const float4 *>(part + (size_t)q * G4)[r4];
        s.x += p.x;
        s.y += p.y;
        s.z += p.z;
        s.w += p.w;
    }
    float *dst = v < 256 ? dW + (size_t)v * G4 : dWnull;
    reinterpret_cast<float4 *>(dst)[r4] = s;
}
// db[r] = sum over the 257 buckets: 64 rows x 4 bucket phases per workgroup, phases folded in order
__global__ __launch_bounds__(256) void k_db_finish(const float *__restrict__ dW, const float *__restrict__ dWnull, int G4,
                                                   float *__restrict__ db) {
    __shared__ float red[4][64];
    const int rr = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + rr;
    float s = 0.0f;
    if (r < G4) {
#pragma unroll 8
        for (int v = ph * 64; v < ph * 64 + 64; v++) s += dW[(size_t)v * G4 + r];
    }
    red[ph][rr] = s;
    __syncthreads();
    if (ph == 0 && r < G4) db[r] = (((red[0][rr] + red[1][rr]) + red[2][rr]) + red[3][rr]) + dWnull[r];
}

size_t dW_scratch_bytes(int T, int G4) {
    const size_t max_chunks = (size_t)T / DW_CHUNK + 258;
    // perm[T] | chunk_start[max_chunks+1] | bucket_chunk[258] | n_chunks[1] | part[max_chunks][G4] | dWnull[G4]
    return sizeof(int32_t) * ((size_t)T + max_chunks + 1 + 258 + 4) + sizeof(float) * (max_chunks * G4 + G4) + 64;
}
struct DwScratch {
    int32_t *perm, *chunk_start, *bucket_chunk, *n_chunks;
    float *part, *dWnull;
    int max_chunks;
};
static DwScratch dw_carve(void *scratch, int T, int G4) {
    DwScratch d;
    d.max_chunks = T / DW_CHUNK + 258;
    d.perm = reinterpret_cast<int32_t *>(scratch);
    d.chunk_start = d.perm + T;
    d.bucket_chunk = d.chunk_start + d.max_chunks + 1;
    d.n_chunks = d.bucket_chunk + 258;
    size_t off = sizeof(int32_t) * ((size_t)T + d.max_chunks + 1 + 258 + 4);
    off = (off + 63) & ~(size_t)63;
    d.part = reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + off);
    d.dWnull = d.part + (size_t)d.max_chunks * G4;
    return d;
}
// pass 1: depends only on the window's input bytes, so it can run beside the backward recurrence
void dW_sort(const int32_t *xi, int T, int G4, void *scratch, hipStream_t st) {
    const DwScratch d = dw_carve(scratch, T, G4);
    if (T <= 64 * 16 * RANK_SLOTS) {
        const size_t lds = (size_t)((T + 63) / 64) * 257 * sizeof(unsigned short);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bucket_columns_rank),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(k_bucket_columns_rank, dim3(1), dim3(1024), lds, st, xi, T, d.perm, d.chunk_start,
                           d.bucket_chunk, d.n_chunks);
    } else {
        hipLaunchKernelGGL(k_bucket_columns, dim3(1), dim3(1024), 0, st, xi, T, d.perm, d.chunk_start, d.bucket_chunk,
                           d.n_chunks);
    }
}
// passes 2-4: need the complete DG
void dW_sums(const float *DG, int T, int G4, float *dW, float *db, void *scratch, hipStream_t st) {
    const DwScratch d = dw_carve(scratch, T, G4);
    hipLaunchKernelGGL(k_dW_segsum, dim3(d.max_chunks), dim3(256), 0, st, DG, G4, d.perm, d.chunk_start, d.bucket_chunk,
                       d.n_chunks, d.part);
    hipLaunchKernelGGL(k_dW_finish, dim3(257, (G4 / 4 + 255) / 256), dim3(256), 0, st, d.part, d.bucket_chunk, G4, dW,
                       d.dWnull);
    hipLaunchKernelGGL(k_db_finish, dim3((G4 + 63) / 64), dim3(256), 0, st, dW, d.dWnull, G4, db);
}
// Short windows in one pass, no sort: workgroup = 16 gate rows x all T columns, eight waves with a private LDS table
// [257 bytes][16 rows] each.  A wave takes four columns per iteration (lane = 16*column + row: four 64-byte runs of DG) and
// adds them to its table one column after the other -- two of the four may carry the same byte, and LDS operations of a wave
// execute in order, so no atomics and a fixed summation order: by wave, within a wave by column.  At the end the eight
// tables are added in wave order: dW[byte][rows] = 64-byte runs, db = the sum over all 257 buckets (bucket 256 = all-zero
// input column).  Measured against the three passes above: T = 1 584 columns (configs[4]) 27 us against 45; T = 6 336
// 72 against 62; T = 12 672 155 against 128 -- the table's zeroing and eight-way fold are a fixed cost per workgroup, and
// G4 / 16 workgroups are all the parallelism there is (splitting the columns over more workgroups with a fold pass behind
// them: 113 and 330 us at the two long shapes).  Hence: T <= DWT_MAX_T.
constexpr int DWT_ROWS = 16, DWT_WAVES = 8, DWT_MAX_T = 2560;
__global__ __launch_bounds__(64 * DWT_WAVES) void k_dW_table(const float *__restrict__ DG, const int32_t *__restrict__ xi, int T, int G4,
                                                             float *__restrict__ dW, float *__restrict__ db) {
    __shared__ float tab[DWT_WAVES][257][DWT_ROWS];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, cs = l >> 4, r = l & 15;
    const int row = blockIdx.x * DWT_ROWS + r;
    for (int i = tid; i < DWT_WAVES * 257 * DWT_ROWS; i += 64 * DWT_WAVES) (&tab[0][0][0])[i] = 0.0f;
    __syncthreads();
    float(*mine)[DWT_ROWS] = tab[w];
    constexpr int STEP = 4 * DWT_WAVES, UNR = 4;
    for (int c0 = 4 * w; c0 < T; c0 += STEP * UNR) {
        float v[UNR];
        int b[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) { // the loads of UNR iterations in flight together
            const int c = c0 + u * STEP + cs;
            const bool in = c < T;
            const int x = in ? xi[c] : 0;
            b[u] = in ? (x < 0 ? 256 : x) : -1;
            v[u] = in ? DG[(size_t)c * G4 + row] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (cs == q && b[u] >= 0) mine[b[u]][r] += v[u];
                asm volatile("" ::: "memory"); // column q's update is issued before column q+1's
            }
    }
    __syncthreads();
    for (int i = tid; i < 257 * DWT_ROWS; i += 64 * DWT_WAVES) {
        const int byte = i / DWT_ROWS, rr = i % DWT_ROWS;
        float a = tab[0][byte][rr];
#pragma unroll
        for (int ww = 1; ww < DWT_WAVES; ww++) a += tab[ww][byte][rr];
        tab[0][byte][rr] = a;
        if (byte < 256) dW[(size_t)byte * G4 + blockIdx.x * DWT_ROWS + rr] = a;
    }
    __syncthreads();
    if (tid < DWT_ROWS) {
        float a = 0.0f;
        for (int byte = 0; byte < 257; byte++) a += tab[0][byte][tid];
        db[blockIdx.x * DWT_ROWS + tid] = a;
    }
}
void dW_db(const float *DG, const int32_t *xi, int T, int G4, float *dW, float *db, void *scratch, hipStream_t st) {
    static const bool three_pass = getenv("LSTM_HIP_DW_THREE_PASS") && atoi(getenv("LSTM_HIP_DW_THREE_PASS")); // A/B
    if (G4 % DWT_ROWS == 0 && T <= DWT_MAX_T && !three_pass) {
        hipLaunchKernelGGL(k_dW_table, dim3(G4 / DWT_ROWS), dim3(64 * DWT_WAVES), 0, st, DG, xi, T, G4, dW, db);
        return;
    }
    dW_sort(xi, T, G4, scratch, st);
    dW_sums(DG, T, G4, dW, db, scratch, st);
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
// The U block additionally refreshes the two MFMA fragment images (what k_pack_U builds), so the
// forward of the next window needs no separate repack launch.  A float4 here is 4 consecutive gate
// rows of one column k of U: one float4 of Ubwd, four scalars of Ufwd.
// FOLD: the gradient is not in dP yet but in the pieces the backward pass left -- the column groups' partial blocks of the
// fused recurrence (dW, db, dWhy: `fold.n_groups` blocks `fold.group_stride` floats apart, laid out like the flat block) and
// the split-K slabs of the dU product -- and is summed here, in the order gemm_fold / k_gemm_reduce use (bit-identical),
// then also stored to dP.  Saves three reduction launches and a round trip of the sums (single-GPU loop only: an
// all-reduce needs the summed block first).
// (k_slide_window's body once more as a device function: the window loop's Adagrad launch carries the NEXT window's slide in
// extra workgroups for short windows, k_adagrad<.., SLIDE>)
struct SlideArgs {
    const uint8_t *text; // null: nothing to do
    uint64_t len;
    uint64_t *pos;
    int32_t *Xr, *Tr, *headp, *xi, *ti;
    float *H, *C;
    int S, B, NB4, stride, carry_col;
};
__device__ __forceinline__ void slide_body(const SlideArgs &a, int bid, int nblk) {
    if (bid > 0) { // carry: column 0 of the next window is column `carry_col` of this one (opt:205-206: 1)
        const size_t src = (size_t)a.carry_col * a.NB4;
        for (int i = (bid - 1) * blockDim.x + threadIdx.x; i < a.NB4; i += (nblk - 1) * blockDim.x) {
            reinterpret_cast<float4 *>(a.H)[i] = reinterpret_cast<const float4 *>(a.H)[src + i];
            reinterpret_cast<float4 *>(a.C)[i] = reinterpret_cast<const float4 *>(a.C)[src + i];
        }
        return;
    }
    const int S = a.S, B = a.B;
    int head = *a.headp;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        uint64_t p = a.pos[b];
        int hd = head;
        for (int k = 0; k < a.stride; k++) { // stride > 1: the segment variant advances several bytes per window
            hd = (hd + 1) % S;
            const int last = (hd + S - 1) % S, prev = (hd + S - 2) % S;
            const int event = a.text[p];
            p++;
            if (p >= a.len) p = (uint64_t)S;
            a.Tr[last * B + b] = event;
            a.Xr[last * B + b] = a.Tr[prev * B + b];
        }
        a.pos[b] = p;
    }
    head = (head + a.stride) % S;
    __syncthreads();
    for (int i = threadIdx.x; i < S * B; i += blockDim.x) {
        const int t = i / B, b = i - t * B;
        const int row = (head + t) % S;
        a.xi[i] = a.Xr[row * B + b];
        a.ti[i] = a.Tr[row * B + b];
    }
    __syncthreads();
    if (threadIdx.x == 0) *a.headp = head;
}
struct GradFold {
    const float *gpart;  // null: no fold
    int n_groups;
    size_t group_stride; // floats
    size_t by_off4;      // float4 index where dby starts (already final in dP)
    const float *slabs;  // dU split-K slabs; null: dU is final in dP
    int n_slabs;
    size_t slab_stride; // floats
    uint2 *u6b;         // bf16 path: the scatter-form backward image of U (persistent.hip, k_pack_U6_bf16), refreshed here; or null
    int u6_uw;          // its units per workgroup
    uint2 *uf6b;        // ... and the two-half forward image (persistent.hip, k_pack_Ufwd6_bf16; QUAD launches only); or null
    int uf6_uw;
    unsigned short *why_b, *whyT_b; // bf16 path: Why as bf16 in place order [hidden][256] and transposed [256][hidden]; or null
    size_t why_off4, why_n4;        // float4 range of Why in the flat block
    SlideArgs slide;                // the next window's slide, done by the workgroups past ada_blocks (text null: none)
    int ada_blocks;
};
template <int CTRL> __device__ __forceinline__ float quad_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// QUAD (the two-half forms' images, Ufwd5 + Ubwd6): inside U the four lanes of a quad take four consecutive k of one group of
// four rows (element (rows 4rg..4rg+3, k = 4*kb4 + lane & 3) instead of consecutive row groups of one k), so that after a 4 x 4
// transpose across the quad BOTH images are written in 16-byte pieces: Ubwd6 wants four rows of a column, Ufwd5 four values of
// k of a row.  (Loads and stores of P / dP / mem stay runs of 256 bytes per sixteen lanes.)
template <bool FOLD, bool SLIDE = false, bool QUAD = false>
__global__ __launch_bounds__(256) void k_adagrad(float *__restrict__ P, float *__restrict__ dP,
                                                 float *__restrict__ mem, size_t n4, float lr, size_t u_off4, int N,
                                                 float4 *__restrict__ Ufwd, float4 *__restrict__ Ubwd,
                                                 float4 *__restrict__ Ubwd4, float4 *__restrict__ Ufwd4, GradFold fold,
                                                 int half_forms) {
    if (SLIDE && (int)blockIdx.x >= fold.ada_blocks) { // the next window's slide: touches nothing this launch reads or writes
        slide_body(fold.slide, (int)blockIdx.x - fold.ada_blocks, (int)gridDim.x - fold.ada_blocks);
        return;
    }
    const size_t stride_ = (size_t)(SLIDE ? fold.ada_blocks : (int)gridDim.x) * blockDim.x;
    const size_t u_n4 = (size_t)N * N; // float4 count of U
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride_) {
        size_t i = i0;
        int q_kb4 = 0;
        if (QUAD && i0 >= u_off4 && i0 < u_off4 + u_n4) { // (quads are aligned: every range of the flat block is a multiple of 4 float4s)
            const size_t e = i0 - u_off4;
            const int rg = (int)((e >> 2) % N);
            q_kb4 = (int)((e >> 2) / N);
            i = u_off4 + (size_t)(4 * q_kb4 + (int)(e & 3)) * N + rg;
        }
        float4 p = reinterpret_cast<float4 *>(P)[i];
        float4 d;
        if (FOLD && i < fold.by_off4) {
            const bool in_u = i >= u_off4 && i < u_off4 + u_n4;
            if (in_u && fold.slabs == nullptr) {
                d = reinterpret_cast<const float4 *>(dP)[i];
            } else {
                const float *src = in_u ? fold.slabs + 4 * (i - u_off4) : fold.gpart + 4 * i;
                const size_t stride = in_u ? fold.slab_stride : fold.group_stride;
                const int n = in_u ? fold.n_slabs : fold.n_groups;
                d = *reinterpret_cast<const float4 *>(src);
                for (int z = 1; z < n; z++) {
                    const float4 q = *reinterpret_cast<const float4 *>(src + (size_t)z * stride);
                    d.x += q.x;
                    d.y += q.y;
                    d.z += q.z;
                    d.w += q.w;
                }
                reinterpret_cast<float4 *>(dP)[i] = d;
            }
        } else {
            d = reinterpret_cast<const float4 *>(dP)[i];
        }
        float4 m = reinterpret_cast<float4 *>(mem)[i];
        p.x = adagrad1(p.x, d.x, m.x, lr);
        p.y = adagrad1(p.y, d.y, m.y, lr);
        p.z = adagrad1(p.z, d.z, m.z, lr);
        p.w = adagrad1(p.w, d.w, m.w, lr);
        reinterpret_cast<float4 *>(P)[i] = p;
        reinterpret_cast<float4 *>(mem)[i] = m;
        if (fold.why_b != nullptr && i >= fold.why_off4 && i < fold.why_off4 + fold.why_n4) {
            auto b16 = [](float v) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v); };
            const size_t f = 4 * (i - fold.why_off4); // Why[m + 256*kh], m = f % 256 .. +3
            *reinterpret_cast<uint2 *>(fold.why_b + f) = uint2{b16(p.x) | (b16(p.y) << 16), b16(p.z) | (b16(p.w) << 16)};
            const size_t m = f % 256, kh = f / 256;
            fold.whyT_b[(m + 0) * N + kh] = (unsigned short)b16(p.x);
            fold.whyT_b[(m + 1) * N + kh] = (unsigned short)b16(p.y);
            fold.whyT_b[(m + 2) * N + kh] = (unsigned short)b16(p.z);
            fold.whyT_b[(m + 3) * N + kh] = (unsigned short)b16(p.w);
        }
        if (fold.u6b != nullptr && i >= u_off4 && i < u_off4 + u_n4) {
            // four consecutive gate rows of one hidden column are one 8-byte element of Ubwd6b (same index as k_pack_U6_bf16)
            const size_t e = i - u_off4;
            const int r = 4 * (int)(e % N), out = (int)(e / N), UW = fold.u6_uw;
            const int gate = r / N, hid = r % N, kb = hid / UW, ab = (gate * UW + hid % UW) >> 2;
            const int NS = N >= 512 ? N / 512 : 1, NPW = N / (64 * NS), ws = out >> 6;
            const size_t idx = ((((size_t)kb * NPW + ws / NS) * NS + ws % NS) * UW + ab) * 64 + (out & 63);
            auto b16 = [](float v) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v); };
            fold.u6b[idx] = uint2{b16(p.x) | (b16(p.y) << 16), b16(p.z) | (b16(p.w) << 16)};
        }
        if ((Ufwd != nullptr || Ufwd4 != nullptr || (QUAD && fold.uf6b != nullptr)) && i >= u_off4 && i < u_off4 + u_n4) {
            const size_t e = i - u_off4;        // float4 index inside U: rows 4*(e % N) .. +3 of column e / N
            const int r = 4 * (int)(e % N), k = (int)(e / N);
            // Ubwd[kb][r4][l] = U[16*r4 + 4*(l>>4) + 0..3][16*kb + (l&15)]
            if (Ubwd != nullptr) Ubwd[((size_t)(k >> 4) * (N / 4) + (r >> 4)) * 64 + (((r & 15) >> 2) << 4) + (k & 15)] = p;
            if (Ubwd4 != nullptr) {
                float *u4 = reinterpret_cast<float *>(Ubwd4);
                if (half_forms & 4) { // Ubwd6: four consecutive gate rows of one hidden column are one 16-byte piece of the image
                    *reinterpret_cast<float4 *>(u4 + ubwd6_index(r, k, N)) = p;
                } else {
                    u4[ubwd45_index(r + 0, k, N, half_forms)] = p.x;
                    u4[ubwd45_index(r + 1, k, N, half_forms)] = p.y;
                    u4[ubwd45_index(r + 2, k, N, half_forms)] = p.z;
                    u4[ubwd45_index(r + 3, k, N, half_forms)] = p.w;
                }
            }
            if (QUAD) { // lane q of the quad: row r + q, k = 4*kb4 .. +3 after the transpose
                const int ta = threadIdx.x & 3;
                float t0 = p.x, t1 = p.y, t2 = p.z, t3 = p.w;
                {
                    const float lo = (ta & 1) ? t0 : t1, hi = (ta & 1) ? t2 : t3;
                    const float rlo = quad_dpp<0xB1>(lo), rhi = quad_dpp<0xB1>(hi); // quad_perm [1,0,3,2]
                    if (ta & 1) {
                        t0 = rlo;
                        t2 = rhi;
                    } else {
                        t1 = rlo;
                        t3 = rhi;
                    }
                    const float s0 = (ta & 2) ? t0 : t2, s1 = (ta & 2) ? t1 : t3;
                    const float r0 = quad_dpp<0x4E>(s0), r1 = quad_dpp<0x4E>(s1); // quad_perm [2,3,0,1]
                    if (ta & 2) {
                        t0 = r0;
                        t1 = r1;
                    } else {
                        t2 = r0;
                        t3 = r1;
                    }
                }
                if (Ufwd4 != nullptr)
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(Ufwd4) + ufwd5_index(r + ta, 4 * q_kb4, N)) = float4{t0, t1, t2, t3};
                if (fold.uf6b != nullptr) { // the same four values as one 8-byte element of Ufwd6b (index as in k_pack_Ufwd6_bf16)
                    const int row = r + ta, k0 = 4 * q_kb4, UW = fold.uf6_uw;
                    const int gate = row / N, hid = row % N, kb = hid / UW, sx = (hid % UW) >> 4, ll = 4 * (hid & 15) + gate;
                    const int Kw = N / 8, NRK = Kw >= 64 ? Kw / 64 : 1, NAB = Kw >= 64 ? 16 : Kw / 4, NSET = UW / 16;
                    const int w = k0 / Kw, kk = k0 % Kw, rr = kk / 64, ab = (kk % 64) >> 2;
                    auto b16 = [](float v) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v); };
                    fold.uf6b[(((((size_t)kb * 8 + w) * NRK + rr) * NSET + sx) * NAB + ab) * 64 + ll] =
                        uint2{b16(t0) | (b16(t1) << 16), b16(t2) | (b16(t3) << 16)};
                }
            } else if (Ufwd4 != nullptr) {
                float *f4 = reinterpret_cast<float *>(Ufwd4);
                f4[ufwd45_index(r + 0, k, N, half_forms)] = p.x;
                f4[ufwd45_index(r + 1, k, N, half_forms)] = p.y;
                f4[ufwd45_index(r + 2, k, N, half_forms)] = p.z;
                f4[ufwd45_index(r + 3, k, N, half_forms)] = p.w;
            }
            if (Ufwd == nullptr) continue;
            // Ufwd[jb][k4][l].i = U[(l&3)*N + 4*jb + ((l&15)>>2)][16*k4 + 4*(l>>4) + i]
            const int gate = r / N, hid = r % N; // 4 rows share the gate (N % 4 == 0)
            const int k4 = k >> 4, kq = (k & 15) >> 2, ki = k & 3;
            float *uf = reinterpret_cast<float *>(Ufwd);
            const float pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
            for (int dlt = 0; dlt < 4; dlt++) {
                const int h = hid + dlt, jb = h >> 2, jj = h & 3;
                const int l = (kq << 4) | (jj << 2) | gate;
                uf[(((size_t)jb * (N / 16) + k4) * 64 + l) * 4 + ki] = pv[dlt];
            }
        }
    }
}
void adagrad(float *P, float *dP, float *mem, size_t n, float lr, size_t u_off, int N, float4 *Ufwd, float4 *Ubwd,
             hipStream_t st, float4 *Ubwd4, float4 *Ufwd4, const float *gpart, int n_groups, size_t group_stride, size_t by_off,
             const float *slabs, int n_slabs, size_t slab_stride, int half_forms, void *u6b, int u6_uw,
             unsigned short *why_b, unsigned short *whyT_b, size_t why_off, const SlideJob *slide, void *uf6b, int uf6_uw) {
    const size_t n4 = n / 4; // the flat block is a multiple of 4 floats (M = 256, N % 16 == 0)
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    SlideArgs sl{};
    int extra = 0;
    if (slide != nullptr) {
        const int nb4 = slide->N * slide->B / 4;
        int copy_blocks = (nb4 + 255) / 256;
        if (copy_blocks > 128) copy_blocks = 128;
        sl = SlideArgs{slide->text, slide->len, slide->pos, slide->Xr, slide->Tr, slide->headp, slide->xi, slide->ti, slide->H, slide->C,
                       slide->S, slide->B, nb4, slide->stride, slide->carry_col};
        extra = 1 + copy_blocks;
    }
    const GradFold fold{gpart, n_groups, group_stride, by_off / 4, slabs, n_slabs, slab_stride, reinterpret_cast<uint2 *>(u6b), u6_uw, reinterpret_cast<uint2 *>(uf6b), uf6_uw, why_b, whyT_b, why_off / 4, (size_t)256 * N / 4, sl, blocks};
    blocks += extra;
    static const bool quad_off = getenv("LSTM_HIP_ADAGRAD_QUAD") && atoi(getenv("LSTM_HIP_ADAGRAD_QUAD")) == 0; // A/B
    const bool quad = !quad_off && Ufwd == nullptr && Ubwd == nullptr &&
                      ((Ufwd4 != nullptr && Ubwd4 != nullptr && (half_forms & 1) && (half_forms & 4)) || // fp32 two-half forms
                       (Ufwd4 == nullptr && Ubwd4 == nullptr && uf6b != nullptr));                          // bf16 two-half forms
#define ADA_GO(F, S_, Q) \
    hipLaunchKernelGGL((k_adagrad<F, S_, Q>), dim3(blocks), dim3(256), 0, st, P, dP, mem, n4, lr, u_off / 4, N, Ufwd, Ubwd, Ubwd4, Ufwd4, fold, half_forms)
    const bool f = gpart != nullptr, sl_ = extra != 0;
    if (f && sl_ && quad) ADA_GO(true, true, true);
    else if (f && sl_) ADA_GO(true, true, false);
    else if (f && quad) ADA_GO(true, false, true);
    else if (f) ADA_GO(true, false, false);
    else if (sl_ && quad) ADA_GO(false, true, true);
    else if (sl_) ADA_GO(false, true, false);
    else if (quad) ADA_GO(false, false, true);
    else ADA_GO(false, false, false);
#undef ADA_GO
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
                                                       int NB4, int stride, int carry_col) {
    if (blockIdx.x > 0) { // carry: column 0 of the next window is column `carry_col` of this one (opt:205-206: 1)
        const size_t src = (size_t)carry_col * NB4;
        for (int i = (blockIdx.x - 1) * blockDim.x + threadIdx.x; i < NB4; i += (gridDim.x - 1) * blockDim.x) {
            reinterpret_cast<float4 *>(H)[i] = reinterpret_cast<const float4 *>(H)[src + i];
            reinterpret_cast<float4 *>(C)[i] = reinterpret_cast<const float4 *>(C)[src + i];
        }
        return;
    }
    int head = *headp;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        uint64_t p = pos[b];
        int hd = head;
        for (int k = 0; k < stride; k++) { // stride > 1: the segment variant advances several bytes per window
            hd = (hd + 1) % S;
            const int last = (hd + S - 1) % S, prev = (hd + S - 2) % S;
            const int event = text[p];
            p++;
            if (p >= len) p = (uint64_t)S;
            Tr[last * B + b] = event;
            Xr[last * B + b] = Tr[prev * B + b];
        }
        pos[b] = p;
    }
    head = (head + stride) % S;
    __syncthreads();
    for (int i = threadIdx.x; i < S * B; i += blockDim.x) {
        const int t = i / B, b = i - t * B;
        const int row = (head + t) % S;
        xi[i] = Xr[row * B + b];
        ti[i] = Tr[row * B + b];
    }
    __syncthreads();
    if (threadIdx.x == 0) *headp = head;
}
void slide_window(const uint8_t *text, uint64_t len, uint64_t *pos, int32_t *Xr, int32_t *Tr, int32_t *headp,
                  int32_t *xi, int32_t *ti, float *H, float *C, int S, int B, int N, int stride, int carry_col,
                  hipStream_t st) {
    const int nb4 = N * B / 4;
    int copy_blocks = (nb4 + 1023) / 1024;
    if (copy_blocks > 32) copy_blocks = 32;
    hipLaunchKernelGGL(k_slide_window, dim3(1 + copy_blocks), dim3(1024), 0, st, text, len, pos, Xr, Tr, headp, xi, ti, H,
                       C, S, B, nb4, stride, carry_col);
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
// One character of the sampler for the multi-workgroup path (lstm_hip_api.cpp: per character this kernel, then one
// k_fwd_step launch with the sampled byte as input): probabilities from h exactly as b1_output / k_sample compute them,
// then the sequential float CDF walk of R/lstm.cc:321-338.  The byte goes to out[0] and, as the next input index, to
// x_next[0].
__global__ __launch_bounds__(256) void k_sample_head(const float *__restrict__ Why, const float *__restrict__ by, int N,
                                                     const float *__restrict__ hvec, const double *__restrict__ u,
                                                     uint8_t *__restrict__ out, int32_t *__restrict__ x_next) {
    __shared__ float ps[256];
    __shared__ float hs[1024];
    const int m = threadIdx.x;
    for (int k = m; k < N; k += 256) hs[k] = hvec[k];
    __syncthreads();
    float y = 0.0f;
    for (int k0 = 0; k0 < N; k0 += 16) { // 16 loads in flight; the additions stay in k order (as b1_output)
        float wv[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wv[i] = Why[(size_t)(k0 + i) * 256 + m];
#pragma unroll
        for (int i = 0; i < 16; i++) y += wv[i] * hs[k0 + i];
    }
    ps[m] = expf(y + by[m]);
    __syncthreads();
    if (m == 0) {
        float s = 0.0f;
        for (int i = 0; i < 256; i++) s += ps[i];
        const float r = (float)u[0];
        float cdf = 0.0f;
        int index = 0;
        for (int i = 0; i < 256; i++) {
            cdf += ps[i] / s;
            if (r < cdf) {
                index = i;
                break;
            }
        }
        out[0] = (uint8_t)index;
        x_next[0] = index;
    }
}
void sample_head(const float *Why, const float *by, int N, const float *hvec, const double *u, uint8_t *out, int32_t *x_next,
                 hipStream_t st) {
    hipLaunchKernelGGL(k_sample_head, dim3(1), dim3(256), 0, st, Why, by, N, hvec, u, out, x_next);
}
void sample(const float *P, int N, float *hc, const double *u, int count, uint8_t *out, float *, hipStream_t st) {
    const size_t lds = (size_t)(6 * N + 256) * sizeof(float);
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(1024), lds, st, P, N, hc, u, count, out);
}

} // namespace lstmk
