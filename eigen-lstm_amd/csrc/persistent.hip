// persistent.hip -- the two recurrences of a window as ONE launch each (gfx950).
//
// Why: a timestep of the reference is  g = U*h_prev (+gather, bias) -> gates -> c,h  (R/lstm.cc:176-192)
// and, going back,  dhnext = U^T*dg -> dc,dg (R/lstm.cc:228-256).  At batch 64 that is ~134 MFLOP,
// <1 us of MFMA, and the only true dependency between steps is the 128 KB h (or 512 KB dg) vector.
// Launching one kernel per step re-reads the recurrent weights from L2 every step (32 MB/step across
// the chip, ~10 us/step measured) and pays a kernel boundary per step.  Here the weights stay in
// REGISTERS for the whole window and the steps are chained inside the launch by a device-scope
// hand-off.
//
// This file holds several generations of that idea; the host side (end of the file) picks one per shape:
//   forward   k_fwd_persistent6   N = 512 / 256, 8-column groups: two alternating 4-column recurrences per workgroup, 4x4x1 MFMA
//                                 blocks, data-as-flag ring, no workgroup barrier                         (the headline shape)
//             k_fwd_persistent4   N = 1024, 8-column groups: one recurrence per workgroup, same ring, one barrier
//             k_fwd_persistent2   second form: 8 units x 16 columns on 16x16x4 tiles, counters   (small N, B <= 8, bf16 twin)
//             k_fwd_persistent    first form: 4 units x 16 columns, counters                     (N = 64 multiples, bf16 twin)
//             k_fwd_halves_bf16   bf16 path, N = 256 / 512 / 1024: the two-half form on 4x4x4 bf16 blocks (32 units per workgroup
//                                 at N = 1024, so that a group fits one XCD)
//             k_small_fwd         one stream, N <= 128: the whole recurrence on one CU, no hand-off between CUs
//   backward  k_bwd_scatter       N = 512 / 256, 8-column groups: two alternating 4-column recurrences, partial sums scattered to
//                                 the owners of the outputs; side waves for the output-layer term and the dW sums  (the headline shape)
//             k_bwd_scatter_bf16  bf16 path, N = 256 / 512 / 1024: the scatter form, unfused, phase-tagged ring without reset stores
//             k_small_bwd         one stream, N <= 128: one CU
//             k_bwd_persistent    everything else: 16x16x4 tiles or 4x4x1 blocks, 4 / 8 / 16-column groups, fp32 or bf16,
//                                 sharded counters, optional fused sums
// Placement of the two-half / scatter forms: a launch takes as many column groups as are co-resident at one workgroup per CU;
// fewer than 8 groups are pinned one to an XCD (the other XCDs' workgroups return at once); a narrow batch runs one half (4
// columns) per workgroup on twice the workgroups; a batch wider than a launch runs as launches over column ranges.
//
// Hand-off, counter form (first / second forward forms, k_bwd_persistent; cdna_hip_programming.md Guideline 16): the
// producing wave stores its slice of h_t with sc1 (write-through) 16-byte stores, drains them (s_waitcnt vmcnt(0)), then ONE
// lane does a relaxed agent-scope atomic add on a sharded counter (cnt[t][g][shard]) so arrivals do not serialise on one
// address.  A consumer's wave 0 polls the shards with sc1 loads until every shard has all its arrivals, a workgroup barrier
// follows, and only then does any wave read h_t, with sc1 loads (never through L1).  Counters are never reset: launch number
// e waits for e x arrivals.
// Hand-off, data-as-flag form (k_fwd_persistent4 / 6, k_bwd_scatter): see HX_RING below -- a ring of step slots whose words
// hold a sentinel until the value is stored; consumers re-issue the loads of their own K-slice until no word is the
// sentinel.  No counters, no drain, no barrier ahead of the loads.
// When a column group's workgroups are verified (per launch, HW_REG_XCC_ID) to sit on one XCD, the payload is published
// with plain stores that stay in that XCD's L2.
// Every spin is bounded; on time-out a global abort word makes every workgroup leave, and the host reports it.
//
// Residency: all workgroups must be co-resident (they wait on each other); the host checks the grid
// against the occupancy of the device and falls back to the per-step engine otherwise.
#include <type_traits>
#include "kernels.h"

#include <cstdlib>

namespace lstmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#pragma clang fp contract(off)
// Gate math of the persistent kernels.  The default forms are built on v_exp_f32 / v_rcp_f32 (1 ulp each) with a
// compensated argument, about 2 ulp overall against 1 ulp for the libm calls the per-step engine keeps (kernels.hip);
// the gates sit on the hand-off chain, where the libm forms (IEEE division, branchy tanhf) cost ~300 cycles a step more.
// tanh is (1-e)/(1+e), e = exp(-2|x|): its ABSOLUTE error stays below 1.2e-7; near zero the relative error does not.
// tests/test_hip_parity.py::test_gate_math_accuracy pins both against float64.  FAST: plain v_exp of x*log2(e).
__device__ __forceinline__ float p_exp_neg(float x) { // exp(-x)
    const float hi = 1.44269502162933349609375f, lo = 1.925963033500011e-08f; // log2(e) = hi + lo
    const float t = -x * hi;
    const float r = __builtin_fmaf(-x, hi, -t) - x * lo; // -x*log2(e) = t + r
    return __builtin_amdgcn_exp2f(t) * (1.0f + r * 0.693147180559945f);
}
template <bool FAST> __device__ __forceinline__ float p_sigm(float x) {
    if (FAST) return __frcp_rn(1.0f + __expf(-x));
    return __builtin_amdgcn_rcpf(1.0f + p_exp_neg(x));
}
template <bool FAST> __device__ __forceinline__ float p_tanh(float x) {
    if (FAST) return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * x) + 1.0f);
    const float e = p_exp_neg(2.0f * __builtin_fabsf(x));
    return __builtin_copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

// the value of `v` in lane l ^ 16 / l ^ 32, on the vector ALU (v_permlane16_swap / v_permlane32_swap; a ds_bpermute through
// the LDS crossbar costs ~10x the latency): both operands of the swap are v, so the odd rows / upper half find the partner's
// value in the first result and the even rows / lower half in the second
__device__ __forceinline__ float xchg_row16(float v, int l) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return __builtin_bit_cast(float, (l & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xchg_half32(float v, int l) {
    const unsigned x = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __builtin_bit_cast(float, (l & 32) ? r[0] : r[1]);
}

// DPP lane move (quad_perm / row shifts): the value of `v` in the lane the control word names; all lanes must be active
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// fragment loads kept in flight ahead of the MFMAs (see the pipelines below); -D overrides for A/B builds
#ifndef XCD_LOCAL
#define XCD_LOCAL 1 // backward recurrence: plain-store hand-off when a column group is verified to sit on one XCD
#endif
#ifndef XCD_FORCE_LOCAL
#define XCD_FORCE_LOCAL 0 // test builds only: skip the verification (wrong results when groups span XCDs)
#endif
#ifndef BWD_PF
#define BWD_PF 6
#endif
#ifndef FWD_PF
#define FWD_PF 4
#endif

#ifndef CNT_STRIDE_U
#define CNT_STRIDE_U 16
#endif
constexpr int CNT_STRIDE = CNT_STRIDE_U; // uints between shards: one 64-byte line each
// Counter shards per (step, group): a power of two <= 64 (one poll instruction reads them all).  Arrivals and polls
// contend on the counter lines: with 128 producers per group (forward, N=512) 64 shards measured 388 us against 419
// with 8 (and 497 with one); the backward recurrence has 32 producers per group and is flat from 8 up.
// Also measured: one flag word per producer instead of a counter (dense 494 us, one line each 440 us) and every wave
// polling for itself instead of poll -> barrier (1275 us): the polls themselves load the counter lines.
// Workgroup -> (tile, column group): blocks are dealt round-robin over the 8 XCDs in dispatch order (observed, not
// promised: speed only).  GROUP_REMAP=1 makes blocks b and b+NG members of one group, so a group's hand-off lines
// are fetched into one or two XCD L2s instead of all eight (backward 504 -> 492 us, forward unchanged).
#ifndef GROUP_REMAP
#define GROUP_REMAP 1
#endif
#ifndef FWD_SH
#define FWD_SH 64
#endif
#ifndef BWD_SH
#define BWD_SH 8
#endif
constexpr int CNT_SLOTS = 64;
constexpr int SPIN_LIMIT = 1 << 21;  // bounded spin; ~seconds

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, size_t bytes) {
    const unsigned n = bytes > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float4 ld_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16); // aux 16 = sc1
    float4 f;
    f.x = __uint_as_float(v.x);
    f.y = __uint_as_float(v.y);
    f.z = __uint_as_float(v.z);
    f.w = __uint_as_float(v.w);
    return f;
}
__device__ __forceinline__ void st_sc1(float4 f, __amdgpu_buffer_rsrc_t r, int byte_off) {
    u32x4 v;
    v.x = __float_as_uint(f.x);
    v.y = __float_as_uint(f.y);
    v.z = __float_as_uint(f.z);
    v.w = __float_as_uint(f.w);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}

// wave-level wait until all `n_prod` producers (sharded by id & 7) have arrived at `cp`.
// Returns false on time-out / abort.  Called by one whole wave.
// Counters are never reset: launch number `epoch` (1, 2, ...) waits for epoch * (arrivals per launch).
#ifndef POLL_SLEEP
#define POLL_SLEEP 1
#endif
template <int CNT_SH>
__device__ __forceinline__ bool wait_arrivals(const unsigned *cp, int n_prod, unsigned epoch, unsigned *abortp, int lane) {
    const unsigned expect = lane < CNT_SH ? epoch * (unsigned)((n_prod - lane + CNT_SH - 1) / CNT_SH) : 0u;
    for (int spins = 0;; spins++) {
        unsigned v = 0;
        if (lane < CNT_SH) v = __hip_atomic_load(cp + lane * CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= expect)) return true;
        if (spins > SPIN_LIMIT) break;
        if ((spins & 255) == 255 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        __builtin_amdgcn_s_sleep(POLL_SLEEP);
    }
    if (lane == 0) __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// ------------------------------------------------------------------------------------------------
// forward recurrence, t = 1..S-1, N = 64*NK4W.  grid (N/4, ceil(B/16)), 256 threads.
// ------------------------------------------------------------------------------------------------
template <int NK4W, bool FAST>
__global__ __launch_bounds__(256) void k_fwd_persistent(const float4 *__restrict__ Ufwd, const float *__restrict__ W,
                                                        const float *__restrict__ bias, float *H, float *__restrict__ C,
                                                        float *__restrict__ G, const int32_t *__restrict__ xi,
                                                        unsigned *cnt, unsigned *abortp, unsigned epoch, int S, int B) {
    constexpr int N = 64 * NK4W, G4 = 4 * N, nk4 = N / 16;
    __shared__ float red[4 * 4 * 64];
    __shared__ int s_abort;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NB = gridDim.x, NG = gridDim.y;
    const int lin_ = blockIdx.x + NB * blockIdx.y;
    const int p = GROUP_REMAP ? lin_ / NG : (int)blockIdx.x, g = GROUP_REMAP ? lin_ % NG : (int)blockIdx.y;
    const int q = l >> 4, c = l & 15;
    const int col = 16 * g + c, colc = col < B ? col : B - 1;
    const int j = 4 * p + q;

    float4 a[NK4W];
#pragma unroll
    for (int i = 0; i < NK4W; i++) a[i] = Ufwd[((size_t)p * nk4 + w * NK4W + i) * 64 + l];
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (w == 0) {
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = bias[gt * N + j];
        cprev = C[(size_t)colc * N + j];
    }
    const __amdgpu_buffer_rsrc_t rH = make_rsrc(H, (size_t)S * N * B * sizeof(float));
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    for (int t = 1; t < S; t++) {
        float wx[4] = {0.f, 0.f, 0.f, 0.f};
        if (w == 0) {
            const int x = xi[t * B + colc];
            if (x >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = W[(size_t)x * G4 + gt * N + j];
            }
            if (t > 1) {
                const unsigned *cp = cnt + (size_t)((t - 1) * NG + g) * CNT_SLOTS * CNT_STRIDE;
                if (!wait_arrivals<FWD_SH>(cp, NB, epoch, abortp, l) && l == 0) s_abort = 1;
            }
        }
        __syncthreads();
        if (s_abort) return;

        const int off = (int)((((size_t)(t - 1) * B + colc) * N + 16 * (w * NK4W) + 4 * q) * sizeof(float));
        // Software pipeline: PF fragment loads in flight ahead of the MFMAs (the scheduler alone keeps two;
        // see k_bwd_persistent).
        constexpr int PF = FWD_PF < NK4W ? FWD_PF : NK4W;
        float4 b[NK4W];
#pragma unroll
        for (int i = 0; i < PF; i++) b[i] = ld_sc1(rH, off + 64 * i);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NK4W; i++) {
            if (i + PF < NK4W) b[i + PF] = ld_sc1(rH, off + 64 * (i + PF));
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
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) red[(w * 4 + r) * 64 + l] = acc0[r] + acc1[r];
        __syncthreads();

        if (w == 0) {
            float pre[4];
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                const float uh = ((red[(0 * 4 + gt) * 64 + l] + red[(1 * 4 + gt) * 64 + l]) + red[(2 * 4 + gt) * 64 + l]) +
                                 red[(3 * 4 + gt) * 64 + l];
                pre[gt] = (wx[gt] + uh) + bs[gt]; // R/lstm.cc:176
            }
            const float ig = p_sigm<FAST>(pre[0]), og = p_sigm<FAST>(pre[1]), fg = p_sigm<FAST>(pre[2]); // :179
            const float ug = p_tanh<FAST>(pre[3]);                                                        // :182
            const float cv = p_tanh<FAST>(ig * ug + fg * cprev);                                          // :185-189
            const float hv = og * cv;                                                                     // :192
            cprev = cv;
            // publish h_t first (it is the only thing the next step of other workgroups waits for)
            float4 h4;
            h4.x = __shfl(hv, c, 64);
            h4.y = __shfl(hv, 16 + c, 64);
            h4.z = __shfl(hv, 32 + c, 64);
            h4.w = __shfl(hv, 48 + c, 64);
                if (q == 0 && col < B) st_sc1(h4, rH, (int)((((size_t)t * B + col) * N + 4 * p) * sizeof(float)));
            if (t + 1 < S) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (l == 0)
                    __hip_atomic_fetch_add(cnt + ((size_t)(t * NG + g) * CNT_SLOTS + (p & (FWD_SH - 1))) * CNT_STRIDE, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
            if (col < B) {
                float *gc = G + ((size_t)t * B + col) * G4 + j;
                gc[0] = ig;
                gc[N] = og;
                gc[2 * N] = fg;
                gc[3 * N] = ug;
                C[((size_t)t * B + col) * N + j] = cv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward recurrence, second form: grid (N/8, ceil(B/16)), 512 threads, one workgroup per CU at the headline shape.
// A workgroup owns 8 units (two of the row tiles above, same packed image) and all of K over 8 waves, so every
// fragment of h_{t-1} a lane loads feeds two MFMA tiles: half the load instructions per lane and half the bytes per
// CU of the first form (a timing probe with half the loads removed from the first form: 387 -> 350 us).  Waves 0
// and 1 each finish one row tile (gates, cell, publish, arrive): to the counters they are the producers 2p and
// 2p+1 of the first form, so the counter protocol and the backward kernel see no difference.
// N = 128*NKW.
// ------------------------------------------------------------------------------------------------
template <int NKW, bool FAST>
__global__ __launch_bounds__(512) void k_fwd_persistent2(const float4 *__restrict__ Ufwd, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *H, float *__restrict__ C,
                                                         float *__restrict__ G, const int32_t *__restrict__ xi,
                                                         unsigned *cnt, unsigned *abortp, unsigned epoch, int S, int B) {
    constexpr int N = 128 * NKW, G4 = 4 * N, nk4 = N / 16;
    __shared__ float red[8 * 2 * 4 * 64];
    __shared__ int s_abort;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NB2 = gridDim.x, NG = gridDim.y;
    const int lin_ = blockIdx.x + NB2 * blockIdx.y;
    const int p2 = GROUP_REMAP ? lin_ / NG : (int)blockIdx.x, g = GROUP_REMAP ? lin_ % NG : (int)blockIdx.y;
    const int q = l >> 4, c = l & 15;
    const int col = 16 * g + c, colc = col < B ? col : B - 1;
    const int p = 2 * p2 + (w & 1); // the row tile a gating wave (w < 2) finishes
    const int j = 4 * p + q;

    float4 a0[NKW], a1[NKW];
#pragma unroll
    for (int i = 0; i < NKW; i++) {
        a0[i] = Ufwd[((size_t)(2 * p2) * nk4 + w * NKW + i) * 64 + l];
        a1[i] = Ufwd[((size_t)(2 * p2 + 1) * nk4 + w * NKW + i) * 64 + l];
    }
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (w < 2) {
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = bias[gt * N + j];
        cprev = C[(size_t)colc * N + j];
    }
    const __amdgpu_buffer_rsrc_t rH = make_rsrc(H, (size_t)S * N * B * sizeof(float));
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();

    for (int t = 1; t < S; t++) {
        float wx[4] = {0.f, 0.f, 0.f, 0.f};
        if (w < 2) {
            const int x = xi[t * B + colc];
            if (x >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = W[(size_t)x * G4 + gt * N + j];
            }
        }
        if (w == 0 && t > 1) {
            const unsigned *cp = cnt + (size_t)((t - 1) * NG + g) * CNT_SLOTS * CNT_STRIDE;
            if (!wait_arrivals<FWD_SH>(cp, 2 * NB2, epoch, abortp, l) && l == 0) s_abort = 1;
        }
        __syncthreads();
        if (s_abort) return;

        const int off = (int)((((size_t)(t - 1) * B + colc) * N + 16 * (w * NKW) + 4 * q) * sizeof(float));
        constexpr int PF = FWD_PF < NKW ? FWD_PF : NKW;
        float4 b[NKW];
#pragma unroll
        for (int i = 0; i < PF; i++) b[i] = ld_sc1(rH, off + 64 * i);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKW; i++) {
            if (i + PF < NKW) b[i + PF] = ld_sc1(rH, off + 64 * (i + PF));
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].x, b[i].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].x, b[i].x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].y, b[i].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].y, b[i].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].z, b[i].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].z, b[i].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i].w, b[i].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i].w, b[i].w, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            red[((w * 2 + 0) * 4 + r) * 64 + l] = acc0[r];
            red[((w * 2 + 1) * 4 + r) * 64 + l] = acc1[r];
        }
        __syncthreads();

        if (w < 2) {
            float pre[4];
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                float uh = red[((0 * 2 + w) * 4 + gt) * 64 + l];
#pragma unroll
                for (int ww = 1; ww < 8; ww++) uh += red[((ww * 2 + w) * 4 + gt) * 64 + l];
                pre[gt] = (wx[gt] + uh) + bs[gt]; // R/lstm.cc:176
            }
            const float ig = p_sigm<FAST>(pre[0]), og = p_sigm<FAST>(pre[1]), fg = p_sigm<FAST>(pre[2]); // :179
            const float ug = p_tanh<FAST>(pre[3]);                                                        // :182
            const float cv = p_tanh<FAST>(ig * ug + fg * cprev);                                          // :185-189
            const float hv = og * cv;                                                                     // :192
            cprev = cv;
            float4 h4;
            h4.x = __shfl(hv, c, 64);
            h4.y = __shfl(hv, 16 + c, 64);
            h4.z = __shfl(hv, 32 + c, 64);
            h4.w = __shfl(hv, 48 + c, 64);
            if (q == 0 && col < B) st_sc1(h4, rH, (int)((((size_t)t * B + col) * N + 4 * p) * sizeof(float)));
            if (t + 1 < S) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0)
                    __hip_atomic_fetch_add(cnt + ((size_t)(t * NG + g) * CNT_SLOTS + (p & (FWD_SH - 1))) * CNT_STRIDE, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
            if (col < B) {
                float *gc = G + ((size_t)t * B + col) * G4 + j;
                gc[0] = ig;
                gc[N] = og;
                gc[2 * N] = fg;
                gc[3 * N] = ug;
                C[((size_t)t * B + col) * N + j] = cv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward recurrence, 8-column form (the default at B > 8): grid (N/16, ceil(B/8)), 512 threads, one workgroup per CU.
// A workgroup owns 16 units (64 gate rows) for a group of EIGHT batch columns, so a column group is N/16 = 32 workgroups
// at N = 512: one XCD's worth.  8-column groups on 16x16x4 tiles would waste half the matrix pipe; v_mfma_f32_4x4x1 with
// operand broadcast does not: block = 8x + u (lane = 4*block + i, u = 0..7) computes
//   D[i][j] += h[k][column 4x+i] * U[gate j of unit 8*pass + u][k]                 (pass = 0, 1)
// CBSZ = 3 / ABID = s makes all eight u-blocks of a half read h from slot s of the loaded register (one register = 8
// values of k x 8 columns = 8 instructions), BLGP = 1 / 2 makes both x-halves read the weights from one half of the
// weight register (2 values of k each).  Every lane carries data and no partial sums need folding across lanes.  K is
// split over waves 0-7; two further waves (8, 9) finish the 128 (unit, column) pairs.  N = 256*NKQ, weights Ufwd4.
//
// Hand-off: THE DATA IS THE FLAG (Guideline 16, recipe R2, with the tag folded into the value space) -- no counter, no
// drain, no poll -> barrier -> load sequence.  h_t is published into a ring of HX_RING = 4 step slots Hx[slot][B][N]
// whose words hold the sentinel 0xFFFFFFFF (a NaN pattern no computed h can have; a NaN input with that payload is
// canonicalised on the way in) until the producer overwrites them, each word exactly once per use of the slot.  A
// consuming WAVE re-issues the sc1 loads of its own K-slice -- the loads it needs anyway -- until no word is the
// sentinel, then goes straight to its MFMAs: per step one store -> L2 -> load hop instead of store, drain, atomic,
// poll, barrier, load (measured against that counter protocol on the same decomposition: 312 -> 259 us, bit-identical).
// Torn 16-byte stores cannot matter: every dword is checked by the lane that consumes it.
//
// Slot reuse.  Step t reads slot(t-1), publishes into slot(t) and, right after publishing, resets its own words of
// slot(t+2) to the sentinel (slot(s) = (s + ring_base) & 3).  Safe because
//   * the reset happens after the workgroup barrier of step t, which every wave reaches only after its poll of
//     slot(t-1) succeeded, i.e. after EVERY workgroup of the group has published h_{t-1}, which each does after its
//     own barrier of step t-1, i.e. after it finished reading slot(t-2) = slot(t+2): nobody still reads the old words;
//   * a wave polls producer P's words of slot(t+2) (in step t+3) only after it consumed P's words of slot(t+1), which
//     P stored after an s_waitcnt vmcnt(0) that covers the reset it issued a step earlier: the reset is visible first.
// At the end of a launch slots S and S+1 hold the sentinel; the next launch starts with ring_base advanced by S-1 so
// that these are its slots 1 and 2 (the two that no in-launch reset precedes).  The host fills the ring with the
// sentinel once (and after an abort).  Step 1 reads the carry column H[0], written before the launch.
// The time-batched products read the plain H, stored off the chain.  One workgroup barrier per step (the K-slice
// reduction: waves 0-7 arrive with their partial sums written, waves 8-9 leave to fold them), `red` double-buffered by
// step parity: a product wave writes red[t & 1] again in step t+2, after the barrier of step t+1, which the gating
// waves join only after they have read step t's sums.
//
// XCD-local publish (speed only, checked every launch): when all workgroups of a column group run on ONE XCD, h_t can be
// published with plain stores -- the lines stay in that XCD's L2, which serves the group's sc1 loads directly -- instead
// of sc1 write-through stores that every consumer pulls back over the fabric.  Placement is observed, not promised, so
// each workgroup publishes its HW_REG_XCC_ID (sc1, drained before its first publish) in the step-0 counter slots of its
// group; once h_1 of every member has arrived the gating waves read the table and take the plain-store path only if
// all entries agree and were written in this launch.  (Control experiment: DESIGN.md.)
//
// STAMP builds (LSTM_HIP_DEBUG_STAMPS, N = 512) record s_memtime at the points marked FSTAMP for two workgroups; the
// stamps go to a buffer nothing else reads and the shipped path is the STAMP = false instantiation.
// ------------------------------------------------------------------------------------------------
constexpr unsigned HX_SENT = 0xFFFFFFFFu;
constexpr int HX_RING = 4;
__device__ __forceinline__ bool hx_ready(const float4 &v) {
    return __float_as_uint(v.x) != HX_SENT && __float_as_uint(v.y) != HX_SENT && __float_as_uint(v.z) != HX_SENT &&
           __float_as_uint(v.w) != HX_SENT;
}
__device__ __forceinline__ float hx_canon(float v) { return __float_as_uint(v) == HX_SENT ? __uint_as_float(0x7FC00000u) : v; }
#ifndef BWDS_TAGGED
#define BWDS_TAGGED 0
#endif
// phase-tagged variant of the ring (no reset stores; k_bwd_scatter_bf16 has the description): the parity of a slot's use count
// in the last mantissa bit of every word
__device__ __forceinline__ float tag_mark(float v, unsigned phase) { return __uint_as_float((__float_as_uint(v) & ~1u) | phase); }
__device__ __forceinline__ float tag_value(float v) { return __uint_as_float(__float_as_uint(v) & ~1u); }
__device__ __forceinline__ bool tag_ready(const float4 &v, unsigned phase) {
    return ((__float_as_uint(v.x) & __float_as_uint(v.y) & __float_as_uint(v.z) & __float_as_uint(v.w) & 1u) == phase) &&
           (((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) & 1u) == phase);
}

// stamp slots [workgroup 0 | gridDim.x/2][t][16]: wave 3 (MFMA) 8-12, wave 8 (gating) 0-7
#define FSTAMP(wave, k)                                                                                        \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = __builtin_amdgcn_s_memtime();
#define FSTAMP_VAL(wave, k, v)                                                                                 \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = (unsigned long long)(v);
constexpr int FWD4_THREADS = 640; // waves 0-7: the product (K split eight ways); waves 8, 9: gates, cell, publish
template <int NKQ, bool FAST, bool STAMP = false>
__global__ __launch_bounds__(FWD4_THREADS) void k_fwd_persistent4(const float4 *__restrict__ Ufwd4, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float *H, float *__restrict__ C,
                                                         float *__restrict__ G, const int32_t *__restrict__ xi, float *Hx,
                                                         unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base,
                                                         int S, int B, int poll_cfg, unsigned long long *stamps = nullptr) {
    constexpr int N = 256 * NKQ, G4 = 4 * N, Kw = N / 8, NL = Kw / 32;
    // partial sums [step parity][wave][pair = column*16 + unit][gate]: the four gates of a pair side by side (16-byte reads)
    __shared__ __attribute__((aligned(16))) float red[2][8 * 128 * 4];
    __shared__ int s_abort;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int NB3 = gridDim.x, NG = gridDim.y;
    const int lin_ = blockIdx.x + NB3 * blockIdx.y;
    const int kb = GROUP_REMAP ? lin_ / NG : (int)blockIdx.x, g = GROUP_REMAP ? lin_ % NG : (int)blockIdx.y;
    const bool gating = w >= 8; // wave-uniform role
    const int lx = l >> 5, lu = (l >> 2) & 7, li = l & 3; // MFMA role: lane = 32x + 4u + i
    const int mcol = 8 * g + 4 * lx + li, mcolc = mcol < B ? mcol : B - 1;
    const int pair = tid & 127; // gating role: pair = column*16 + unit
    const int gc = pair >> 4, gu = pair & 15;
    const int col = 8 * g + gc, colc = col < B ? col : B - 1;
    const int j = 16 * kb + gu;

    const __amdgpu_buffer_rsrc_t rH = make_rsrc(H, (size_t)S * N * B * sizeof(float));
    const __amdgpu_buffer_rsrc_t rHx = make_rsrc(Hx, (size_t)HX_RING * N * B * sizeof(float));
    unsigned *xcc_tab = cnt + (size_t)g * CNT_SLOTS * CNT_STRIDE;
    if (tid == 0) {
        s_abort = 0;
        if (XCD_LOCAL) {
            __hip_atomic_store(xcc_tab + kb, (epoch << 4) | (__builtin_amdgcn_s_getreg(6164) & 15u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // visible before anything this workgroup publishes
        }
    }
    const int poll_sleep = poll_cfg & 255, poll_first = (poll_cfg >> 8) & 255;
    __syncthreads();

    if (!gating) {
        // ---------------- waves 0-7: h_{t-1} -> this wave's K-slice of U*h for 16 units x 8 columns ----------------
        float4 wq[2][NL][2][2];
#pragma unroll
        for (int ps = 0; ps < 2; ps++)
#pragma unroll
            for (int L = 0; L < NL; L++)
#pragma unroll
                for (int eh = 0; eh < 2; eh++)
#pragma unroll
                    for (int sh = 0; sh < 2; sh++)
                        wq[ps][L][eh][sh] = Ufwd4[(((((((size_t)kb * 8 + w) * 2 + ps) * NL + L) * 2 + eh) * 2 + sh) * 64) + l];
        for (int t = 1; t < S; t++) {
            float4 b[NL];
            FSTAMP(3, 8)
            int polls = 0;
            if (t == 1) {
                const int off = (int)((((size_t)mcolc) * N + Kw * w + 4 * lu) * sizeof(float));
#pragma unroll
                for (int i = 0; i < NL; i++) b[i] = ld_sc1(rH, off + 128 * i);
            } else {
                const int slot = (t - 1 + ring_base) & (HX_RING - 1);
                const int off = (int)((((size_t)slot * B + mcolc) * N + Kw * w + 4 * lu) * sizeof(float));
                for (int i = 0; i < poll_first; i++) __builtin_amdgcn_s_sleep(1);
                bool ok = false;
                // HINT.  The polls themselves load the L2 (three in flight per wave instead of one made a step 40 % longer), so
                // the wave first polls ONE 16-byte piece of one of its producers (lane 0 only: a single 64-byte request) and
                // fetches its K-slice only once that has flipped: 265.8 -> 253.6 us against polling the slice itself.  The
                // slice is still checked word by word below and re-fetched while any producer is later than the hinted one.
                {
                    int hcol = 8 * g + 3;
                    hcol = hcol < B ? hcol : B - 1;
                    const int hoff = (int)((((size_t)slot * B + hcol) * N + Kw * w + 12) * sizeof(float)); // units Kw*w+12 .. +15
                    for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                        float4 hv = {0.f, 0.f, 0.f, 0.f};
                        if (l == 0) hv = ld_sc1(rHx, hoff);
                        if (__all(hx_ready(hv))) break;
                        if ((spins & 255) == 255 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        for (int i = 0; i < poll_sleep; i++) __builtin_amdgcn_s_sleep(1);
                    }
                }
                for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                    bool good = true;
                    FSTAMP(3, 13) // issue time of the latest poll: at exit, of the one that succeeded
#pragma unroll
                    for (int i = 0; i < NL; i++) {
                        b[i] = ld_sc1(rHx, off + 128 * i); // (nt and sc0|sc1 loads measured the same: 265-266 us)
                        good = good && hx_ready(b[i]);
                    }
                    if (STAMP) polls = spins + 1;
                    if (__all(good)) {
                        ok = true;
                        break;
                    }
                    if ((spins & 255) == 255 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    for (int i = 0; i < poll_sleep; i++) __builtin_amdgcn_s_sleep(1); // 64 cycles each
                }
                if (!ok && l == 0) {
                    __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                }
            }
            FSTAMP(3, 9) FSTAMP_VAL(3, 12, polls)
            // four independent accumulation chains: [pass][slot parity]
            f32x4 c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
#define F4_HALF(av, q0, q1, s0, blgp)                                                   \
    c00 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q0.x, c00, 3, s0 + 0, blgp);          \
    c10 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q1.x, c10, 3, s0 + 0, blgp);          \
    c01 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q0.y, c01, 3, s0 + 1, blgp);          \
    c11 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q1.y, c11, 3, s0 + 1, blgp);          \
    c00 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q0.z, c00, 3, s0 + 2, blgp);          \
    c10 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q1.z, c10, 3, s0 + 2, blgp);          \
    c01 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q0.w, c01, 3, s0 + 3, blgp);          \
    c11 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, q1.w, c11, 3, s0 + 3, blgp);
#define F4_STEP(av, L, eh, blgp)                             \
    F4_HALF(av, wq[0][L][eh][0], wq[1][L][eh][0], 0, blgp)   \
    F4_HALF(av, wq[0][L][eh][1], wq[1][L][eh][1], 4, blgp)
#pragma unroll
            for (int i = 0; i < NL; i++) {
                F4_STEP(b[i].x, i, 0, 1)
                F4_STEP(b[i].y, i, 0, 2)
                F4_STEP(b[i].z, i, 1, 1)
                F4_STEP(b[i].w, i, 1, 2)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef F4_STEP
#undef F4_HALF
            FSTAMP(3, 10)
            // lane (x, u, gate li), register r of pass ps: gate li of unit 8*ps + u for column 4x + r
            float *rp = red[t & 1] + w * 512;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                rp[((4 * lx + r) * 16 + lu) * 4 + li] = c00[r] + c01[r];
                rp[((4 * lx + r) * 16 + 8 + lu) * 4 + li] = c10[r] + c11[r];
            }
            __syncthreads();
            if (s_abort) return;
            FSTAMP(3, 11)
        }
    } else {
        // ---------------- waves 8, 9: fold the K-slices, gates, cell, publish h_t; one (unit, column) pair per lane -------
        // These waves sit at the workgroup barrier while the product runs, so the fold starts the moment the last partial
        // sum is in LDS.  (With the gating done by two of the product waves, those two were the last to every barrier:
        // between publishing h_t and their next MFMA they alone issued the off-chain stores, the W gather and the poll;
        // the other six waited 1 300-1 500 cycles a step at the barrier -- stamps of that form, DESIGN.md.)
        float bs[4], cprev, wx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = bias[gt * N + j];
        cprev = C[(size_t)colc * N + j];
        int xnext = xi[1 * B + colc]; // input byte of the NEXT step's column, fetched a step ahead of the W gather
        bool local_pub = false;
        for (int t = 1; t < S; t++) {
            // W column of this step's input byte (R/lstm.cc:176 with a one-hot x): in flight while the product runs
#pragma unroll
            for (int gt = 0; gt < 4; gt++) wx[gt] = 0.f;
            if (xnext >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = W[(size_t)xnext * G4 + gt * N + j];
            }
            if (t + 1 < S) xnext = xi[(t + 1) * B + colc];
            FSTAMP(8, 0)
            __syncthreads();
            if (s_abort) return;
            FSTAMP(8, 1)
            if (XCD_LOCAL && t == 2) { // every workgroup of the group has published h_1, hence its XCC id before it
                unsigned mine = 0;
                bool same = true;
                if (l < NB3) {
                    mine = __hip_atomic_load(xcc_tab + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    same = (mine >> 4) == epoch;
                }
                const unsigned first = __builtin_amdgcn_readfirstlane(mine);
                if (l < NB3) same = same && mine == first;
                local_pub = (XCD_FORCE_LOCAL || __all(same)) && NB3 <= 64;
            }
            const float *rp = red[t & 1] + pair * 4;
            float4 uh = *reinterpret_cast<const float4 *>(rp);
#pragma unroll
            for (int ww = 1; ww < 8; ww++) {
                const float4 v = *reinterpret_cast<const float4 *>(rp + ww * 512);
                uh.x += v.x;
                uh.y += v.y;
                uh.z += v.z;
                uh.w += v.w;
            }
            const float pre0 = (wx[0] + uh.x) + bs[0], pre1 = (wx[1] + uh.y) + bs[1]; // R/lstm.cc:176
            const float pre2 = (wx[2] + uh.z) + bs[2], pre3 = (wx[3] + uh.w) + bs[3];
            const float ig = p_sigm<FAST>(pre0), og = p_sigm<FAST>(pre1), fg = p_sigm<FAST>(pre2); // :179
            const float ug_ = p_tanh<FAST>(pre3);                                                 // :182
            const float cv = p_tanh<FAST>(ig * ug_ + fg * cprev);                                 // :185-189
            const float hv = og * cv;                                                             // :192
            cprev = cv;
            // four consecutive units sit in the four lanes of a quad: gather them for one 16-byte store
            float4 h4;
            h4.x = dpp_f<0x00>(hv);
            h4.y = dpp_f<0x55>(hv);
            h4.z = dpp_f<0xAA>(hv);
            h4.w = dpp_f<0xFF>(hv);
            FSTAMP(8, 2)
            // the reset this wave issued a step ago (and every older store) has completed before h_t can be seen
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FSTAMP(8, 3)
            if ((gu & 3) == 0 && col < B) {
                const float4 hp = {hx_canon(h4.x), hx_canon(h4.y), hx_canon(h4.z), hx_canon(h4.w)};
                const float4 sent = {__uint_as_float(HX_SENT), __uint_as_float(HX_SENT), __uint_as_float(HX_SENT),
                                     __uint_as_float(HX_SENT)};
                const size_t e_pub = ((size_t)((t + ring_base) & (HX_RING - 1)) * B + col) * N + j;
                const size_t e_rst = ((size_t)((t + 2 + ring_base) & (HX_RING - 1)) * B + col) * N + j;
                if (XCD_LOCAL && local_pub) {
                    *reinterpret_cast<float4 *>(Hx + e_pub) = hp;
                    *reinterpret_cast<float4 *>(Hx + e_rst) = sent;
                } else {
                    st_sc1(hp, rHx, (int)(e_pub * sizeof(float)));
                    st_sc1(sent, rHx, (int)(e_rst * sizeof(float)));
                }
                *reinterpret_cast<float4 *>(H + ((size_t)t * B + col) * N + j) = h4;
            }
            if (col < B) {
                float *gcp = G + ((size_t)t * B + col) * G4 + j;
                gcp[0] = ig;
                gcp[N] = og;
                gcp[2 * N] = fg;
                gcp[3 * N] = ug_;
                C[((size_t)t * B + col) * N + j] = cv;
            }
            FSTAMP(8, 4)
        }
    }
}
#undef FSTAMP
#undef FSTAMP_VAL

// ------------------------------------------------------------------------------------------------
// forward recurrence, two half-groups per workgroup (N = 512): the grid, ring and gating of k_fwd_persistent4, but the
// eight columns of a workgroup are TWO independent recurrences of four columns, A and B, which the workgroup advances
// alternately.  A step of one recurrence is a chain  publish -> L2 -> poll hit -> MFMA -> fold -> gates -> publish  whose
// latencies (the hand-off alone: 2 700 of 6 300 cycles) the matrix pipe sits out; with two recurrences in one workgroup
// the product waves run B's MFMAs while A's h_t is being gated, published and fetched, and the other way round.  Each
// chain's period is then  hand-off + gates + HALF the MFMAs, and both chains complete a step per period.  (Two
// workgroups per CU on different 4-column groups would be the hardware's version of this; measured slower -- twice the
// polling waves, and which workgroups share a CU is the dispatcher's choice.)
//
// Product: v_mfma_f32_4x4x1, block = unit (lane = 4*unit + j), CBSZ = 4 / ABID = ab: all sixteen unit-blocks read the four
// columns of h[k] from block ab of the loaded register -- ONE 16-byte load per lane (64 values of k x 4 columns) feeds the
// wave's 64 instructions of a half-step:
//   D[i][j] (+)= h[k = Kw*w + 4*ab + r][column i of the half] * U[gate j of unit 16*kb + unit][k]       (r = register)
// Weights Ufwd5 (k_pack_U, ufwd5_index), one copy for both halves.  Wave 8 gates half A (64 pairs), wave 9 half B.
// The loop has NO workgroup barrier: the product waves count their partial-sum images into an LDS word per half, the
// gating wave of the half waits for the count, and the images are double-buffered by step parity (reuse is ordered by the
// ring itself: a product wave writing step t+2 has fetched h_{t+1}, published after the gating wave read image t+1).
// The waves of the two halves therefore never wait for each other, only for data.
// ------------------------------------------------------------------------------------------------
#define FSTAMP(wave, k)                                                                                        \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = __builtin_amdgcn_s_memtime();
template <int N_, bool FAST, bool STAMP = false>
__global__ __launch_bounds__(FWD4_THREADS) void k_fwd_persistent6(const float4 *__restrict__ Ufwd5, const float *__restrict__ W,
                                                                  const float *__restrict__ bias, float *H, float *__restrict__ C,
                                                                  float *__restrict__ G, const int32_t *__restrict__ xi,
                                                                  float *Hx, unsigned *cnt, unsigned *abortp, unsigned epoch,
                                                                  int ring_base, int S, int B, int poll_cfg,
                                                                  unsigned long long *stamps = nullptr) {
    constexpr int N = N_, G4 = 4 * N, Kw = N / 8, RS = 68; // RS: padded row of the partial-sum image (bank spread)
    constexpr int NAB = Kw / 4; // 16-byte pieces (= lane blocks that carry data) in a wave's K-slice of a half: 16 or 8
    static_assert(N == 512 || N == 256, "one 16-byte load per lane covers a wave's K-slice of a half (N = 256: half the lanes)");
    __shared__ __attribute__((aligned(16))) float red[2][2][8 * 4 * RS]; // [half][step parity][wave][column][4*unit + gate]
    __shared__ int s_abort;
    // Per half and STEP PARITY: partial-sum images written so far, summed over the product waves (8 per step of that parity).
    // One count per half would not do: a product wave whose K-slice is fed by other workgroups can be a step ahead of a
    // wave of its own workgroup that is still waiting for a slow load (two steps ahead it cannot be: that needs this
    // workgroup's h of the step in between), and its early count would stand in for the late wave's missing one.
    __shared__ unsigned s_done[2][2];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    // poll_cfg bits 16-19: pinned launch (fewer than 8 groups): gridDim.y = 8, workgroup i runs on XCD i % 8, group g is the
    // workgroups of XCD g, the workgroups of the other XCDs leave at once
    const int pin_ng = (poll_cfg >> 16) & 15;
    const int g0 = (poll_cfg >> 20) & 255; // first column group of this launch (a wide batch runs as launches over column ranges)
    // bit 28: ONE half per workgroup (4-column groups; half B's gating wave leaves): where the chip has room for twice the
    // workgroups, a half's data never waits behind the other half's matrix phase (k_fwd_halves_bf16 has the measurements)
    const int GC = (poll_cfg >> 28) & 1 ? 4 : 8;
    const int NB3 = gridDim.x, NG = pin_ng ? pin_ng : (int)gridDim.y;
    const int lin_ = blockIdx.x + NB3 * blockIdx.y;
    const int kb = pin_ng ? lin_ >> 3 : GROUP_REMAP ? lin_ / NG : (int)blockIdx.x;
    const int g = pin_ng ? lin_ & 7 : GROUP_REMAP ? lin_ % NG : (int)blockIdx.y;
    if (pin_ng && g >= NG) return;
    const __amdgpu_buffer_rsrc_t rH = make_rsrc(H, (size_t)S * N * B * sizeof(float));
    const __amdgpu_buffer_rsrc_t rHx = make_rsrc(Hx, (size_t)HX_RING * N * B * sizeof(float));
    unsigned *xcc_tab = cnt + (size_t)g * CNT_SLOTS * CNT_STRIDE;
    if (tid == 0) {
        s_abort = 0;
        s_done[0][0] = s_done[0][1] = s_done[1][0] = s_done[1][1] = 0;
        if (XCD_LOCAL) {
            __hip_atomic_store(xcc_tab + kb, (epoch << 4) | (__builtin_amdgcn_s_getreg(6164) & 15u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // visible before anything this workgroup publishes
        }
    }
    const int poll_sleep = poll_cfg & 255;
    __syncthreads();

    if (w < 8) {
        // ---------------- waves 0-7: the product, half A then half B ----------------
        const int lb = l >> 2, li = l & 3; // lane = 4*block + i
        const bool ld_lane = lb < NAB; // N = 256: blocks 8-15 have nothing to fetch (their registers read as complete)
        float4 wq[NAB];
#pragma unroll
        for (int ab = 0; ab < NAB; ab++) wq[ab] = Ufwd5[(((size_t)kb * 8 + w) * NAB + ab) * 64 + l];
        int colv[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const int c = GC * (g + g0) + 4 * hf + li;
            colv[hf] = c < B ? c : B - 1;
        }
        // h_0 of both halves (plain window state in H); later fragments come from the ring, requested a half-step ahead
        float4 bvq[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
            bvq[hf] = ld_lane ? ld_sc1(rH, (int)((((size_t)colv[hf]) * N + Kw * w + 4 * lb) * sizeof(float))) : float4{0.f, 0.f, 0.f, 0.f};
        for (int t = 1; t < S; t++) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                if (hf == 1 && GC == 4) continue;
                if (hf == 0) { FSTAMP(3, 8) }
                float4 bv = bvq[hf];
                if (t > 1 && !__all(hx_ready(bv))) {
                    // The fragment requested ahead came back incomplete: fetch the K-slice again until no word of it is the
                    // sentinel.  No one-request hint ahead of it as in k_fwd_persistent4: here the wave has the other half
                    // to work on, arrives late at most polls, and the hint's extra round trip costs more than it saves
                    // (measured 243 us with the hint, 229-232 without).
                    const int slot = (t - 1 + ring_base) & (HX_RING - 1);
                    const int off = (int)((((size_t)slot * B + colv[hf]) * N + Kw * w + 4 * lb) * sizeof(float));
                    bool ok = false;
                    for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                        bv = ld_lane ? ld_sc1(rHx, off) : float4{0.f, 0.f, 0.f, 0.f};
                        if (__all(hx_ready(bv))) {
                            ok = true;
                            break;
                        }
                        if ((spins & 255) == 255 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        for (int i = 0; i < poll_sleep; i++) __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok) {
                        if (l == 0) {
                            __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        return;
                    }
                }
                if (hf == 0) { FSTAMP(3, 9) } else { FSTAMP(3, 5) }
                // the OTHER half's next fragment -- (t, B) after A's product, (t+1, A) after B's -- is requested now and
                // looked at after this half's barrier: its round trip runs under the matrix instructions below
                const int nslot = (t - 1 + hf + ring_base) & (HX_RING - 1);
                const int noff = (int)((((size_t)nslot * B + colv[hf ^ 1]) * N + Kw * w + 4 * lb) * sizeof(float));
                const bool req = GC == 8 && (hf == 0 ? t > 1 : t + 1 < S);
                // four independent accumulation chains, one per register of the loaded fragment
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#define F6(ab)                                                              \
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv.x, wq[ab].x, c0, 4, ab, 0); \
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv.y, wq[ab].y, c1, 4, ab, 0); \
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv.z, wq[ab].z, c2, 4, ab, 0); \
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv.w, wq[ab].w, c3, 4, ab, 0);
                F6(0) F6(1) F6(2) F6(3) F6(4) F6(5) F6(6) F6(7)
                if constexpr (NAB == 16) { F6(NAB - 8) F6(NAB - 7) F6(NAB - 6) F6(NAB - 5) F6(NAB - 4) F6(NAB - 3) F6(NAB - 2) F6(NAB - 1) }
#undef F6
                __builtin_amdgcn_sched_barrier(0);
                if (req) bvq[hf ^ 1] = ld_lane ? ld_sc1(rHx, noff) : float4{0.f, 0.f, 0.f, 0.f}; // behind the last matrix instruction (ahead of them: 236-240 us)
                if (GC == 4) { // one half only: its next fragment cannot have been published yet, the poll above fetches it
                    const float sv = __uint_as_float(HX_SENT);
                    bvq[0] = ld_lane ? float4{sv, sv, sv, sv} : float4{0.f, 0.f, 0.f, 0.f};
                }
                __builtin_amdgcn_sched_barrier(0);
                if (hf == 0) { FSTAMP(3, 10) } else { FSTAMP(3, 6) }
                // lane (unit, gate j), register i = column i of the half: one row of the image per (wave, column)
                // No workgroup barrier anywhere in the loop: the gating wave of the half counts the images in (LDS operations
                // of a wave execute in order, so the count lands behind the sums), and an image is overwritten two steps
                // later, by which time this wave has fetched an h that the gating wave published after reading it.
                float *rp = red[hf][t & 1];
#pragma unroll
                for (int r = 0; r < 4; r++) rp[(w * 4 + r) * RS + l] = (c0[r] + c1[r]) + (c2[r] + c3[r]);
                asm volatile("" ::: "memory");
                if (l == 0) __hip_atomic_fetch_add(&s_done[hf][t & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (hf == 0) { FSTAMP(3, 11) } else { FSTAMP(3, 7) }
            }
        }
    } else {
        // ---------------- wave 8: gates of half A; wave 9: gates of half B; lane = column*16 + unit ----------------
        const int hf = w - 8;
        if (hf == 1 && GC == 4) return;
        __builtin_amdgcn_s_setprio(3); // the gates are on the chain; the other half's product, issuing beside them, is not
        const int gc = l >> 4, gu = l & 15;
        const int col = GC * (g + g0) + 4 * hf + gc, colc = col < B ? col : B - 1;
        const int j = 16 * kb + gu;
        float bs[4], cprev, wx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = bias[gt * N + j];
        cprev = C[(size_t)colc * N + j];
        int xnext = xi[1 * B + colc]; // input byte of the NEXT step's column, fetched a step ahead of the W gather
        bool local_pub = false;
        const float *rp0 = red[hf][0] + gc * RS + 4 * gu;
        for (int t = 1; t < S; t++) {
            // W column of this step's input byte (R/lstm.cc:176 with a one-hot x): in flight while the product runs
#pragma unroll
            for (int gt = 0; gt < 4; gt++) wx[gt] = 0.f;
            if (xnext >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = W[(size_t)xnext * G4 + gt * N + j];
            }
            if (t + 1 < S) xnext = xi[(t + 1) * B + colc];
            FSTAMP(8, 0)
            {   // all eight partial-sum images of this half and step are in LDS
                bool in = false;
                for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                    // steps 1..t of the parity of t: (t + 1) / 2 of them
                    in = __hip_atomic_load(&s_done[hf][t & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 8u * (unsigned)((t + 1) / 2);
                    if (in || __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!in) {
                    if (l == 0) {
                        __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    return;
                }
                asm volatile("" ::: "memory");
            }
            const float *rp = rp0 + (t & 1) * (8 * 4 * RS);
            FSTAMP(8, 1)
            if (XCD_LOCAL && t == 2) { // every workgroup of the group has published h_1, hence its XCC id before it
                unsigned mine = 0;
                bool same = true;
                if (l < NB3) {
                    mine = __hip_atomic_load(xcc_tab + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    same = (mine >> 4) == epoch;
                }
                const unsigned first = __builtin_amdgcn_readfirstlane(mine);
                if (l < NB3) same = same && mine == first;
                local_pub = (XCD_FORCE_LOCAL || __all(same)) && NB3 <= 64;
            }
            // the four gates of (unit gu, column gc) sit side by side in every wave's row gc: one 16-byte read each
            float4 uh = *reinterpret_cast<const float4 *>(rp);
#pragma unroll
            for (int ww = 1; ww < 8; ww++) {
                const float4 v = *reinterpret_cast<const float4 *>(rp + ww * 4 * RS);
                uh.x += v.x;
                uh.y += v.y;
                uh.z += v.z;
                uh.w += v.w;
            }
            const float pre0 = (wx[0] + uh.x) + bs[0], pre1 = (wx[1] + uh.y) + bs[1]; // R/lstm.cc:176
            const float pre2 = (wx[2] + uh.z) + bs[2], pre3 = (wx[3] + uh.w) + bs[3];
            const float ig = p_sigm<FAST>(pre0), og = p_sigm<FAST>(pre1), fg = p_sigm<FAST>(pre2); // :179
            const float ug_ = p_tanh<FAST>(pre3);                                                 // :182
            const float cv = p_tanh<FAST>(ig * ug_ + fg * cprev);                                 // :185-189
            const float hv = og * cv;                                                             // :192
            cprev = cv;
            float4 h4;
            h4.x = dpp_f<0x00>(hv);
            h4.y = dpp_f<0x55>(hv);
            h4.z = dpp_f<0xAA>(hv);
            h4.w = dpp_f<0xFF>(hv);
            FSTAMP(8, 2)
            // the reset this wave issued a step ago (and every older store) has completed before h_t can be seen
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FSTAMP(8, 3)
            if ((gu & 3) == 0 && col < B) {
                const float4 hp = {hx_canon(h4.x), hx_canon(h4.y), hx_canon(h4.z), hx_canon(h4.w)};
                const float4 sent = {__uint_as_float(HX_SENT), __uint_as_float(HX_SENT), __uint_as_float(HX_SENT),
                                     __uint_as_float(HX_SENT)};
                const size_t e_pub = ((size_t)((t + ring_base) & (HX_RING - 1)) * B + col) * N + j;
                const size_t e_rst = ((size_t)((t + 2 + ring_base) & (HX_RING - 1)) * B + col) * N + j;
                if (XCD_LOCAL && local_pub) {
                    *reinterpret_cast<float4 *>(Hx + e_pub) = hp;
                    *reinterpret_cast<float4 *>(Hx + e_rst) = sent;
                } else {
                    st_sc1(hp, rHx, (int)(e_pub * sizeof(float)));
                    st_sc1(sent, rHx, (int)(e_rst * sizeof(float)));
                }
                *reinterpret_cast<float4 *>(H + ((size_t)t * B + col) * N + j) = h4;
            }
            if (col < B) {
                float *gcp = G + ((size_t)t * B + col) * G4 + j;
                gcp[0] = ig;
                gcp[N] = og;
                gcp[2 * N] = fg;
                gcp[3 * N] = ug_;
                C[((size_t)t * B + col) * N + j] = cv;
            }
            FSTAMP(8, 4)
        }
    }
}
#undef FSTAMP

// ------------------------------------------------------------------------------------------------
// bf16 recurrence (LSTM_HIP_BF16_RECURRENCE, BASELINE configs[4]): the operands of U*h_prev are bfloat16
// (round-to-nearest-even of the fp32 master weights and of the published h), the accumulation is fp32,
// everything else is the fp32 kernel above.  MFMA 16x16x32 bf16: A[row=l&15][k=8*(l>>4)+j],
// B[k=8*(l>>4)+j][col=l&15], j = 0..7; one lane-fragment is 16 bytes.  N = 128*NKS.
//   Ufwd16[p][ks][l] = { bf16(U[(l&3)*N + 4p + ((l&15)>>2)][32*ks + 8*(l>>4) + j]) }
//   Ubwd16[kb][rs][l] = { bf16(U[32*rs + 8*(l>>4) + j][16*kb + (l&15)]) }
// h_t is published twice: fp32 (plain stores, for the time-batched products) and bf16 (sc1, the hand-off).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) |
           ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
__device__ __forceinline__ u32x4 pack_bf16x8(const float4 &a, const float4 &b) {
    u32x4 v;
    v.x = pack_bf16x2(a.x, a.y);
    v.y = pack_bf16x2(a.z, a.w);
    v.z = pack_bf16x2(b.x, b.y);
    v.w = pack_bf16x2(b.z, b.w);
    return v;
}

// Ubwd6b[kb][w][s][ab][l] = bf16 x 4 { U[gate*N + UW*kb + unit][output], k = gate*UW + unit = 4*ab + e, e = 0..3 }, output =
// (w*NS + s)*64 + l, ab = 0 .. UW-1: the scatter form's image (k_bwd_scatter_bf16), 8 bytes per lane and (s, ab)
__global__ __launch_bounds__(256) void k_pack_U6_bf16(const float *__restrict__ U, uint2 *__restrict__ Ubwd6b, int N, int UW) {
    const int G4 = 4 * N, NS = N >= 512 ? N / 512 : 1, NPW = N / (64 * NS);
    const size_t total = (size_t)N * N; // 8-byte elements: 4N*N values / 4
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63), ab = (int)((e >> 6) % UW);
        const size_t r = (e >> 6) / UW;
        const int sx = (int)(r % NS), w = (int)((r / NS) % NPW), kb = (int)(r / ((size_t)NS * NPW));
        const int out = (w * NS + sx) * 64 + l;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int kk = 4 * ab + q, row = (kk / UW) * N + UW * kb + (kk % UW);
            v[q] = U[(size_t)out * G4 + row];
        }
        Ubwd6b[e] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}
int bwd_scatter_bf16_units(int N);
void pack_U6_bf16(const float *U, void *Ubwd6b, int N, hipStream_t st) {
    const size_t n = (size_t)N * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_U6_bf16, dim3(blocks), dim3(256), 0, st, U, reinterpret_cast<uint2 *>(Ubwd6b), N, bwd_scatter_bf16_units(N));
}
__global__ __launch_bounds__(256) void k_pack_U_bf16(const float *__restrict__ U, u32x4 *__restrict__ Ufwd16,
                                                     u32x4 *__restrict__ Ubwd16, int N) {
    const int G4 = 4 * N;
    const size_t n16 = (size_t)N * N / 2; // 16-byte fragments per image
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < 2 * n16; e += (size_t)gridDim.x * blockDim.x) {
        float v[8];
        if (e < n16) {
            const int l = (int)(e & 63);
            const size_t qq = e >> 6;
            const int ks = (int)(qq % (N / 32)), p = (int)(qq / (N / 32));
            const int row = (l & 3) * N + 4 * p + ((l & 15) >> 2), k = 32 * ks + 8 * (l >> 4);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = U[(size_t)(k + j) * G4 + row];
        } else {
            const size_t e2 = e - n16;
            const int l = (int)(e2 & 63);
            const size_t qq = e2 >> 6;
            const int rs = (int)(qq % (G4 / 32)), kb = (int)(qq / (G4 / 32));
            const float *src = U + (size_t)(16 * kb + (l & 15)) * G4 + 32 * rs + 8 * (l >> 4);
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = src[j];
        }
        u32x4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        if (e < n16) Ufwd16[e] = o;
        else Ubwd16[e - n16] = o;
    }
}
void pack_U_bf16(const float *U, void *Ufwd16, void *Ubwd16, int N, hipStream_t st) {
    const size_t n = (size_t)N * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_U_bf16, dim3(blocks), dim3(256), 0, st, U, reinterpret_cast<u32x4 *>(Ufwd16),
                       reinterpret_cast<u32x4 *>(Ubwd16), N);
}

template <int NKS, bool FAST>
__global__ __launch_bounds__(256) void k_fwd_persistent_bf16(const u32x4 *__restrict__ Ufwd16, const float *__restrict__ W,
                                                             const float *__restrict__ bias, float *__restrict__ H,
                                                             unsigned short *Hb, float *__restrict__ C,
                                                             float *__restrict__ G, const int32_t *__restrict__ xi,
                                                             unsigned *cnt, unsigned *abortp, unsigned epoch, int S,
                                                             int B) {
    constexpr int N = 128 * NKS, G4 = 4 * N;
    __shared__ float red[4 * 4 * 64];
    __shared__ int s_abort;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NB = gridDim.x, NG = gridDim.y;
    const int lin_ = blockIdx.x + NB * blockIdx.y;
    const int p = GROUP_REMAP ? lin_ / NG : (int)blockIdx.x, g = GROUP_REMAP ? lin_ % NG : (int)blockIdx.y;
    const int q = l >> 4, c = l & 15;
    const int col = 16 * g + c, colc = col < B ? col : B - 1;
    const int j = 4 * p + q;

    u32x4 a[NKS];
#pragma unroll
    for (int i = 0; i < NKS; i++) a[i] = Ufwd16[((size_t)p * (N / 32) + w * NKS + i) * 64 + l];
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (w == 0) {
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = bias[gt * N + j];
        cprev = C[(size_t)colc * N + j];
    }
    const __amdgpu_buffer_rsrc_t rHb = make_rsrc(Hb, (size_t)S * N * B * sizeof(unsigned short));
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();

    for (int t = 1; t < S; t++) {
        float wx[4] = {0.f, 0.f, 0.f, 0.f};
        if (w == 0) {
            const int x = xi[t * B + colc];
            if (x >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = W[(size_t)x * G4 + gt * N + j];
            }
            if (t > 1) {
                const unsigned *cp = cnt + (size_t)((t - 1) * NG + g) * CNT_SLOTS * CNT_STRIDE;
                if (!wait_arrivals<FWD_SH>(cp, NB, epoch, abortp, l) && l == 0) s_abort = 1;
            }
        }
        __syncthreads();
        if (s_abort) return;

        u32x4 b[NKS];
        if (t == 1) { // the carry column exists only in fp32 (written before the launch): round it here
            const float *hp = H + (size_t)colc * N + 32 * (w * NKS) + 8 * q;
#pragma unroll
            for (int i = 0; i < NKS; i++)
                b[i] = pack_bf16x8(*reinterpret_cast<const float4 *>(hp + 32 * i), *reinterpret_cast<const float4 *>(hp + 32 * i + 4));
        } else {
            const int off = (int)((((size_t)(t - 1) * B + colc) * N + 32 * (w * NKS) + 8 * q) * sizeof(unsigned short));
#pragma unroll
            for (int i = 0; i < NKS; i++) b[i] = __builtin_amdgcn_raw_buffer_load_b128(rHb, off + 64 * i, 0, 16);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKS; i++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[i]), acc,
                                                          0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) red[(w * 4 + r) * 64 + l] = acc[r];
        __syncthreads();

        if (w == 0) {
            float pre[4];
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                const float uh = ((red[(0 * 4 + gt) * 64 + l] + red[(1 * 4 + gt) * 64 + l]) + red[(2 * 4 + gt) * 64 + l]) +
                                 red[(3 * 4 + gt) * 64 + l];
                pre[gt] = (wx[gt] + uh) + bs[gt]; // R/lstm.cc:176
            }
            const float ig = p_sigm<FAST>(pre[0]), og = p_sigm<FAST>(pre[1]), fg = p_sigm<FAST>(pre[2]); // :179
            const float ug = p_tanh<FAST>(pre[3]);                                                        // :182
            const float cv = p_tanh<FAST>(ig * ug + fg * cprev);                                          // :185-189
            const float hv = og * cv;                                                                     // :192
            cprev = cv;
            float4 h4;
            h4.x = __shfl(hv, c, 64);
            h4.y = __shfl(hv, 16 + c, 64);
            h4.z = __shfl(hv, 32 + c, 64);
            h4.w = __shfl(hv, 48 + c, 64);
            if (q == 0 && col < B) { // the hand-off copy: 4 units as bf16 = one 8-byte sc1 store
                const unsigned long long pk = (unsigned long long)pack_bf16x2(h4.x, h4.y) |
                                              ((unsigned long long)pack_bf16x2(h4.z, h4.w) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(Hb + ((size_t)t * B + col) * N + 4 * p), pk,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t + 1 < S) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0)
                    __hip_atomic_fetch_add(cnt + ((size_t)(t * NG + g) * CNT_SLOTS + (p & (FWD_SH - 1))) * CNT_STRIDE, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
            if (col < B) {
                if (q == 0) *reinterpret_cast<float4 *>(H + ((size_t)t * B + col) * N + 4 * p) = h4;
                float *gc = G + ((size_t)t * B + col) * G4 + j;
                gc[0] = ig;
                gc[N] = og;
                gc[2 * N] = fg;
                gc[3 * N] = ug;
                C[((size_t)t * B + col) * N + j] = cv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 recurrence, second form (see k_fwd_persistent2): 8 units per workgroup, K over 8 waves, every loaded fragment
// of bf16(h_{t-1}) feeds two MFMA tiles.  N = 256*NKS2.
// ------------------------------------------------------------------------------------------------
template <int NKS2, bool FAST, int COLS = 16>
__global__ __launch_bounds__(512) void k_fwd_persistent2_bf16(const u32x4 *__restrict__ Ufwd16, const float *__restrict__ W,
                                                              const float *__restrict__ bias, float *__restrict__ H,
                                                              unsigned short *Hb, float *__restrict__ C,
                                                              float *__restrict__ G, const int32_t *__restrict__ xi,
                                                              unsigned *cnt, unsigned *abortp, unsigned epoch, int S,
                                                              int B) {
    constexpr int N = 256 * NKS2, G4 = 4 * N;
    __shared__ float red[8 * 2 * 4 * 64];
    __shared__ int s_abort;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NB2 = gridDim.x, NG = gridDim.y;
    const int lin_ = blockIdx.x + NB2 * blockIdx.y;
    const int p2 = GROUP_REMAP ? lin_ / NG : (int)blockIdx.x, g = GROUP_REMAP ? lin_ % NG : (int)blockIdx.y;
    const int q = l >> 4, c = l & 15;
    // COLS = 8: half the tile columns carry no data (lanes c >= 8 load nothing and store nothing) -- the matrix pipe is not
    // what bounds a step, and 8-column groups put twice as many CUs to work on half the h bytes each (small batches)
    const bool live = c < COLS;
    const int col = COLS * g + (c & (COLS - 1)), colc = col < B ? col : B - 1;
    const int p = 2 * p2 + (w & 1); // the row tile a gating wave (w < 2) finishes
    const int j = 4 * p + q;

    u32x4 a0[NKS2], a1[NKS2];
#pragma unroll
    for (int i = 0; i < NKS2; i++) {
        a0[i] = Ufwd16[((size_t)(2 * p2) * (N / 32) + w * NKS2 + i) * 64 + l];
        a1[i] = Ufwd16[((size_t)(2 * p2 + 1) * (N / 32) + w * NKS2 + i) * 64 + l];
    }
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (w < 2) {
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = bias[gt * N + j];
        cprev = C[(size_t)colc * N + j];
    }
    const __amdgpu_buffer_rsrc_t rHb = make_rsrc(Hb, (size_t)S * N * B * sizeof(unsigned short));
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();

    for (int t = 1; t < S; t++) {
        float wx[4] = {0.f, 0.f, 0.f, 0.f};
        if (w < 2) {
            const int x = xi[t * B + colc];
            if (x >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = W[(size_t)x * G4 + gt * N + j];
            }
        }
        if (w == 0 && t > 1) {
            const unsigned *cp = cnt + (size_t)((t - 1) * NG + g) * CNT_SLOTS * CNT_STRIDE;
            if (!wait_arrivals<FWD_SH>(cp, 2 * NB2, epoch, abortp, l) && l == 0) s_abort = 1;
        }
        __syncthreads();
        if (s_abort) return;

        u32x4 b[NKS2];
        if (t == 1) { // the carry column exists only in fp32 (written before the launch): round it here
            const float *hp = H + (size_t)colc * N + 32 * (w * NKS2) + 8 * q;
#pragma unroll
            for (int i = 0; i < NKS2; i++)
                b[i] = live ? pack_bf16x8(*reinterpret_cast<const float4 *>(hp + 32 * i), *reinterpret_cast<const float4 *>(hp + 32 * i + 4))
                            : u32x4{0u, 0u, 0u, 0u};
        } else {
            const int off = (int)((((size_t)(t - 1) * B + colc) * N + 32 * (w * NKS2) + 8 * q) * sizeof(unsigned short));
#pragma unroll
            for (int i = 0; i < NKS2; i++) b[i] = live ? __builtin_amdgcn_raw_buffer_load_b128(rHb, off + 64 * i, 0, 16) : u32x4{0u, 0u, 0u, 0u};
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NKS2; i++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a0[i]), __builtin_bit_cast(bf16x8, b[i]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1[i]), __builtin_bit_cast(bf16x8, b[i]), acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            red[((w * 2 + 0) * 4 + r) * 64 + l] = acc0[r];
            red[((w * 2 + 1) * 4 + r) * 64 + l] = acc1[r];
        }
        __syncthreads();

        if (w < 2) {
            float pre[4];
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                float uh = red[((0 * 2 + w) * 4 + gt) * 64 + l];
#pragma unroll
                for (int ww = 1; ww < 8; ww++) uh += red[((ww * 2 + w) * 4 + gt) * 64 + l];
                pre[gt] = (wx[gt] + uh) + bs[gt]; // R/lstm.cc:176
            }
            const float ig = p_sigm<FAST>(pre[0]), og = p_sigm<FAST>(pre[1]), fg = p_sigm<FAST>(pre[2]); // :179
            const float ug = p_tanh<FAST>(pre[3]);                                                        // :182
            const float cv = p_tanh<FAST>(ig * ug + fg * cprev);                                          // :185-189
            const float hv = og * cv;                                                                     // :192
            cprev = cv;
            float4 h4;
            h4.x = __shfl(hv, c, 64);
            h4.y = __shfl(hv, 16 + c, 64);
            h4.z = __shfl(hv, 32 + c, 64);
            h4.w = __shfl(hv, 48 + c, 64);
            if (q == 0 && live && col < B) { // the hand-off copy: 4 units as bf16 = one 8-byte sc1 store
                const unsigned long long pk = (unsigned long long)pack_bf16x2(h4.x, h4.y) |
                                              ((unsigned long long)pack_bf16x2(h4.z, h4.w) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(Hb + ((size_t)t * B + col) * N + 4 * p), pk,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t + 1 < S) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0)
                    __hip_atomic_fetch_add(cnt + ((size_t)(t * NG + g) * CNT_SLOTS + (p & (FWD_SH - 1))) * CNT_STRIDE, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
            if (live && col < B) {
                if (q == 0) *reinterpret_cast<float4 *>(H + ((size_t)t * B + col) * N + 4 * p) = h4;
                float *gc = G + ((size_t)t * B + col) * G4 + j;
                gc[0] = ig;
                gc[N] = og;
                gc[2 * N] = fg;
                gc[3 * N] = ug;
                C[((size_t)t * B + col) * N + j] = cv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward recurrence, two half-groups per workgroup, bf16 operands (LSTM_HIP_BF16_RECURRENCE; N = 256 / 512 / 1024, 8-column
// groups): k_fwd_persistent6 with the product on v_mfma_f32_4x4x4_16b_bf16.  UW units to a workgroup: 16, or 32 at hidden
// 1024 so that a group is 32 workgroups and fits one XCD (pinned launch as in k_bwd_scatter_bf16): every hand-off of the
// group then stays inside that XCD's L2.  h_t travels as bfloat16 (RNE, what the product consumes) through a 4-slot sentinel
// ring Hxb[slot][column][N]: a lane's A operand is one 8-byte load (4 values of k of its column).
//   D[column i][gate j of unit 16*set + u] += sum_e h[k = Kw*w + 64*r + 4*ab + e][column i] * U[gate j of unit][k]
//   lane = 4*u + j on the weights' side, 4*ab + i on h's side; CBSZ = 4 / ABID = ab
// Waves 0-7: the product (K split eight ways, half A then half B); waves 8..: gates / cell / publish, UW/16 waves per half.
// (v_mfma_f32_4x4x4_16b_bf16 issues in 16 cycles -- stamps: 16 of them 232-264 cycles, 64 of them 1 450-1 800 with the
// SIMD's other product wave beside -- twice the rate of the fp32 4x4x1 per value of k.  v_mfma_f32_16x16x32_bf16 with the four
// columns in a quarter of its width was tried in its place: 8 instead of 16 instructions at hidden 512, but 500 instead of 250
// cycles for the phase, forward 147 -> 171 us; at hidden 1024 its operands no longer fit the 168 registers.  Hidden 1024 as ONE
// 8-column recurrence per workgroup on 16x16x32 -- half its width used, 32 instructions a step and wave -- measured 262 us against
// this form's 252: what the two halves hide behind each other is worth more than the matrix time saved.  Without any matrix
// instruction the hidden-1024 launch takes 189 us, the hidden-512 one 147.)
// ------------------------------------------------------------------------------------------------
typedef short fbf16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int I, int E, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
template <int N_, int UW> struct FwdhbShape {
    static constexpr int N = N_, NB = N / UW, Kw = N / 8, NRK = Kw >= 64 ? Kw / 64 : 1, NAB = Kw >= 64 ? 16 : Kw / 4;
    static constexpr int NSET = UW / 16, NEH = UW / 16, THREADS = (8 + 2 * NEH) * 64;
    static constexpr int ROW = 64 * NSET, RS = ROW + 4; // a (wave, column) row of the partial-sum image, padded
    static constexpr size_t LDS = sizeof(float) * 2 * 2 * 8 * 4 * RS;
};
struct FwdhbArgs {
    const uint2 *Ufwd6b;
    const float *W, *bias;
    float *H;
    unsigned short *Hb;
    float *C, *G;
    const int32_t *xi;
    unsigned *Hxb; // ring, as 32-bit words (two bf16 each)
    unsigned *cnt, *abortp;
    unsigned epoch;
    int ring_base, S, B, NG, pinned, col0, gcols; // columns col0 .. col0 + gcols*NG - 1 of the B (a launch takes as many groups as are co-resident)
    unsigned long long *stamps;
};
__device__ __forceinline__ bool hxb_ready(const u32x2_t &v) { return v.x != HX_SENT && v.y != HX_SENT; }
__device__ __forceinline__ unsigned hxb_canon(unsigned v) { return v == HX_SENT ? 0x7FC07FC0u : v; }
template <int N_, int UW, bool FAST, bool STAMP = false>
__global__ __launch_bounds__((FwdhbShape<N_, UW>::THREADS)) void k_fwd_halves_bf16(const FwdhbArgs p) {
    using Sh = FwdhbShape<N_, UW>;
    constexpr int N = N_, G4 = 4 * N, NB = Sh::NB, Kw = Sh::Kw, NRK = Sh::NRK, NAB = Sh::NAB, NSET = Sh::NSET, NEH = Sh::NEH, RS = Sh::RS;
    static_assert((N == 256 || N == 512 || N == 1024) && (UW == 16 || UW == 32), "bf16 two-half forward form: shapes");
    extern __shared__ __attribute__((aligned(16))) float red_[]; // [half][step parity][wave][column][RS]: [4*unit + gate]
    __shared__ int s_abort;
    __shared__ unsigned s_done[2][2];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int NG = p.NG, GC = p.gcols; // GC = 8: two halves of four columns per workgroup; 4: one (half B's waves leave)
    int kb, g;
    if (p.pinned) { // 8 * NB workgroups launched, workgroup i on XCD i % 8: group g = the workgroups of XCD g
        g = (int)blockIdx.x & 7, kb = (int)blockIdx.x >> 3;
        if (g >= NG) return;
    } else {
        kb = (int)blockIdx.x / NG, g = (int)blockIdx.x % NG;
    }
    const int S = p.S, B = p.B, ring_base = p.ring_base;
    // diagnostics (LSTM_HIP_DEBUG_STAMPS): lane 0 of waves 3 and 8 of workgroups (0, 0) and (NB/2, 0); slots as in k_fwd_persistent6
    unsigned long long *stq = STAMP && g == 0 && (kb == 0 || kb == NB / 2) && l == 0 && (w == 3 || w == 8) ? p.stamps + (size_t)(kb ? 1 : 0) * S * 16 : nullptr;
#define HSTAMPQ(k) if (STAMP && stq) stq[(size_t)t * 16 + (k)] = __builtin_amdgcn_s_memtime();
    const __amdgpu_buffer_rsrc_t rH = make_rsrc(p.H, (size_t)S * N * B * sizeof(float));
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(p.Hxb, (size_t)HX_RING * N * B * sizeof(unsigned short));
    unsigned *xcc_tab = p.cnt + (size_t)g * CNT_SLOTS * CNT_STRIDE;
    if (tid == 0) {
        s_abort = 0;
        s_done[0][0] = s_done[0][1] = s_done[1][0] = s_done[1][1] = 0;
        if (XCD_LOCAL) {
            __hip_atomic_store(xcc_tab + kb, (p.epoch << 4) | (__builtin_amdgcn_s_getreg(6164) & 15u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // visible before anything this workgroup publishes
        }
    }
    __syncthreads();
    auto give_up = [&]() {
        if (l == 0) {
            __hip_atomic_store(p.abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    if (w < 8) {
        // ---------------- waves 0-7: the product, half A then half B ----------------
        const int lb = l >> 2, li = l & 3; // lane = 4*block + i
        const bool ld_lane = lb < NAB;     // N = 256: blocks 8-15 have nothing to fetch (their registers read as complete)
        uint2 wq[NRK][NSET][NAB];
#pragma unroll
        for (int r = 0; r < NRK; r++)
#pragma unroll
            for (int sx = 0; sx < NSET; sx++)
#pragma unroll
                for (int ab = 0; ab < NAB; ab++)
                    wq[r][sx][ab] = p.Ufwd6b[(((((size_t)kb * 8 + w) * NRK + r) * NSET + sx) * NAB + ab) * 64 + l];
#pragma unroll
        for (int r = 0; r < NRK; r++)
#pragma unroll
            for (int sx = 0; sx < NSET; sx++)
#pragma unroll
                for (int ab = 0; ab < NAB; ab++) asm volatile("" ::"v"(wq[r][sx][ab].x), "v"(wq[r][sx][ab].y)); // complete before the loop
        int colv[2];
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const int c = p.col0 + GC * g + 4 * hf + li;
            colv[hf] = c < B ? c : B - 1;
        }
        // h_0 of both halves (plain fp32 window state in H, rounded here); later fragments come from the ring
        u32x2_t bvq[2][NRK];
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int r = 0; r < NRK; r++) {
                bvq[hf][r] = u32x2_t{0u, 0u};
                if (ld_lane) {
                    const float4 f = ld_sc1(rH, (int)((((size_t)colv[hf]) * N + Kw * w + 64 * r + 4 * lb) * sizeof(float)));
                    bvq[hf][r] = u32x2_t{pack_bf16x2(f.x, f.y), pack_bf16x2(f.z, f.w)};
                }
            }
        for (int t = 1; t < S; t++) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                if (hf == 1 && GC == 4) continue;
                if (hf == 0) { HSTAMPQ(8) }
                u32x2_t bv[NRK];
                bool have = true;
#pragma unroll
                for (int r = 0; r < NRK; r++) {
                    bv[r] = bvq[hf][r];
                    have = have && hxb_ready(bv[r]);
                }
                if (t > 1 && !__all(have)) { // the fragment requested ahead came back incomplete: fetch until no word is the sentinel
                    const int slot = (t - 1 + ring_base) & (HX_RING - 1);
                    const int off = (int)((((size_t)slot * B + colv[hf]) * N + Kw * w + 4 * lb) * sizeof(unsigned short));
                    bool ok = false;
                    for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                        bool gd = true;
#pragma unroll
                        for (int r = 0; r < NRK; r++) {
                            bv[r] = ld_lane ? __builtin_amdgcn_raw_buffer_load_b64(rX, off + 128 * r, 0, 16) : u32x2_t{0u, 0u};
                            gd = gd && hxb_ready(bv[r]);
                        }
                        if (__all(gd)) {
                            ok = true;
                            break;
                        }
                        if ((spins & 255) == 255 && __hip_atomic_load(p.abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    }
                    if (!ok) {
                        give_up();
                        return;
                    }
                }
                if (hf == 0) { HSTAMPQ(9) } else { HSTAMPQ(5) }
                // the OTHER half's next fragment -- (t, B) after A's product, (t+1, A) after B's -- is requested behind the matrix
                // instructions and looked at when that half is next
                const int nslot = (t - 1 + hf + ring_base) & (HX_RING - 1);
                const int noff = (int)((((size_t)nslot * B + colv[hf ^ 1]) * N + Kw * w + 4 * lb) * sizeof(unsigned short));
                const bool req = GC == 8 && (hf == 0 ? t > 1 : t + 1 < S);
                f32x4 acc[NSET][2];
#pragma unroll
                for (int sx = 0; sx < NSET; sx++) acc[sx][0] = acc[sx][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < NRK; r++) {
                    const fbf16x4_t av = __builtin_bit_cast(fbf16x4_t, bv[r]);
                    static_for<0, NAB / 2>([&](auto ic) {
                        constexpr int ab = 2 * decltype(ic)::value;
#pragma unroll
                        for (int sx = 0; sx < NSET; sx++) {
                            acc[sx][0] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, __builtin_bit_cast(fbf16x4_t, wq[r][sx][ab]), acc[sx][0], 4, ab, 0);
                            acc[sx][1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, __builtin_bit_cast(fbf16x4_t, wq[r][sx][ab + 1]), acc[sx][1], 4, ab + 1, 0);
                        }
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
                if (req) {
#pragma unroll
                    for (int r = 0; r < NRK; r++)
                        bvq[hf ^ 1][r] = ld_lane ? __builtin_amdgcn_raw_buffer_load_b64(rX, noff + 128 * r, 0, 16) : u32x2_t{0u, 0u};
                }
                if (GC == 4) { // one half only: its next fragment cannot have been published yet, the poll above fetches it
#pragma unroll
                    for (int r = 0; r < NRK; r++) bvq[0][r] = u32x2_t{HX_SENT, HX_SENT};
                }
                __builtin_amdgcn_sched_barrier(0);
                if (hf == 0) { HSTAMPQ(10) } else { HSTAMPQ(6) }
                // lane (unit u, gate j), register i = column i of the half: one row of the image per (wave, column).  No workgroup
                // barrier: the gating waves of the half count the images in (LDS operations of a wave execute in order)
                float *rp = red_ + (size_t)(hf * 2 + (t & 1)) * (8 * 4 * RS);
#pragma unroll
                for (int sx = 0; sx < NSET; sx++)
#pragma unroll
                    for (int i = 0; i < 4; i++) rp[(w * 4 + i) * RS + 64 * sx + l] = acc[sx][0][i] + acc[sx][1][i];
                asm volatile("" ::: "memory");
                if (l == 0) __hip_atomic_fetch_add(&s_done[hf][t & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (hf == 0) { HSTAMPQ(11) } else { HSTAMPQ(7) }
            }
        }
    } else {
        // ---------------- gating waves: NEH per half, 16 units each; lane = column*16 + unit ----------------
        const int hf = (w - 8) / NEH, uh = (w - 8) % NEH;
        if (hf == 1 && GC == 4) return;
        __builtin_amdgcn_s_setprio(3);
        const int gc = l >> 4, gu = l & 15;
        const int col = p.col0 + GC * g + 4 * hf + gc, colc = col < B ? col : B - 1;
        const int j = UW * kb + 16 * uh + gu;
        float bs[4], cprev, wx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gt = 0; gt < 4; gt++) bs[gt] = p.bias[gt * N + j];
        cprev = p.C[(size_t)colc * N + j];
        int xnext = p.xi[1 * B + colc];
        bool local_pub = false;
        const float *rp0 = red_ + (size_t)(hf * 2) * (8 * 4 * RS) + gc * RS + 64 * uh + 4 * gu;
        for (int t = 1; t < S; t++) {
#pragma unroll
            for (int gt = 0; gt < 4; gt++) wx[gt] = 0.f;
            if (xnext >= 0) {
#pragma unroll
                for (int gt = 0; gt < 4; gt++) wx[gt] = p.W[(size_t)xnext * G4 + gt * N + j]; // R/lstm.cc:176 with a one-hot x
            }
            if (t + 1 < S) xnext = p.xi[(t + 1) * B + colc];
            HSTAMPQ(0)
            {   // all eight partial-sum images of this half and step are in LDS
                bool in = false;
                for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                    in = __hip_atomic_load(&s_done[hf][t & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= 8u * (unsigned)((t + 1) / 2);
                    if (in || __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!in) {
                    give_up();
                    return;
                }
                asm volatile("" ::: "memory");
            }
            const float *rp = rp0 + (t & 1) * (8 * 4 * RS);
            HSTAMPQ(1)
            if (XCD_LOCAL && t == 2) { // every workgroup of the group has published h_1, hence its XCC id before it
                unsigned mine = 0;
                bool same = true;
                if (l < NB) {
                    mine = __hip_atomic_load(xcc_tab + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    same = (mine >> 4) == p.epoch;
                }
                const unsigned first = __builtin_amdgcn_readfirstlane(mine);
                if (l < NB) same = same && mine == first;
                local_pub = XCD_FORCE_LOCAL || __all(same);
            }
            float4 uhs = *reinterpret_cast<const float4 *>(rp); // the four gates of (unit, column), wave 0's K-slice
#pragma unroll
            for (int ww = 1; ww < 8; ww++) {
                const float4 v = *reinterpret_cast<const float4 *>(rp + ww * 4 * RS);
                uhs.x += v.x;
                uhs.y += v.y;
                uhs.z += v.z;
                uhs.w += v.w;
            }
            const float pre0 = (wx[0] + uhs.x) + bs[0], pre1 = (wx[1] + uhs.y) + bs[1]; // R/lstm.cc:176
            const float pre2 = (wx[2] + uhs.z) + bs[2], pre3 = (wx[3] + uhs.w) + bs[3];
            const float ig = p_sigm<FAST>(pre0), og = p_sigm<FAST>(pre1), fg = p_sigm<FAST>(pre2); // :179
            const float ug_ = p_tanh<FAST>(pre3);                                                 // :182
            const float cv = p_tanh<FAST>(ig * ug_ + fg * cprev);                                 // :185-189
            const float hv = og * cv;                                                             // :192
            cprev = cv;
            float4 h4; // four consecutive units sit in the four lanes of a quad
            h4.x = dpp_f<0x00>(hv);
            h4.y = dpp_f<0x55>(hv);
            h4.z = dpp_f<0xAA>(hv);
            h4.w = dpp_f<0xFF>(hv);
            HSTAMPQ(2)
            // the reset this wave issued a step ago (and every older store) has completed before h_t can be seen
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            HSTAMPQ(3)
            if ((gu & 3) == 0 && col < B) {
                const u32x2_t hp = {hxb_canon(pack_bf16x2(h4.x, h4.y)), hxb_canon(pack_bf16x2(h4.z, h4.w))};
                const u32x2_t sent = {HX_SENT, HX_SENT};
                const size_t e_pub = ((size_t)((t + ring_base) & (HX_RING - 1)) * B + col) * N + j;
                const size_t e_rst = ((size_t)((t + 2 + ring_base) & (HX_RING - 1)) * B + col) * N + j;
                if (XCD_LOCAL && local_pub) {
                    *reinterpret_cast<u32x2_t *>(reinterpret_cast<unsigned short *>(p.Hxb) + e_pub) = hp;
                    *reinterpret_cast<u32x2_t *>(reinterpret_cast<unsigned short *>(p.Hxb) + e_rst) = sent;
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(hp, rX, (int)(e_pub * sizeof(unsigned short)), 0, 16);
                    __builtin_amdgcn_raw_buffer_store_b64(sent, rX, (int)(e_rst * sizeof(unsigned short)), 0, 16);
                }
                // off the chain: the plain copies the time-batched products read (fp32 H, bf16 Hb)
                *reinterpret_cast<u32x2_t *>(p.Hb + ((size_t)t * B + col) * N + j) = u32x2_t{pack_bf16x2(h4.x, h4.y), pack_bf16x2(h4.z, h4.w)};
                *reinterpret_cast<float4 *>(p.H + ((size_t)t * B + col) * N + j) = h4;
            }
            if (col < B) {
                float *gcp = p.G + ((size_t)t * B + col) * G4 + j;
                gcp[0] = ig;
                gcp[N] = og;
                gcp[2 * N] = fg;
                gcp[3 * N] = ug_;
                p.C[((size_t)t * B + col) * N + j] = cv;
            }
            HSTAMPQ(4)
        }
    }
#undef HSTAMPQ
}

// ------------------------------------------------------------------------------------------------
// backward recurrence, two half-groups per workgroup (N = 512 / 256, 8-column groups; the backward twin of
// k_fwd_persistent6): grid (N/16, ceil(B/8)), 768 threads.  The eight columns of workgroup (kb, g) are two independent
// recurrences of four columns, A and B, advanced alternately: while A's step is summed, computed elementwise and handed
// on, the product waves work on B, and the other way round.  This part holds what the roles share -- arguments, the LDS
// block, the two side waves -- the recurrence itself is k_bwd_scatter below.
//   waves 0-7  product (bwds_product), and in FUSE mode dWhy[:, units] += dy_t h_t^T (R/lstm.cc:226) in their idle time
//   wave 8 / 9 elementwise (R/lstm.cc:228-247,256) of half A / B (bwds_elementwise); db (:252) in their registers
//   wave 10    FUSE: dW[rows, x] += dg_t[rows, column] for the column's input byte x (R/lstm.cc:251): a [257][64] LDS
//              table fed through a four-deep LDS copy of dg.  Partial blocks per column group go to gpart.
//   wave 11    the output-layer term dhy_t = Why^T dy_t (R/lstm.cc:228), up to four steps ahead of the chain
// No workgroup barrier in the loop: LDS counters (s_done: dg_t handed to the product waves; s_stage / s_tab: dg copies for
// the dW wave; s_ol: dhy ready).
// ------------------------------------------------------------------------------------------------
constexpr int BWDH_THREADS = 768;
constexpr int BWDH_RED = 2 * 2 * 8 * 256, BWDH_YTMP = 256, BWDH_DHY = 4 * 128, BWDH_STAGE = 2 * 4 * 256;
constexpr size_t bwdh_lds_bytes(bool fuse) {
    return sizeof(float) * (size_t)(16 + BWDH_RED + BWDH_YTMP + BWDH_DHY + (fuse ? BWDH_STAGE + 257 * 64 : 0));
}
#define HSTAMP(wave, k)                                                                                        \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = __builtin_amdgcn_s_memtime();
// The four roles are separate functions for readability (inlined); what they share comes through BwdhArgs (the kernel's
// arguments) and the LDS block.  (Not inlined, the argument block is read through a generic pointer into vector registers
// and every buffer access turns into a waterfall loop: 1.5 ms.)
struct BwdhArgs {
    const float4 *Ubwd5;
    float *DG;
    const float *Why, *dY, *G, *C, *H;
    const int32_t *xi;
    float *gpart, *DGx;
    unsigned *cnt, *abortp;
    unsigned epoch;
    int ring_base, S, B, cfg;
    unsigned long long *stamps;
};
constexpr int BWDH_SYNC = 16; // words at the head of the LDS block: abort, -, -, s_stage[2], s_dy, s_ol, s_tab, s_done[2][2], s_loc[2], s_wy, consumed
#define BWDH_COMMON(p)                                                                                                          \
    constexpr int N = N_, G4 = 4 * N, Kw = N / 2, NL = Kw / 64; /* NL 16-byte loads per lane, half and step */                  \
    static_assert(N == 512 || N == 256, "two-half backward form: hidden 512 or 256");                                           \
    constexpr int WS = 256;                                      /* partial-sum image: [wave][column*16 + unit][Y] */           \
    extern __shared__ __attribute__((aligned(16))) float lds[];                                                                 \
    unsigned *sync_ = reinterpret_cast<unsigned *>(lds);                                                                        \
    int *s_abort = reinterpret_cast<int *>(sync_);                                                                              \
    unsigned *s_stage = sync_ + 3, *s_dy = sync_ + 5, *s_ol = sync_ + 6, *s_tab = sync_ + 7;                                    \
    /* s_done[half][step parity]: images written, 8 per step of that parity.  Per parity because a product wave fed by     */  \
    /* other workgroups can be one step (never two) ahead of a wave of its own workgroup that waits for a slow load: with   */  \
    /* one count per half its early count would stand in for the late wave's missing one.                                   */  \
    unsigned *s_done = sync_ + 8;                                                                                               \
    float *red = lds + BWDH_SYNC;     /* [half][step parity][8 * WS] */                                                         \
    float *ytmp = red + BWDH_RED;     /* wave 11: [column*16 + unit][k-class Y] of the half it is folding */                    \
    float *dhyb = ytmp + BWDH_YTMP;   /* [step & 3][column][unit]: Why^T dy of the step */                                      \
    float *stage = dhyb + BWDH_DHY;   /* FUSE: [half][step & 3][column][gate*16 + unit] */                                      \
    float *dWt = stage + BWDH_STAGE;  /* FUSE: [257][64] */                                                                     \
    const float4 *__restrict__ Ubwd5 = p.Ubwd5;                                                                                 \
    float *DG = p.DG, *DGx = p.DGx, *gpart = p.gpart;                                                                           \
    const float *__restrict__ Why = p.Why, *__restrict__ dY = p.dY, *__restrict__ G = p.G, *__restrict__ C = p.C,               \
                             *__restrict__ H = p.H;                                                                             \
    const int32_t *__restrict__ xi = p.xi;                                                                                      \
    unsigned *cnt = p.cnt, *abortp = p.abortp;                                                                                  \
    const unsigned epoch = p.epoch;                                                                                             \
    const int ring_base = p.ring_base, S = p.S, B = p.B, cfg = p.cfg;                                                           \
    unsigned long long *stamps = p.stamps;                                                                                      \
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;                                                                    \
    /* cfg bits 16-19: pinned launch (fewer than 8 groups): gridDim.y = 8, workgroup i runs on XCD i % 8 and group g is the   */ \
    /* workgroups of XCD g; those of the other XCDs leave at once (k_bwd_scatter)                                             */ \
    const int pin_ng_ = (cfg >> 16) & 15;                                                                                       \
    /* cfg bits 20-27: first column group of this launch (a wide batch runs as launches over column ranges); gq = the group    */ \
    /* of the whole batch: columns, the partial gradient block and the ring region go by it                                   */ \
    /* cfg bit 28: ONE half per workgroup (4-column groups; half B's elementwise wave idles): see k_fwd_persistent6          */ \
    const int g0_ = (cfg >> 20) & 255, GC_ = (cfg >> 28) & 1 ? 4 : 8, NRG_ = 2 * ((B + 7) / 8);                                 \
    /* BWDS_TAGGED (compile time, experiment): the ring without reset stores, the parity of a slot's use count in the last   */ \
    /* mantissa bit of every word (k_bwd_scatter_bf16); ring_base is then a publication number                               */ \
    constexpr bool tagged_ = BWDS_TAGGED != 0;                                                                                  \
    const int NBK = gridDim.x, NG = pin_ng_ ? pin_ng_ : (int)gridDim.y;                                                         \
    const int lin_ = blockIdx.x + NBK * blockIdx.y;                                                                             \
    /* cfg bit 16 (tests): keep the dispatch-order mapping, which spreads every column group over all XCDs -- the placement  */ \
    /* the XCD-local publish must detect and decline                                                                         */ \
    const bool remap_ = GROUP_REMAP && !(cfg & 16);                                                                             \
    const int kb = pin_ng_ ? lin_ >> 3 : remap_ ? lin_ / NG : (int)blockIdx.x;                                                  \
    const int g = pin_ng_ ? lin_ & 7 : remap_ ? lin_ % NG : (int)blockIdx.y;                                                    \
    const int gq = g + g0_;                                                                                                     \
    const int rg0_ = GC_ * gq / 4; /* ring regions are counted in halves: region of (group, half) = its first column / 4 */   \
    (void)tagged_, (void)rg0_, (void)NRG_; /* (not every role uses every one of these) */                                       \
    const __amdgpu_buffer_rsrc_t rDG = make_rsrc(DGx, (size_t)HX_RING * G4 * B * sizeof(float));                                \
    unsigned *xcc_tab = cnt + (size_t)g * CNT_SLOTS * CNT_STRIDE;                                                               \
    /* a wave that gives up: the abort word ends the launch everywhere, the LDS word releases this workgroup's other waves */   \
    auto give_up = [&]() {                                                                                                      \
        if (l == 0) {                                                                                                           \
            __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                         \
            __hip_atomic_store(s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                                     \
        }                                                                                                                       \
    };                                                                                                                          \
    auto lds_wait = [&](unsigned *word, unsigned want) -> bool { /* bounded like every other spin of these kernels */           \
        for (int spins = 0; spins <= SPIN_LIMIT; spins++) {                                                                     \
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) {                              \
                asm volatile("" ::: "memory");                                                                                  \
                return true;                                                                                                    \
            }                                                                                                                   \
            if (__hip_atomic_load(s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;                       \
            __builtin_amdgcn_s_sleep(1);                                                                                        \
        }                                                                                                                       \
        return false;                                                                                                           \
    };                                                                                                                          \
    /* FUSE, after the loops (every wave, at the end of its role): barrier, the dW table and the db / dWhy partial blocks */    \
    float *base = FUSE ? gpart + (size_t)gq * ((size_t)G4 * 256 + (size_t)G4 * N + G4 + (size_t)256 * N) : nullptr;              \
    float *dbs = red; /* epilogue scratch in `red` (free then): [column 0..7][gate][unit] */                                    \
    auto table_out = [&]() { /* dW partial: table row r = gate*16 + unit  ->  gradient row gate*N + 16*kb + unit */             \
        static_assert(BWDH_THREADS == 768, "roles: waves 0-7 product, 8-9 elementwise, 10 dW table, 11 output layer");            \
        const int pid = tid < 512 ? tid : tid - 128; /* called by the product waves and wave 10 (threads 640-703): 576 threads */  \
        for (int i = pid; i < 256 * 64; i += 576) {                                                                             \
            const int x = i >> 6, r = i & 63;                                                                                   \
            base[(size_t)x * G4 + (r >> 4) * N + 16 * kb + (r & 15)] = dWt[i];                                                  \
        }                                                                                                                       \
    };                                                                                                                          \
    (void)give_up, (void)lds_wait, (void)table_out, (void)S,                                                                    \
    (void)Ubwd5, (void)DG, (void)Why, (void)dY, (void)G, (void)C, (void)H, (void)xi, (void)epoch, (void)ring_base, (void)cfg,   \
        (void)stamps, (void)NBK, (void)rDG, (void)xcc_tab, (void)s_done, (void)s_stage, (void)s_dy, (void)s_ol, (void)s_tab,    \
        (void)ytmp, (void)dhyb, (void)stage, (void)dWt, (void)base, (void)dbs, (void)Kw, (void)NL, (void)WS,         \
        (void)kb, (void)w;

template <int N_, bool FUSE, bool STAMP> __device__ __forceinline__ void bwdh_output_layer(const BwdhArgs &p) {
    BWDH_COMMON(p)
    // ---------------- wave 11: the output-layer term dhy_t = Why^T dy_t (R/lstm.cc:228), ahead of the chain ----------------
    // The product waves' arrangement with K = 256: lane (Y, z', i) loads dy_t[m = 64q + 16Y + 4z' + r][column 4*half + i]
    // (16 bytes, q = 0..3), weights Why[m][unit 4z + j] in 64 registers, 64 instructions v_mfma_f32_4x4x1 per half; the four
    // k-classes Y are folded through LDS.  Matrix instructions, but at the lowest priority and up to four steps ahead of the
    // chain: they fill gaps the product waves leave.  (On the vector ALU with dy staged in LDS the same work measured 12 600
    // cycles a step -- an LDS round trip per group of reads -- and held the whole recurrence back.)
    const int lY = l >> 4, lz = (l >> 2) & 3, li = l & 3;
    float4 wy[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float *wp = Why + (size_t)(16 * kb + 4 * lz + li) * 256 + 64 * q + 16 * lY; // + 4z' + r
        const float4 v0 = *reinterpret_cast<const float4 *>(wp), v1 = *reinterpret_cast<const float4 *>(wp + 4);
        const float4 v2 = *reinterpret_cast<const float4 *>(wp + 8), v3 = *reinterpret_cast<const float4 *>(wp + 12);
        wy[4 * q + 0] = float4{v0.x, v1.x, v2.x, v3.x};
        wy[4 * q + 1] = float4{v0.y, v1.y, v2.y, v3.y};
        wy[4 * q + 2] = float4{v0.z, v1.z, v2.z, v3.z};
        wy[4 * q + 3] = float4{v0.w, v1.w, v2.w, v3.w};
    }
    int yofs[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const int c = GC_ * gq + 4 * hf + li;
        yofs[hf] = (c < B ? c : B - 1) * 256 + 16 * lY + 4 * lz;
    }
    // (named registers, not arrays: see the note on scratch memory in the git history of this file)
    float4 a0, a1, a2, a3, b0, b1, b2, b3; // dy of the step worked on next, halves A and B; requested a step ahead
#define OL_REQUEST(tu)                                                   \
    do { /* dY holds steps 1.. at column (t-1)*B + b */                  \
        const float *dp_ = dY + (size_t)((tu) - 1) * B * 256;            \
        a0 = *reinterpret_cast<const float4 *>(dp_ + yofs[0]);           \
        a1 = *reinterpret_cast<const float4 *>(dp_ + yofs[0] + 64);      \
        a2 = *reinterpret_cast<const float4 *>(dp_ + yofs[0] + 128);     \
        a3 = *reinterpret_cast<const float4 *>(dp_ + yofs[0] + 192);     \
        b0 = *reinterpret_cast<const float4 *>(dp_ + yofs[1]);           \
        b1 = *reinterpret_cast<const float4 *>(dp_ + yofs[1] + 64);      \
        b2 = *reinterpret_cast<const float4 *>(dp_ + yofs[1] + 128);     \
        b3 = *reinterpret_cast<const float4 *>(dp_ + yofs[1] + 192);     \
    } while (0)
#define Y_STEP(av, wq)                                              \
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.x, c0, 2, 0, 0); \
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.y, c1, 2, 1, 0); \
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.z, c2, 2, 2, 0); \
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.w, c3, 2, 3, 0);
#define OL_HALF(d0, d1, d2, d3, hf)                                                                             \
    do {                                                                                                        \
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;                                             \
        Y_STEP(d0.x, wy[0]) Y_STEP(d0.y, wy[1]) Y_STEP(d0.z, wy[2]) Y_STEP(d0.w, wy[3])                         \
        Y_STEP(d1.x, wy[4]) Y_STEP(d1.y, wy[5]) Y_STEP(d1.z, wy[6]) Y_STEP(d1.w, wy[7])                         \
        Y_STEP(d2.x, wy[8]) Y_STEP(d2.y, wy[9]) Y_STEP(d2.z, wy[10]) Y_STEP(d2.w, wy[11])                       \
        Y_STEP(d3.x, wy[12]) Y_STEP(d3.y, wy[13]) Y_STEP(d3.z, wy[14]) Y_STEP(d3.w, wy[15])                     \
        float *yp_ = ytmp + 4 * (4 * lz + li) + lY; /* lane (Y, z, j): unit 4z + j, register r = column r */    \
        _Pragma("unroll") for (int r = 0; r < 4; r++) yp_[64 * r] = (c0[r] + c1[r]) + (c2[r] + c3[r]);          \
        asm volatile("" ::: "memory");                                                                          \
        const float4 v_ = *reinterpret_cast<const float4 *>(ytmp + 4 * l); /* lane = column*16 + unit */        \
        asm volatile("" ::: "memory");                                                                          \
        dst[64 * (hf) + l] = (v_.x + v_.y) + (v_.z + v_.w);                                                     \
    } while (0)
    // FUSE: h_tu for the dWhy sums (lane l: unit l & 15 of columns l >> 4 and 4 + (l >> 4); zero for a padding column)
    float hcur[2] = {0.f, 0.f}, hnext[2] = {0.f, 0.f};
    auto h_request = [&](int tu) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int col = GC_ * gq + 4 * k + (l >> 4);
            hnext[k] = col < B ? H[((size_t)tu * B + col) * N + 16 * kb + (l & 15)] : 0.0f;
        }
    };
    const bool one_half = GC_ == 4; // half B's columns belong to another workgroup: nothing staged, nothing computed for them
#define OL_DROP_B()                                                              \
    do {                                                                         \
        if (one_half) {                                                          \
            b0 = b1 = b2 = b3 = float4{0.f, 0.f, 0.f, 0.f};                      \
            hnext[1] = 0.0f;                                                     \
        }                                                                        \
    } while (0)
    OL_REQUEST(S - 1);
    if (FUSE) h_request(S - 1);
    OL_DROP_B();
    for (int tu = S - 1; tu >= 1; tu--) {
        const int t = tu;
        hcur[0] = hnext[0], hcur[1] = hnext[1];
        HSTAMP(11, 12)
        // slot tu & 3 held the term of step tu+4: both elementwise waves must have read it
        if (tu + 4 <= S - 1) {
            const unsigned need = (unsigned)(S - (tu + 4));
            if (!lds_wait(&s_stage[0], need) || (!one_half && !lds_wait(&s_stage[1], need))) {
                give_up();
                break;
            }
        }
        HSTAMP(11, 13)
        if (FUSE) { // dy_tu and h_tu for the dWhy sums of the product waves (three-deep ring; sync_[15]: waves that have consumed)
            if (tu + 3 <= S - 1 && !lds_wait(&sync_[15], 8u * (unsigned)(S - tu - 3))) {
                give_up();
                break;
            }
            float *sp = red + 1024 + (tu % 3) * (8 * 272) + 16 * lY + 4 * lz;
            *reinterpret_cast<float4 *>(sp + li * 272) = a0;
            *reinterpret_cast<float4 *>(sp + li * 272 + 64) = a1;
            *reinterpret_cast<float4 *>(sp + li * 272 + 128) = a2;
            *reinterpret_cast<float4 *>(sp + li * 272 + 192) = a3;
            *reinterpret_cast<float4 *>(sp + (4 + li) * 272) = b0;
            *reinterpret_cast<float4 *>(sp + (4 + li) * 272 + 64) = b1;
            *reinterpret_cast<float4 *>(sp + (4 + li) * 272 + 128) = b2;
            *reinterpret_cast<float4 *>(sp + (4 + li) * 272 + 192) = b3;
            float *hp = red + 1024 + (tu % 3) * (8 * 272) + 256 + (l & 15);
            hp[(l >> 4) * 272] = hcur[0];
            hp[(4 + (l >> 4)) * 272] = hcur[1];
            asm volatile("" ::: "memory");
            if (l == 0) __hip_atomic_store(&sync_[14], (unsigned)(S - tu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        float *dst = dhyb + (tu & 3) * 128;
        OL_HALF(a0, a1, a2, a3, 0);
        if (!one_half) OL_HALF(b0, b1, b2, b3, 1);
        HSTAMP(11, 14)
        if (tu >= 2) OL_REQUEST(tu - 1); // in flight until this wave comes round again
        if (FUSE && tu >= 2) h_request(tu - 1);
        OL_DROP_B();
        asm volatile("" ::: "memory");
        if (l == 0) __hip_atomic_store(s_ol, (unsigned)(S - tu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        HSTAMP(11, 15)
    }
#undef OL_DROP_B
#undef OL_HALF
#undef Y_STEP
#undef OL_REQUEST
    if (FUSE) {
        __syncthreads();
        __syncthreads();
    }
}
template <int N_, bool FUSE, bool STAMP> __device__ __forceinline__ void bwdh_weight_sums(const BwdhArgs &p) {
    BWDH_COMMON(p)
    // ---------------- wave 10: the dW sums (the dWhy sums run in the product waves) ----------------
    //   dW[:, x] += dg_t[:, column]  (R/lstm.cc:251)  one lane per row (gate*16 + unit) of the workgroup, [257][64] LDS table
    // input bytes of the eight columns: lane c < 8 loads column c's, a step ahead (a vector load on purpose: scalar loads
    // share the LDS wait counter and would serialise with every LDS access below)
    auto xfetch = [&](int tu) -> int {
        const int col = GC_ * gq + (l & 7);
        const int x = col < B && (l & 7) < GC_ ? xi[(size_t)tu * B + col] : -2;
        return x == -1 ? 256 : x; // -1: empty input column -> bucket 256; -2: padding column, skipped
    };
    int xnext = xfetch(S - 1);
    for (int tu = S - 1; tu >= 1; tu--) {
        const int xcur = xnext;
        if (tu >= 2) xnext = xfetch(tu - 1);
        bool ok = true;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            if (hf == 1 && GC_ == 4) continue;
            if (!ok) break;
            if (!lds_wait(&s_stage[hf], (unsigned)(S - tu))) {
                ok = false;
                break;
            }
            const float *sg_ = stage + (hf * 4 + (tu & 3)) * 256;
            float sv[4];
#pragma unroll
            for (int c = 0; c < 4; c++) sv[c] = sg_[c * 64 + l];
#pragma unroll
            for (int c = 0; c < 4; c++) { // columns in order (two may share an input byte): deterministic sums
                const int x = __builtin_amdgcn_readlane(xcur, 4 * hf + c);
                if (x >= 0) dWt[x * 64 + l] += sv[c]; // (as ds_add_f32 without return: 285 -> 307 us -- LDS float atomics hold the LDS up for the chain's waves)
            }
        }
        if (!ok) {
            give_up();
            break;
        }
        asm volatile("" ::: "memory");
        if (l == 0) __hip_atomic_store(s_tab, (unsigned)(S - tu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (__hip_atomic_load(s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
    table_out();
    __syncthreads();
}
// ------------------------------------------------------------------------------------------------
// backward recurrence, SCATTER form (N = 512 / 256, 8-column groups, two alternating 4-column halves per workgroup).
//
// dhnext = U^T dg_{t+1} contracts over the 4N gate rows and yields N values per column.  The forward recurrence gathers its
// input (h: N values per column) and keeps its 64 output rows; doing the same here -- 16 OUTPUT units per workgroup, the
// whole dg_{t+1} gathered -- means 4N values per column (32 KB per half and step), which rounds 1 and 2 fetched through a
// hint poll and a second round trip and folded from 8 x 4 partial sums through LDS (336 us a window).  The backward
// product is split the other way round: every workgroup keeps the 64 gate rows IT PRODUCES (its 16 units x 4 gates) as
// its slice of the contraction -- the same slice of U the forward recurrence holds -- and multiplies its own dg_t,
// straight from LDS, into partial sums for ALL N outputs; those are scattered to the workgroups that own the outputs (64
// values to each) and summed there, in source order, by the elementwise wave (285 us with the side waves as they were,
// 259 us with the dWhy sums moved into the product waves):
//   product waves 0 .. N/64-1 : wave w owns outputs [64w, 64w+64), i.e. destination workgroups 4w .. 4w+3; v_mfma_f32_4x4x1,
//        block = four outputs (lane = 4*block + j), CBSZ = 4 / ABID = ab: all sixteen blocks read dg[k = 4ab + r][column i]
//        from block ab of ONE 16-byte LDS read; 64 instructions per half and step, weights Ubwd6 in 64 registers;
//        a lane ends with the four columns of one output: one 16-byte store into the destination's ring slot.
//        No K split over waves, so no partial-sum images, no counts, no fold.
//   waves 8 / 9 (half A / B)  : poll the N/16 sources' 256-byte pieces of the slot (N/64 16-byte loads per lane, the data
//        is the flag), sum them in source order (lane-local over every fourth source, then a 4 x 4 transpose-sum over the
//        four 16-lane rows), do R/lstm.cc:228-247,256, hand dg_t to the product waves through LDS (1 KB) and store the
//        plain DG off the chain.
// Per half and step a workgroup now receives 8 KB instead of 32 KB, in one round trip instead of two, and the chain is
//   partials stored -> L2 -> poll hit -> sum + elementwise -> LDS -> 64 matrix instructions -> partials stored.
// Ring Qx[slot][group][half][destination][source][unit][column] (DGx buffer; bwd_ring_floats), data-as-flag as HX_RING:
// P(t) (the product from dg_t, consumed by step t-1) writes slot(t) = (t + ring_base) & 3 for t = S-1 .. 2 and then resets
// its own words of slot(t-2); the launch-to-launch advance is (ring_base - (S-2)) & 3 (tools/probes/ring_protocol_sim.py).
//   * the reset is safe: a source holds ALL of Q_{t+1} before it computes Q_t, so every destination has finished the step
//     that read Q_{t+2} (= slot(t-2));
//   * a destination polls a source's words of slot(t-2) only after it consumed that source's Q_{t-1}, stored after an
//     s_waitcnt vmcnt(0) that covers the reset: the reset is visible first.
// The dg hand-off buffer is double-buffered by step parity: step t-2 overwrites it only after Q_{t-1} of every source is
// in, each of which needed this workgroup's Q_t from every one of its product waves.
// ------------------------------------------------------------------------------------------------
#define SSTAMP(wave, k)                                                                                        \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = __builtin_amdgcn_s_memtime();
__host__ __device__ inline size_t bwds_ring_floats(int N, int B) { return (size_t)HX_RING * ((B + 7) / 8) * 2 * (N / 16) * (N / 16) * 64; }

template <int N_, bool FUSE, bool STAMP> __device__ __forceinline__ void bwds_product(const BwdhArgs &p) {
    BWDH_COMMON(p)
    constexpr int NB = N / 16, NPW = N / 64; // sources / destinations per column group; product waves
    float *dgl = red;                        // [half][step parity][column 4][k 64]: dg_t of this workgroup's own gate rows
    unsigned *s_loc = sync_ + 12;            // [half]: the XCD-local publish was verified (set by the elementwise wave)
    __builtin_amdgcn_s_setprio(2);           // below the elementwise waves (3), above the side waves 10 and 11 (0)
    const bool active = w < NPW;
    const int lb = l >> 2, lj = l & 3;
    float4 a[16];
#pragma unroll
    for (int ab = 0; ab < 16; ab++) a[ab] = active ? Ubwd5[(((size_t)kb * NPW + w) * 16 + ab) * 64 + l] : float4{0.f, 0.f, 0.f, 0.f};
    // The weights are complete in their registers before the loop starts: left to its own bookkeeping the compiler waits
    // for them lazily INSIDE the loop with counted waits (vmcnt(23) .. vmcnt(8) between the matrix instructions), which
    // from the second step on wait for whatever else is in flight -- the dWhy operand loads -- on the chain.
#pragma unroll
    for (int ab = 0; ab < 16; ab++) asm volatile("" ::"v"(a[ab].x), "v"(a[ab].y), "v"(a[ab].z), "v"(a[ab].w));
    const int d = 4 * w + (lb >> 2), u = 4 * (lb & 3) + lj; // this lane's output: unit u of destination workgroup d
    const __amdgpu_buffer_rsrc_t rQ = make_rsrc(DGx, bwds_ring_floats(N, B) * sizeof(float));
    auto qoff = [&](int tt, int hf) { // float offset of this lane's 16 bytes in slot(tt)
        const size_t slot = (size_t)((tagged_ ? ring_base + (S - 1 - tt) : tt + ring_base) & (HX_RING - 1));
        return (((slot * NRG_ + rg0_ + hf) * NB + d) * NB + kb) * 64 + (size_t)u * 4;
    };
    // FUSE: dWhy[:, units] += dy_t h_t^T (R/lstm.cc:226) between the products, while the wave would otherwise wait for the
    // next hand-off.  v_mfma_f32_4x4x1, one instruction = one column c, 64 output rows (lane l = row 64mg + l) and four units:
    // D[i][j] += h_t[unit 4q + i][c] * dy_t[row][c], the h operand broadcast from block q of a register that holds the 16
    // units in lanes 0-15 (CBSZ = 4 / ABID = q).  The 16 accumulators (mg, q) are dealt two to a wave (mg = w >> 1,
    // q = 2(w & 1), 2(w & 1) + 1): 16 instructions a step and wave, spread over all four SIMDs.  (As one side wave of 128
    // instructions at the lowest priority they cost the chain 30 us a window: every matrix instruction of the two product
    // waves on that SIMD queued behind one of them.)
    const int wmg = w >> 1, wq0 = 2 * (w & 1);
    f32x4 wacc0 = {0.f, 0.f, 0.f, 0.f}, wacc1 = wacc0;
    // Operands: wave 11, which runs ahead of the chain and holds dy_tu in registers for its own product anyway, leaves dy_tu
    // (8 columns x 256 rows) and h_tu (8 x 16 units) in a three-deep LDS ring; each product wave reads its rows from there.
    // (Fetched by the product waves themselves -- 16 scalar loads a wave and step, 128 a workgroup -- the loads, not the
    // matrix instructions, cost the chain 17 us a window: the chain's polls and stores queue behind them in the vector
    // memory pipe.  Staged by wave 10, whose dW table work runs late at the lowest priority, the product waves waited for
    // it: 300 us.)
    float hq[8], dq[8]; // h_tu[unit l & 15][column c] (zero for a padding column: no dWhy from it), dy_tu[64 wmg + l][column c]
    unsigned *s_wy = sync_ + 14;   // steps staged by wave 10 (S - tu); sync_[15]: product waves that have consumed, 8 a step
    float *wyS = red + 1024;       // [step % 3][8 columns][256 rows | 16 units]: beside the dg hand-off buffers in `red`
    auto wfetch = [&](int tu) -> bool {
        if (!lds_wait(s_wy, (unsigned)(S - tu))) return false;
        const float *sp = wyS + (tu % 3) * (8 * 272);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            dq[c] = sp[c * 272 + 64 * wmg + l];
            hq[c] = sp[c * 272 + 256 + (l & 15)];
        }
        return true;
    };
    auto wrelease = [&]() {
        asm volatile("" ::: "memory");
        if (l == 0) __hip_atomic_fetch_add(&sync_[15], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wstep = [&]() {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            if (wq0 == 0) {
                wacc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq[c], dq[c], wacc0, 4, 0, 0);
                wacc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq[c], dq[c], wacc1, 4, 1, 0);
            } else {
                wacc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq[c], dq[c], wacc0, 4, 2, 0);
                wacc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq[c], dq[c], wacc1, 4, 3, 0);
            }
        }
    };
    bool live = true;
    for (int t = S - 1; t >= 2 && live; t--) {
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            if (hf == 1 && GC_ == 4) continue;
            if (!live) break;
            if (hf == 0) { SSTAMP(3, 8) }
            // E(t) of this half has written dg_t: steps S-1 .. t of the parity of t, (S-1-t)/2 + 1 of them.  (Spinning on the LDS
            // word without lds_wait's 64-cycle pauses: 290 -> 303 us -- eight waves hammering the LDS slow the elementwise waves.)
            if (!lds_wait(&s_done[2 * hf + (t & 1)], (unsigned)((S - 1 - t) / 2 + 1))) {
                give_up();
                live = false;
                break;
            }
            if (hf == 0) { SSTAMP(3, 9) } else { SSTAMP(3, 5) }
            if (active) {
                const float4 dv = *reinterpret_cast<const float4 *>(dgl + (hf * 2 + (t & 1)) * 256 + lj * 64 + 4 * lb);
                const bool local = t < S - 1 && __hip_atomic_load(&s_loc[hf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
                f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#define S6(ab)                                                            \
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.x, a[ab].x, c0, 4, ab, 0); \
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.y, a[ab].y, c1, 4, ab, 0); \
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.z, a[ab].z, c2, 4, ab, 0); \
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(dv.w, a[ab].w, c3, 4, ab, 0);
                S6(0) S6(1) S6(2) S6(3) S6(4) S6(5) S6(6) S6(7) S6(8) S6(9) S6(10) S6(11) S6(12) S6(13) S6(14) S6(15)
#undef S6
                if (hf == 0) { SSTAMP(3, 10) } else { SSTAMP(3, 6) }
                float4 q; // register r = column r of the half
                q.x = hx_canon((c0[0] + c1[0]) + (c2[0] + c3[0]));
                q.y = hx_canon((c0[1] + c1[1]) + (c2[1] + c3[1]));
                q.z = hx_canon((c0[2] + c1[2]) + (c2[2] + c3[2]));
                q.w = hx_canon((c0[3] + c1[3]) + (c2[3] + c3[3]));
                if (tagged_) {
                    const unsigned ph = (unsigned)((ring_base + (S - 1 - t)) >> 2) & 1u;
                    q = float4{tag_mark(q.x, ph), tag_mark(q.y, ph), tag_mark(q.z, ph), tag_mark(q.w, ph)};
                }
                const float4 sent = {__uint_as_float(HX_SENT), __uint_as_float(HX_SENT), __uint_as_float(HX_SENT),
                                     __uint_as_float(HX_SENT)};
                // this wave's older stores (the last reset) are complete
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const size_t e_pub = qoff(t, hf), e_rst = qoff(t - 2, hf);
                if (d < NB) { // (N = 256: every lane of an active wave has a destination; kept for clarity)
                    if (XCD_LOCAL && local) {
                        *reinterpret_cast<float4 *>(DGx + e_pub) = q;
                        if (!tagged_) *reinterpret_cast<float4 *>(DGx + e_rst) = sent;
                    } else {
                        st_sc1(q, rQ, (int)(e_pub * sizeof(float)));
                        if (!tagged_) st_sc1(sent, rQ, (int)(e_rst * sizeof(float)));
                    }
                }
                if (hf == 0) { SSTAMP(3, 11) } else { SSTAMP(3, 7) }
            }
        }
        if (FUSE && live) { // step t's share of dWhy, behind half B's product: the wave would wait for the next hand-off anyway
            if (!wfetch(t)) {
                give_up();
                live = false;
                break;
            }
            if (!(cfg & 8)) wstep();
            wrelease();
        }
    }
    if (FUSE && live) { // step 1 (or S - 1 = 1: the only step)
        if (wfetch(1)) {
            if (!(cfg & 8)) wstep();
            wrelease();
        } else
            give_up();
    }
    if (FUSE) {
        __syncthreads();
        if (__hip_atomic_load(s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return; // the host reports the abort
        {   // accumulator (mg, q): lane l = output row 64mg + l, register i = unit 4q + i
            float *Yp = base + (size_t)G4 * 256 + (size_t)G4 * N + G4 + (size_t)(16 * kb) * 256 + 64 * wmg + l;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Yp[(size_t)(4 * wq0 + i) * 256] = wacc0[i];
                Yp[(size_t)(4 * (wq0 + 1) + i) * 256] = wacc1[i];
            }
        }
        table_out();
        __syncthreads();
        if (w == 0) { // db partial: the eight columns in order
            const int gt = l >> 4, rj = l & 15;
            float sum = 0.0f;
            for (int c = 0; c < 8; c++) sum += dbs[(c * 4 + gt) * 16 + rj];
            base[(size_t)G4 * 256 + (size_t)G4 * N + gt * N + 16 * kb + rj] = sum;
        }
    }
}
template <int N_, bool FUSE, bool STAMP> __device__ __forceinline__ void bwds_elementwise(const BwdhArgs &p) {
    BWDH_COMMON(p)
    constexpr int NB = N / 16, NLD = NB / 4; // sources per column group; 16-byte loads per lane and step
    float *dgl = red;
    unsigned *s_loc = sync_ + 12;
    // ---------------- elementwise waves: wave 8 half A, wave 9 half B; lane = column*16 + unit ----------------
    const int hf = w - 8;
    __builtin_amdgcn_s_setprio(3);
    const int cc = l >> 4, jj = l & 15;
    const int ecol = GC_ * gq + 4 * hf + cc, ecolc = ecol < B ? ecol : B - 1;
    const int j = 16 * kb + jj;
    float dcn = 0.0f; // dcnext, R/lstm.cc:217
    bool local_pub = false;
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rQ = make_rsrc(DGx, bwds_ring_floats(N, B) * sizeof(float));
    // this lane's piece of a slot: sources 4i + (l >> 4), unit l & 15, the four columns
    auto qbase = [&](int tt) {
        const size_t slot = (size_t)((tagged_ ? ring_base + (S - 1 - tt) : tt + ring_base) & (HX_RING - 1));
        return (int)(((((slot * NRG_ + rg0_ + hf) * NB + kb) * NB + (size_t)(l >> 4)) * 64 + (size_t)(l & 15) * 4) * sizeof(float));
    };
    // operands that do not depend on the chain are requested a step ahead
    float ig, og, fg, ug, cv, cp;
    auto fetch = [&](int tu) {
        const float *gc = G + ((size_t)tu * B + ecolc) * G4 + j;
        ig = gc[0], og = gc[N], fg = gc[2 * N], ug = gc[3 * N];
        cv = C[((size_t)tu * B + ecolc) * N + j], cp = C[((size_t)(tu - 1) * B + ecolc) * N + j];
    };
    fetch(S - 1);
    float dhy = 0.0f;
    const bool alive = lds_wait(s_ol, 1u);
    if (alive) dhy = dhyb[((S - 1) & 3) * 128 + (4 * hf + cc) * 16 + jj];
    else give_up();
    const bool idle = hf == 1 && GC_ == 4; // one half per workgroup: wave 9 only takes part in the closing barriers (db = 0)
    for (int t = S - 1; t >= 1 && alive && !idle; t--) {
        SSTAMP(8, 0)
        float dhn = 0.0f;
        if (t < S - 1) {
            // Q_{t+1}: the partial sums every workgroup of the group computed from its dg_{t+1} for this workgroup's units
            const int off = qbase(t + 1);
            const unsigned ph = (unsigned)((ring_base + (S - 2 - t)) >> 2) & 1u; // (tagged ring) parity of this use of the slot
            float4 v[NLD];
            bool ok = false;
            for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                bool gd = true;
#pragma unroll
                for (int i = 0; i < NLD; i++) v[i] = ld_sc1(rQ, off + i * (4 * 64 * (int)sizeof(float)));
#pragma unroll
                for (int i = 0; i < NLD; i++) gd = gd && (tagged_ ? tag_ready(v[i], ph) : hx_ready(v[i]));
                if (__all(gd)) {
                    ok = true;
                    break;
                }
                if ((spins & 255) == 255 && __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                // (Requesting only the incomplete pieces again makes the retries faster and the recurrence slower, 279 -> 325 us:
                // the more often a line is polled, the later its store gets through.  cfg bits 9-11: pauses between polls.)
                for (int i = 0; i < ((cfg >> 9) & 7); i++) __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) {
                give_up();
                break;
            }
            SSTAMP(8, 1)
            if (tagged_) { // the parity bit is cleared before the sum
#pragma unroll
                for (int i = 0; i < NLD; i++) v[i] = float4{tag_value(v[i].x), tag_value(v[i].y), tag_value(v[i].z), tag_value(v[i].w)};
            }
            // sum over the sources: lane-local over sources 4i + q (ascending i), then over the four rows q of 16 lanes
            float4 sm = v[0];
#pragma unroll
            for (int i = 1; i < NLD; i++) {
                sm.x += v[i].x;
                sm.y += v[i].y;
                sm.z += v[i].z;
                sm.w += v[i].w;
            }
            // 4 x 4 transpose-sum: row q = l >> 4 needs column q.  Round 1 (partner row q ^ 1): keep the columns of q's
            // parity, send the other two; round 2 (partner row q ^ 2): keep column q, send the other.
            const int q = l >> 4;
            const float k0 = (q & 1) ? sm.y : sm.x, k1 = (q & 1) ? sm.w : sm.z;   // columns (q & 1), (q & 1) + 2
            const float s0 = (q & 1) ? sm.x : sm.y, s1 = (q & 1) ? sm.z : sm.w;   // the partner's
            const float z0 = k0 + xchg_row16(s0, l), z1 = k1 + xchg_row16(s1, l);
            const float keep = (q & 2) ? z1 : z0, send = (q & 2) ? z0 : z1;
            dhn = keep + xchg_half32(send, l);
        }
        // (dhy: the output-layer term of this step, picked up from wave 11's buffer a step ago, off the chain)
        if (XCD_LOCAL && t == S - 2) { // every workgroup of the group has published Q_{S-1}, its XCC id before it
            unsigned mine = 0;
            bool same = true;
            if (l < NBK) {
                mine = __hip_atomic_load(xcc_tab + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                same = (mine >> 4) == epoch;
            }
            const unsigned first = __builtin_amdgcn_readfirstlane(mine);
            if (l < NBK) same = same && mine == first;
            local_pub = (XCD_FORCE_LOCAL || __all(same)) && NBK <= 64;
            if (l == 0) __hip_atomic_store(&s_loc[hf], local_pub ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const float dh = dhy + dhn;                         // R/lstm.cc:228
        float dcv = dh * og + dcn;                          // :233
        dcv = dcv * (1.0f - cv * cv);                       // :235
        const float d_o = (dh * cv) * (og * (1.0f - og));   // :238,244
        const float d_i = (dcv * ug) * (ig * (1.0f - ig));  // :239,244
        const float d_f = (dcv * cp) * (fg * (1.0f - fg));  // :240,244
        const float d_u = (dcv * ig) * (1.0f - ug * ug);    // :241,247
        dcn = dcv * fg;                                     // :256
        SSTAMP(8, 2)
        // dg_t for this workgroup's product waves: [column][k = gate*16 + unit], then the count (LDS operations of a wave
        // execute in order: the count lands behind the values)
        {
            float *dp = dgl + (hf * 2 + (t & 1)) * 256 + cc * 64 + jj;
            dp[0] = d_i;
            dp[16] = d_o;
            dp[32] = d_f;
            dp[48] = d_u;
            asm volatile("" ::: "memory");
            if (l == 0) __hip_atomic_fetch_add(&s_done[2 * hf + (t & 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        SSTAMP(8, 3)
        // off the chain: the plain DG the dU product reads after the launch.  4x4 transpose over the four lanes of a quad by
        // DPP: lane (column cc, unit jj = 4*tq + ta) ends up with gate ta of units 4*tq .. 4*tq+3, one 16-byte store
        const int ta = jj & 3, tq = jj >> 2;
        float t0 = d_i, t1 = d_o, t2 = d_f, t3 = d_u;
        {
            const float lo = (ta & 1) ? t0 : t1, hi = (ta & 1) ? t2 : t3;
            const float rlo = dpp_f<0xB1>(lo), rhi = dpp_f<0xB1>(hi); // quad_perm [1,0,3,2]
            if (ta & 1) {
                t0 = rlo;
                t2 = rhi;
            } else {
                t1 = rlo;
                t3 = rhi;
            }
            const float s0 = (ta & 2) ? t0 : t2, s1 = (ta & 2) ? t1 : t3;
            const float r0 = dpp_f<0x4E>(s0), r1 = dpp_f<0x4E>(s1); // quad_perm [2,3,0,1]
            if (ta & 2) {
                t0 = r0;
                t1 = r1;
            } else {
                t2 = r0;
                t3 = r1;
            }
        }
        if (ecol < B) {
            const float4 v = {t0, t1, t2, t3};
            *reinterpret_cast<float4 *>(DG + ((size_t)t * B + ecol) * G4 + ta * N + 16 * kb + 4 * tq) = v;
        }
        SSTAMP(8, 4)
        if (t >= 2) {
            fetch(t - 1);
            // wave 11 runs up to four steps ahead of the chain, so this does not wait in practice
            if (!lds_wait(s_ol, (unsigned)(S - (t - 1)))) {
                give_up();
                break;
            }
            dhy = dhyb[((t - 1) & 3) * 128 + (4 * hf + cc) * 16 + jj];
        }
        if (FUSE) {
            if (ecol < B) { // db += dg, R/lstm.cc:252
                dbacc[0] += d_i;
                dbacc[1] += d_o;
                dbacc[2] += d_f;
                dbacc[3] += d_u;
            }
            // dg_t for the dW wave (four copies deep; s_tab counts the steps that wave has finished)
            if (t + 4 <= S - 1 && !(cfg & 32) && !lds_wait(s_tab, (unsigned)(S - (t + 4)))) {
                give_up();
                break;
            }
            float *sp = stage + (hf * 4 + (t & 3)) * 256 + cc * 64 + jj;
            sp[0] = d_i;
            sp[16] = d_o;
            sp[32] = d_f;
            sp[48] = d_u;
            asm volatile("" ::: "memory");
        }
        // step t done here: its dhy slot is free, its dg copy is in place
        if (l == 0) __hip_atomic_fetch_add(&s_stage[hf], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (FUSE) {
        __syncthreads();
        if (__hip_atomic_load(s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
#pragma unroll
        for (int gt = 0; gt < 4; gt++) dbs[((4 * hf + cc) * 4 + gt) * 16 + jj] = dbacc[gt]; // through `red` (free now)
        __syncthreads();
    }
}
template <int N_, bool FUSE, bool STAMP = false> __global__ __launch_bounds__(BWDH_THREADS) void k_bwd_scatter(const BwdhArgs p) {
    BWDH_COMMON(p)
    if (pin_ng_ && g >= NG) return; // pinned launch: this workgroup sits on an XCD that hosts no group
    if (tid == 0) {
        *s_abort = 0;
        s_done[0] = s_done[1] = s_done[2] = s_done[3] = 0;
        s_stage[0] = s_stage[1] = 0;
        *s_dy = *s_ol = 0;
        *s_tab = 0;
        sync_[12] = sync_[13] = sync_[14] = sync_[15] = 0;
        if (XCD_LOCAL) {
            __hip_atomic_store(xcc_tab + kb, (epoch << 4) | (__builtin_amdgcn_s_getreg(6164) & 15u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // visible before anything this workgroup publishes
        }
    }
    if (FUSE)
        for (int i = tid; i < 257 * 64; i += BWDH_THREADS) dWt[i] = 0.0f;
    __syncthreads();
    if (w < 8) bwds_product<N_, FUSE, STAMP>(p);
    else if (w < 10) bwds_elementwise<N_, FUSE, STAMP>(p);
    else if (w == 11) bwdh_output_layer<N_, FUSE, STAMP>(p);
    else if (FUSE) bwdh_weight_sums<N_, FUSE, STAMP>(p);
}
#undef SSTAMP
#undef HSTAMP

// ------------------------------------------------------------------------------------------------
// backward recurrence, scatter form, bf16 operands (LSTM_HIP_BF16_RECURRENCE; N = 256 / 512 / 1024, 8-column groups): the
// decomposition and ring of k_bwd_scatter with the recurrent product on v_mfma_f32_4x4x4_16b_bf16 -- dg_t rounded to
// bfloat16 (RNE) on its way into LDS, U as a bf16 image (k_pack_U_bf16, Ubwd6b), fp32 accumulation, fp32 partial sums on the
// ring -- which is what the oracle's bf16 recurrence mode computes (oracle/lstm_ref.c, ref_set_bf16_recurrence).  Half the
// weight registers of the fp32 form, so hidden 1024 fits: a workgroup keeps its 64 gate rows for all N outputs in
// N/16 registers per lane.  The bf16 path fuses nothing into the recurrence (DHy, dW, db, dWhy are separate launches),
// so the roles are: waves 0-7 product (wave w: outputs [OW*w, OW*(w+1)), OW = max(64, N/8), in sets of 64), waves 8 / 9
// elementwise of half A / B.  One instruction = 4 values of k for 64 outputs and 4 columns:
//   D[column i][output j of the block] += sum_e dg[k = 4ab + e][column i] * U[k][output]      CBSZ = 4 / ABID = ab
// ------------------------------------------------------------------------------------------------
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
struct BwdsbArgs {
    const uint2 *Ubwd6b;
    float *DG;
    const float *DHy, *G, *C;
    float *Qx;
    unsigned *cnt, *abortp;
    unsigned epoch;
    int ring_base, S, B, NG, pinned, col0, gcols; // columns col0 .. col0 + gcols*NG - 1 of the B
    unsigned short *DGt_b;                          // or null: transposed bf16 image of dg, [4N][Tpad]
    int Tpad;
    unsigned long long *stamps;
};
// UW = units per workgroup: 16, or 32 where a group of N/16 workgroups would not fit one XCD (hidden 1024: 32 workgroups of
// 128 gate rows each, a quarter of the chip per 8 columns, but every hand-off stays inside one L2 -- measured at hidden 512,
// 64 streams: 182 us with XCD-local groups, 292 us with the same groups spread over the XCDs)
template <int N_, int UW> struct BwdsbShape {
    static constexpr int N = N_, NB = N / UW, KK = 4 * UW, NAB = KK / 4, NR = NAB / 16;
    static constexpr int NS = N >= 512 ? N / 512 : 1, NPW = N / (64 * NS);
    static constexpr int NEH = UW / 16, THREADS = (8 + 2 * NEH) * 64;
};
__host__ __device__ inline size_t bwdsb_ring_floats(int N, int UW, int B) {
    const size_t nb = (size_t)N / UW;
    return (size_t)HX_RING * ((B + 7) / 8) * 2 * nb * nb * (4 * UW);
}
// Hand-off without a reset store (BWDSB_TAGGED): a published partial sum carries the parity of its slot's use count in its last
// mantissa bit (the consumer clears it again: a partial sum, whose operands were rounded to bf16, loses its last bit), so a consumer
// tells this use of the slot from the previous one by that bit, in all four words of a 16-byte piece, and nobody has to write
// the sentinel back: half the ring's write traffic.  The ring starts as all ones (parity 1), the first use publishes parity 0.
#ifndef BWDSB_TAGGED
#define BWDSB_TAGGED 1
#endif
#ifndef BF16_SINGLE_HALF_DEFAULT
#define BF16_SINGLE_HALF_DEFAULT 1
#endif
__device__ __forceinline__ float bwdsb_mark(float v, unsigned phase) {
    if (!BWDSB_TAGGED) return hx_canon(v);
    return __uint_as_float((__float_as_uint(v) & ~1u) | phase);
}
__device__ __forceinline__ float bwdsb_value(float v) { return BWDSB_TAGGED ? __uint_as_float(__float_as_uint(v) & ~1u) : v; }
__device__ __forceinline__ bool bwdsb_ready(const float4 &v, unsigned phase) {
    if (!BWDSB_TAGGED) return hx_ready(v);
    return ((__float_as_uint(v.x) & __float_as_uint(v.y) & __float_as_uint(v.z) & __float_as_uint(v.w) & 1u) == phase) &&
           (((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) & 1u) == phase);
}
template <int N_, int UW, bool STAMP = false>
__global__ __launch_bounds__((BwdsbShape<N_, UW>::THREADS)) void k_bwd_scatter_bf16(const BwdsbArgs p) {
    using Sh = BwdsbShape<N_, UW>;
    constexpr int N = N_, G4 = 4 * N, NB = Sh::NB, NLD = NB / 4, KK = Sh::KK, NR = Sh::NR, NS = Sh::NS, NPW = Sh::NPW, NEH = Sh::NEH;
    static_assert((N == 256 || N == 512 || N == 1024) && (UW == 16 || UW == 32) && NB % 4 == 0, "bf16 scatter form: shapes");
    __shared__ __attribute__((aligned(16))) unsigned short dgl[2][2][4 * KK]; // [half][step parity][column][k]: bf16 dg_t
    __shared__ unsigned s_done[2][2], s_loc[2];
    __shared__ int s_abort;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int NG = p.NG, GC = p.gcols; // GC = 8: two halves of four columns per workgroup; 4: one (half B's waves leave)
    int kb, g;
    if (p.pinned) { // 8 * NB workgroups launched: workgroup i runs on XCD i % 8; group g lives on XCD g, the rest leave
        g = (int)blockIdx.x & 7, kb = (int)blockIdx.x >> 3;
        if (g >= NG) return;
    } else {
        kb = (int)blockIdx.x / NG, g = (int)blockIdx.x % NG;
    }
    const int S = p.S, B = p.B, ring_base = p.ring_base;
    float *Qx = p.Qx;
    // diagnostics (LSTM_HIP_DEBUG_STAMPS): lane 0 of waves 3 and 8 of workgroups (0, 0) and (NB/2, 0); slots as in k_bwd_scatter
    unsigned long long *stq = STAMP && g == 0 && (kb == 0 || kb == NB / 2) && l == 0 && (w == 3 || w == 8) ? p.stamps + (size_t)(kb ? 1 : 0) * S * 16 : nullptr;
#define HSTAMPQ(k) if (STAMP && stq) stq[(size_t)t * 16 + (k)] = __builtin_amdgcn_s_memtime();
    // the ring has a region per group of the WHOLE batch (gg): a region is written once per window, by the launch that owns its
    // columns, so every region sees the same sequence of slots and phases whatever the batch is split into
    // (regions are counted in halves: region of (group, half) = its first column / 4)
    const int NRG = 2 * ((B + 7) / 8), rg0 = (p.col0 + GC * g) / 4;
    const __amdgpu_buffer_rsrc_t rQ = make_rsrc(Qx, bwdsb_ring_floats(N, UW, B) * sizeof(float));
    unsigned *xcc_tab = p.cnt + (size_t)g * CNT_SLOTS * CNT_STRIDE;
    if (tid == 0) {
        s_abort = 0;
        s_done[0][0] = s_done[0][1] = s_done[1][0] = s_done[1][1] = 0;
        s_loc[0] = s_loc[1] = 0;
        if (XCD_LOCAL) {
            __hip_atomic_store(xcc_tab + kb, (p.epoch << 4) | (__builtin_amdgcn_s_getreg(6164) & 15u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // visible before anything this workgroup publishes
        }
    }
    __syncthreads();
    auto give_up = [&]() {
        if (l == 0) {
            __hip_atomic_store(p.abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&s_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto lds_wait = [&](unsigned *word, unsigned want) -> bool {
        for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) {
                asm volatile("" ::: "memory");
                return true;
            }
            if (__hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        return false;
    };
    if (w < 8) {
        // ---------------- product waves ----------------
        __builtin_amdgcn_s_setprio(2);
        const bool active = w < NPW;
        const int lb = l >> 2, lj = l & 3;
        uint2 a[NS][NR][16]; // bf16 x 4: U[row(k = 4(16r + ab) + e)][output] for e = 0..3
#pragma unroll
        for (int sx = 0; sx < NS; sx++)
#pragma unroll
            for (int r = 0; r < NR; r++)
#pragma unroll
                for (int ab = 0; ab < 16; ab++)
                    a[sx][r][ab] = active ? p.Ubwd6b[(((((size_t)kb * NPW + w) * NS + sx) * NR + r) * 16 + ab) * 64 + l] : uint2{0u, 0u};
#pragma unroll
        for (int sx = 0; sx < NS; sx++)
#pragma unroll
            for (int r = 0; r < NR; r++)
#pragma unroll
                for (int ab = 0; ab < 16; ab++) asm volatile("" ::"v"(a[sx][r][ab].x), "v"(a[sx][r][ab].y)); // complete before the loop
        for (int t = S - 1; t >= 2; t--) {
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                if (hf == 1 && GC == 4) continue;
                if (hf == 0) { HSTAMPQ(8) }
                if (!lds_wait(&s_done[hf][t & 1], (unsigned)(NEH * ((S - 1 - t) / 2 + 1)))) {
                    give_up();
                    return;
                }
                if (!active) continue;
                if (hf == 0) { HSTAMPQ(9) } else { HSTAMPQ(5) }
                bf16x4_t av[NR]; // column lj, k = 4*(16r + lb) .. +3
#pragma unroll
                for (int r = 0; r < NR; r++)
                    av[r] = __builtin_bit_cast(bf16x4_t, *reinterpret_cast<const uint2 *>(&dgl[hf][t & 1][lj * KK + 4 * (16 * r + lb)]));
                const bool local = t < S - 1 && __hip_atomic_load(&s_loc[hf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's older stores (the last reset) are complete
                const int seq = ring_base + (S - 1 - t); // publication number: slot = seq & 3, phase = bit 2
                const int spub = seq & (HX_RING - 1), srst = (seq + 2) & (HX_RING - 1);
                const unsigned phase = (unsigned)(seq >> 2) & 1u;
#pragma unroll
                for (int sx = 0; sx < NS; sx++) {
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#define SB4(r, ab)                                                                                                           \
    c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av[r], __builtin_bit_cast(bf16x4_t, a[sx][r][ab + 0]), c0, 4, ab + 0, 0);    \
    c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av[r], __builtin_bit_cast(bf16x4_t, a[sx][r][ab + 1]), c1, 4, ab + 1, 0);    \
    c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av[r], __builtin_bit_cast(bf16x4_t, a[sx][r][ab + 2]), c2, 4, ab + 2, 0);    \
    c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av[r], __builtin_bit_cast(bf16x4_t, a[sx][r][ab + 3]), c3, 4, ab + 3, 0);
#pragma unroll
                    for (int r = 0; r < NR; r++) { SB4(r, 0) SB4(r, 4) SB4(r, 8) SB4(r, 12) }
#undef SB4
                    if (sx == NS - 1) {
                        if (hf == 0) { HSTAMPQ(10) } else { HSTAMPQ(6) }
                    }
                    float4 q;
                    q.x = bwdsb_mark((c0[0] + c1[0]) + (c2[0] + c3[0]), phase);
                    q.y = bwdsb_mark((c0[1] + c1[1]) + (c2[1] + c3[1]), phase);
                    q.z = bwdsb_mark((c0[2] + c1[2]) + (c2[2] + c3[2]), phase);
                    q.w = bwdsb_mark((c0[3] + c1[3]) + (c2[3] + c3[3]), phase);
                    const float sv = __uint_as_float(HX_SENT);
                    const float4 sent = {sv, sv, sv, sv};
                    const int out = (w * NS + sx) * 64 + l, d = out / UW, u = out % UW;
                    const int e_pub = ((((spub * NRG + rg0 + hf)) * NB + d) * NB + kb) * KK + u * 4;
                    const int e_rst = ((((srst * NRG + rg0 + hf)) * NB + d) * NB + kb) * KK + u * 4;
                    if (XCD_LOCAL && local) {
                        *reinterpret_cast<float4 *>(Qx + e_pub) = q;
                        if (!BWDSB_TAGGED) *reinterpret_cast<float4 *>(Qx + e_rst) = sent;
                    } else {
                        st_sc1(q, rQ, e_pub * (int)sizeof(float));
                        if (!BWDSB_TAGGED) st_sc1(sent, rQ, e_rst * (int)sizeof(float));
                    }
                    if (sx == NS - 1) {
                        if (hf == 0) { HSTAMPQ(11) } else { HSTAMPQ(7) }
                    }
                }
            }
        }
    } else {
        // ---------------- elementwise waves: NEH per half (16 units each); lane = column*16 + unit ----------------
        const int hf = (w - 8) / NEH, uh = (w - 8) % NEH;
        if (hf == 1 && GC == 4) return;
        __builtin_amdgcn_s_setprio(3);
        const int cc = l >> 4, jj = l & 15;
        const int ecol = p.col0 + GC * g + 4 * hf + cc, ecolc = ecol < B ? ecol : B - 1;
        const int uu = 16 * uh + jj, j = UW * kb + uu;
        float dcn = 0.0f; // dcnext, R/lstm.cc:217
        bool local_pub = false;
        float ig, og, fg, ug, cv, cp, dhy;
        auto fetch = [&](int tu) {
            const float *gc = p.G + ((size_t)tu * B + ecolc) * G4 + j;
            ig = gc[0], og = gc[N], fg = gc[2 * N], ug = gc[3 * N];
            cv = p.C[((size_t)tu * B + ecolc) * N + j], cp = p.C[((size_t)(tu - 1) * B + ecolc) * N + j];
            dhy = p.DHy[((size_t)tu * B + ecolc) * N + j]; // Why^T dy_tu, a launch of its own in the bf16 path
        };
        fetch(S - 1);
        for (int t = S - 1; t >= 1; t--) {
            HSTAMPQ(0)
            float dhn = 0.0f;
            if (t < S - 1) {
                // Q_{t+1}: this lane's piece = sources 4i + (l >> 4), unit 16*uh + (l & 15), the four columns
                const int seq = ring_base + (S - 2 - t); // publication number of Q_{t+1}
                const size_t slot = (size_t)(seq & (HX_RING - 1));
                const unsigned phase = (unsigned)(seq >> 2) & 1u;
                const int off = (int)((((((slot * NRG + rg0 + hf)) * NB + kb) * NB + (size_t)(l >> 4)) * KK + (size_t)(16 * uh + (l & 15)) * 4) * sizeof(float));
                float4 v[NLD];
                bool ok = false;
                // (A one-lane hint poll ahead of this loop, or pauses between the polls: 275 -> 270-272 us at hidden 1024, nothing at 512.)
                for (int spins = 0; spins <= SPIN_LIMIT; spins++) {
                    bool gd = true;
#pragma unroll
                    for (int i = 0; i < NLD; i++) v[i] = ld_sc1(rQ, off + i * (4 * KK * (int)sizeof(float)));
#pragma unroll
                    for (int i = 0; i < NLD; i++) gd = gd && bwdsb_ready(v[i], phase);
                    if (__all(gd)) {
                        ok = true;
                        break;
                    }
                    if ((spins & 255) == 255 && __hip_atomic_load(p.abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                }
                if (!ok) {
                    give_up();
                    return;
                }
                // the phase bit is cleared before the sum: what is added does not depend on which use of the slot this is
                float4 sm = {bwdsb_value(v[0].x), bwdsb_value(v[0].y), bwdsb_value(v[0].z), bwdsb_value(v[0].w)};
#pragma unroll
                for (int i = 1; i < NLD; i++) {
                    sm.x += bwdsb_value(v[i].x);
                    sm.y += bwdsb_value(v[i].y);
                    sm.z += bwdsb_value(v[i].z);
                    sm.w += bwdsb_value(v[i].w);
                }
                const int q = l >> 4;
                const float k0 = (q & 1) ? sm.y : sm.x, k1 = (q & 1) ? sm.w : sm.z;
                const float s0 = (q & 1) ? sm.x : sm.y, s1 = (q & 1) ? sm.z : sm.w;
                const float z0 = k0 + xchg_row16(s0, l), z1 = k1 + xchg_row16(s1, l);
                const float keep = (q & 2) ? z1 : z0, send = (q & 2) ? z0 : z1;
                dhn = keep + xchg_half32(send, l);
            }
            HSTAMPQ(1)
            if (XCD_LOCAL && t == S - 2) { // every workgroup of the group has published Q_{S-1}, its XCC id before it
                unsigned mine = 0;
                bool same = true;
                if (l < NB) {
                    mine = __hip_atomic_load(xcc_tab + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    same = (mine >> 4) == p.epoch;
                }
                const unsigned first = __builtin_amdgcn_readfirstlane(mine);
                if (l < NB) same = same && mine == first;
                local_pub = XCD_FORCE_LOCAL || __all(same);
                if (l == 0 && uh == 0)
                    __hip_atomic_store(&s_loc[hf], local_pub ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const float dh = dhy + dhn;                         // R/lstm.cc:228
            float dcv = dh * og + dcn;                          // :233
            dcv = dcv * (1.0f - cv * cv);                       // :235
            const float d_o = (dh * cv) * (og * (1.0f - og));   // :238,244
            const float d_i = (dcv * ug) * (ig * (1.0f - ig));  // :239,244
            const float d_f = (dcv * cp) * (fg * (1.0f - fg));  // :240,244
            const float d_u = (dcv * ig) * (1.0f - ug * ug);    // :241,247
            dcn = dcv * fg;                                     // :256
            HSTAMPQ(2)
            {   // dg_t, rounded to bfloat16, for this workgroup's product waves: [column][k = gate*UW + unit]
                unsigned short *dp = &dgl[hf][t & 1][cc * KK + uu];
                dp[0] = __builtin_bit_cast(unsigned short, (__bf16)d_i);
                dp[UW] = __builtin_bit_cast(unsigned short, (__bf16)d_o);
                dp[2 * UW] = __builtin_bit_cast(unsigned short, (__bf16)d_f);
                dp[3 * UW] = __builtin_bit_cast(unsigned short, (__bf16)d_u);
                asm volatile("" ::: "memory");
                if (l == 0) __hip_atomic_fetch_add(&s_done[hf][t & 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            HSTAMPQ(3)
            // off the chain: the plain fp32 DG the dU / dW / db launches read afterwards (4x4 DPP transpose, one 16-byte store)
            const int ta = jj & 3, tq = jj >> 2;
            float t0 = d_i, t1 = d_o, t2 = d_f, t3 = d_u;
            {
                const float lo = (ta & 1) ? t0 : t1, hi = (ta & 1) ? t2 : t3;
                const float rlo = dpp_f<0xB1>(lo), rhi = dpp_f<0xB1>(hi); // quad_perm [1,0,3,2]
                if (ta & 1) {
                    t0 = rlo;
                    t2 = rhi;
                } else {
                    t1 = rlo;
                    t3 = rhi;
                }
                const float s0 = (ta & 2) ? t0 : t2, s1 = (ta & 2) ? t1 : t3;
                const float r0 = dpp_f<0x4E>(s0), r1 = dpp_f<0x4E>(s1); // quad_perm [2,3,0,1]
                if (ta & 2) {
                    t0 = r0;
                    t1 = r1;
                } else {
                    t2 = r0;
                    t3 = r1;
                }
            }
            if (ecol < B && p.DGt_b != nullptr) { // the k-contiguous bf16 image the dU product reads: row = gate row, column (t-1)*B + stream
                unsigned short *q = p.DGt_b + (size_t)j * p.Tpad + (size_t)(t - 1) * B + ecol;
                q[0] = __builtin_bit_cast(unsigned short, (__bf16)d_i);
                q[(size_t)N * p.Tpad] = __builtin_bit_cast(unsigned short, (__bf16)d_o);
                q[(size_t)2 * N * p.Tpad] = __builtin_bit_cast(unsigned short, (__bf16)d_f);
                q[(size_t)3 * N * p.Tpad] = __builtin_bit_cast(unsigned short, (__bf16)d_u);
            }
            if (ecol < B) {
                const float4 v = {t0, t1, t2, t3};
                *reinterpret_cast<float4 *>(p.DG + ((size_t)t * B + ecol) * G4 + ta * N + UW * kb + 16 * uh + 4 * tq) = v;
            }
            HSTAMPQ(4)
            if (t >= 2) fetch(t - 1);
        }
    }
#undef HSTAMPQ
}

// ------------------------------------------------------------------------------------------------
// backward recurrence, t = S-1..1, N = 32*NR4W.  grid (N/16, ceil(B/COLS)), 512 threads.
// Workgroup (kb, g) owns hidden units 16kb..16kb+15 for column group g (COLS = 8 or 16 batch columns): the tile
// dhnext = U^T * dg[t+1] (R/lstm.cc:255) with K = 4N split over its 8 waves (U^T fragments in VGPRs) -- 16x16x4 MFMA
// tiles, or for 8-column fp32 groups (M4) v_mfma_f32_4x4x1 blocks with operand broadcast, which leave no tile column
// empty -- then one thread per (unit, column) does R/lstm.cc:228-247,256 and dg[t] is published.
// ------------------------------------------------------------------------------------------------
// STAMP builds (LSTM_HIP_DEBUG_STAMPS, N = 512): s_memtime at the points marked BSTAMP, for two workgroups, into a buffer
// nothing else reads.  Slots [workgroup 0 | gridDim.x/2][t][16]: wave 0 (MFMA + elementwise) 0-7, wave 3 (MFMA, then the
// dW table) 8-12, wave 5 (MFMA, then output-layer follower) 13-15.
#define BSTAMP(wave, k)                                                                                        \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = __builtin_amdgcn_s_memtime();
#define BSTAMP_VAL(wave, k, v)                                                                                 \
    if (STAMP && l == 0 && w == (wave) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && blockIdx.y == 0) \
        stamps[((size_t)(blockIdx.x ? 1 : 0) * S + t) * 16 + (k)] = (unsigned long long)(v);
// FUSE: the input-side weight-gradient sums (R/lstm.cc:251-252) are accumulated by the same workgroups:
//   dW[rows, x] += dg_t[rows, col] for the column's input byte x   a [257][64] LDS table kept by one wave that
//                                                          has no part in the elementwise / publish phase
//   db[rows]    += dg_t[rows, col]                         registers of the elementwise threads
// The dW update for step t+1 runs during step t (dg is double-buffered in LDS), after the step's second
// workgroup barrier; from there on the elementwise waves do not synchronise with anything (each transposes its own
// gates by DPP, stores, drains and arrives on the counter for itself), so neither the updater nor the followers hold
// up the store / drain / signal / poll path.
//   DHy_t = Why^T*dy_t (R/lstm.cc:228) for this workgroup's 16 units, one step ahead of its use, and
//   dWhy[:, units] += dy_t * h_t[units]^T (R/lstm.cc:226)    four "follower" waves (4..7), 24 MFMAs a step
// Each column group g leaves one partial block [dW | (dU, unused) | db | dWhy] (the layout of the flat
// gradient block) in gpart[g]; gemm_fold adds the groups in order afterwards.
// (Accumulating dU the same way -- 4 x N/16 MFMA accumulator tiles per workgroup -- was built and
// measured: it needs ~100 more VGPRs, spills, and cost more than the separate GEMM it replaced.)
template <int NR4W, int COLS, bool FUSE, bool STAMP = false, bool BF16 = false, bool M4 = false>
__global__ __launch_bounds__(512, 2) void k_bwd_persistent(const float4 *__restrict__ Ubwd, float *DG,
                                                           const float *__restrict__ DHy, const float *__restrict__ G,
                                                           const float *__restrict__ C, const float *__restrict__ H,
                                                           const int32_t *__restrict__ xi, float *__restrict__ gpart,
                                                           const float *__restrict__ Why, const float *__restrict__ dY,
                                                           unsigned *cnt, unsigned *abortp, unsigned epoch, int S, int B,
                                                           int spread, unsigned long long *stamps = nullptr,
                                                           unsigned short *DGb = nullptr) {
    static_assert(!STAMP || M4, "stamped builds exist for the fp32 4x4x1 form");
    constexpr int N = 32 * NR4W, G4 = 4 * N, nr4 = N / 4;
    // BF16: Ubwd holds the bf16 image (N/64 16-byte fragments per wave), dg_{t+1} is read from the bf16 copy DGb
    // (the hand-off), and dg_t is published to DGb (sc1) as well as to DG (fp32, plain, for the dU product)
    constexpr int NRS = BF16 ? NR4W / 2 : NR4W; // A/B fragments per wave
    constexpr int ETH = 16 * COLS;        // threads with an elementwise / store role
    constexpr int EW = ETH / 64;          // ... i.e. waves 0..EW-1
    extern __shared__ __attribute__((aligned(16))) float dWt[]; // FUSE: [257][64] per-input-byte sums of this WG's rows
    __shared__ float red[8 * 4 * 64];
    __shared__ __attribute__((aligned(16))) float stage[2][16 * 4 * 16]; // dg of this WG, double-buffered by step parity
    __shared__ int s_abort;
    // FUSE: output-layer followers (waves 4..7): DHy_t = Why^T * dy_t for this workgroup's units, one step
    // ahead of its use, and the dWhy columns of those units
    __shared__ float olred[4 * 4 * 64];
    __shared__ float dhyb[2][16 * 16];
    __shared__ unsigned s_ol;
    __shared__ int s_local;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int NBK = gridDim.x, NG = gridDim.y;
    const int lin_ = blockIdx.x + NBK * blockIdx.y;
    // `spread` (LSTM_HIP_BWD_SPREAD=1, tests): keep the dispatch-order mapping, which spreads every column group over
    // all XCDs -- the placement the XCD-local hand-off below must detect and decline
    const bool remap = GROUP_REMAP && !spread;
    const int kb = remap ? lin_ / NG : (int)blockIdx.x, g = remap ? lin_ % NG : (int)blockIdx.y;
    const int q = l >> 4;
    // COLS = 16: the MFMA tile is full.  COLS = 8: lanes 8..15 of the B operand are never used (their tile
    // columns are discarded) -- the matrix pipe is not what bounds a step, and 8-column groups put
    // twice as many CUs to work on half the dg_{t+1} bytes each.
    const int mcol = COLS * g + (l & (COLS - 1)), mcolc = mcol < B ? mcol : B - 1; // MFMA B-operand column
    // elementwise role (threads 0..ETH-1): unit jj, column cc
    const int jj = tid & 15, cc = (tid >> 4) & (COLS - 1);
    const int ecol = COLS * g + cc, ecolc = ecol < B ? ecol : B - 1;
    const int j = 16 * kb + jj;

    float4 a[NRS];
#pragma unroll
    for (int i = 0; i < NRS; i++)
        a[i] = BF16 ? Ubwd[((size_t)kb * (G4 / 32) + w * NRS + i) * 64 + l]
               : M4 ? Ubwd[(((size_t)kb * 8 + w) * NR4W + i) * 64 + l] // the 4x4x1 image: see k_pack_U
                    : Ubwd[((size_t)kb * nr4 + w * NR4W + i) * 64 + l];
    const __amdgpu_buffer_rsrc_t rDG = BF16 ? make_rsrc(DGb, (size_t)S * G4 * B * sizeof(unsigned short))
                                            : make_rsrc(DG, (size_t)S * G4 * B * sizeof(float));
    float dcn = 0.0f; // dcnext, R/lstm.cc:217
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    if (FUSE)
        for (int i = tid; i < 257 * 64; i += 512) dWt[i] = 0.0f;
    if (tid == 0) {
        s_abort = 0;
        s_ol = 0;
        s_local = 0;
    }
    // XCD-local hand-off (speed only, checked every launch): when all workgroups of this column group run on ONE XCD, dg
    // can be published with plain stores -- the lines stay in that XCD's L2, which serves the group's sc1 loads directly
    // (406 -> 391 us) -- instead of sc1 write-through stores that every consumer pulls back over the fabric.  Placement is
    // observed, not promised, so each workgroup publishes its HW_REG_XCC_ID (sc1, before its first arrival) in the unused
    // step-0 counter slots of its group; after the first wait every workgroup of the group reads the same NBK words
    // and takes the plain-store path only if they all agree.  The first publish is always sc1.
    unsigned *xcc_tab = cnt + (size_t)g * CNT_SLOTS * CNT_STRIDE; // step index 0 is never a hand-off step (t = 1..S-1)
    if (XCD_LOCAL && tid == 0) {
        __hip_atomic_store(xcc_tab + kb, (epoch << 4) | (__builtin_amdgcn_s_getreg(6164) & 15u), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT); // getreg(id 20 = XCC_ID, offset 0, size 4)
    }
    unsigned ol_target = 0;
    bool local_pub = false;
    // Why^T A-fragments of the follower waves: wave ow = w-4 takes output rows m in [64*ow, 64*ow+64);
    // fragment i of k-step ks4 is Why[m = 64*ow + 16*ks4 + 4*(l>>4) + i][16*kb + (l&15)]
    float4 wa[FUSE ? 4 : 1];
    f32x4 yacc[FUSE ? 4 : 1]; // dWhy tiles: rows m in [16*(4*ow+mt), +16), columns 16*kb..+15
    if (FUSE) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++) yacc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (w >= 4) {
            const int ow = w - 4;
#pragma unroll
            for (int ks4 = 0; ks4 < 4; ks4++) {
                const int m = 64 * ow + 16 * ks4 + 4 * q;
                const float *wp = Why + (size_t)(16 * kb + (l & 15)) * 256 + m;
                wa[ks4] = *reinterpret_cast<const float4 *>(wp);
            }
        }
    }
    __syncthreads();

    // dW[:, x] += dg[:, col] (R/lstm.cc:251) for the step whose dg sits in stage[par]: wave EW, one thread per
    // row, walking the columns in order (deterministic); tu is that step
    auto update = [&](int par, int tu) {
        if (w == EW) {
            int xs[COLS];
#pragma unroll
            for (int c = 0; c < COLS; c++) {
                const int col = COLS * g + c;
                const int x = col < B ? xi[(size_t)tu * B + col] : -2;
                xs[c] = x == -1 ? 256 : x; // -1: empty input column -> bucket 256; -2: padding column, skipped
            }
            const float *sg_ = stage[par];
            const int gt = l >> 4, rj = l & 15;
#pragma unroll
            for (int c = 0; c < COLS; c++) // (LDS float atomics instead of read-add-write: 481 -> 523 us)
                if (xs[c] >= 0) dWt[xs[c] * 64 + l] += sg_[(c * 4 + gt) * 16 + rj];
        }
    };

    // Output-layer work of step tu by the follower waves (R/lstm.cc:226,228):
    //   dhyb[tu&1][c][unit] = sum_m Why[m][unit] * dy_tu[m][c]            (K = 256 split over the 4 waves)
    //   dWhy[m][unit]      += sum_c dy_tu[m][c] * h_tu[unit][c]           (K = COLS)
    auto ol_sync = [&]() { // the four follower waves meet (LDS counter); bounded like every other spin of the kernel:
        ol_target += 4;    // a follower that never arrives sets the abort word and the launch ends at its next barrier
        if (l == 0) __hip_atomic_fetch_add(&s_ol, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        int spins = 0;
        while (__hip_atomic_load(&s_ol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < ol_target) {
            if (++spins > SPIN_LIMIT) {
                if (l == 0) {
                    __hip_atomic_store(abortp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                }
                break;
            }
        }
        asm volatile("" ::: "memory");
    };
    // operands are fetched a whole chain phase ahead of their use (the followers must not arrive late at
    // the workgroup barrier that opens the next step)
    float4 ol_dy[FUSE ? 4 : 1];
    float ol_av[FUSE ? COLS / 4 : 1][FUSE ? 4 : 1], ol_hv[FUSE ? COLS / 4 : 1];
    auto output_layer_fetch = [&](int tu) {
        const int ow = w - 4;
        const int c = l & 15, col = COLS * g + c;
        const bool cvalid = c < COLS && col < B;
        const float *dyc = dY + ((size_t)(tu - 1) * B + (cvalid ? col : 0)) * 256; // dY holds steps 1.. at column (t-1)*B+b
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ks4++) {
            ol_dy[ks4] = float4{0.f, 0.f, 0.f, 0.f};
            if (cvalid) ol_dy[ks4] = *reinterpret_cast<const float4 *>(dyc + 64 * ow + 16 * ks4 + 4 * q);
        }
#pragma unroll
        for (int ks = 0; ks < COLS / 4; ks++) {
            const int kc = COLS * g + 4 * ks + q;
            const bool kvalid = kc < B;
            ol_hv[ks] = kvalid ? H[((size_t)tu * B + kc) * N + 16 * kb + (l & 15)] : 0.0f;
            const float *dyk = dY + ((size_t)(tu - 1) * B + (kvalid ? kc : 0)) * 256 + 64 * ow + (l & 15);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) ol_av[ks][mt] = kvalid ? dyk[16 * mt] : 0.0f;
        }
    };
    auto output_layer = [&](int tu) {
        const int ow = w - 4;
        const int c = l & 15;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ks4++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks4].x, ol_dy[ks4].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks4].y, ol_dy[ks4].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks4].z, ol_dy[ks4].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ks4].w, ol_dy[ks4].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) olred[(ow * 4 + r) * 64 + l] = acc[r];
        // dWhy: A[i = m in tile][k = column], B[k = column][j = unit]
#pragma unroll
        for (int ks = 0; ks < COLS / 4; ks++)
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
                yacc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ol_av[ks][mt], ol_hv[ks], yacc[mt], 0, 0, 0);
        ol_sync();
        // fold the four K-quarters: D[row = unit = 4*(lane>>4)+reg][col = c = lane&15]
        if (ow == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = ((olred[(0 * 4 + r) * 64 + l] + olred[(1 * 4 + r) * 64 + l]) + olred[(2 * 4 + r) * 64 + l]) +
                                olred[(3 * 4 + r) * 64 + l];
                dhyb[tu & 1][c * 16 + 4 * q + r] = v;
            }
        }
        ol_sync(); // olred may be rewritten
    };
    if (FUSE) {
        if (w >= 4) {
            output_layer_fetch(S - 1);
            output_layer(S - 1);
        }
        __syncthreads();
    }

    for (int t = S - 1; t >= 1; t--) {
        BSTAMP(0, 0) BSTAMP(3, 8) BSTAMP(5, 13)
        const int cur = t & 1;
        const bool has_next = t < S - 1;
        // operands that do not depend on the chain: fetch them first
        float ig = 0.f, og = 0.f, fg = 0.f, ug = 0.f, cv = 0.f, cp = 0.f, dhy = 0.f;
        if (tid < ETH) {
            const float *gc = G + ((size_t)t * B + ecolc) * G4 + j;
            ig = gc[0];
            og = gc[N];
            fg = gc[2 * N];
            ug = gc[3 * N];
            cv = C[((size_t)t * B + ecolc) * N + j];
            cp = C[((size_t)(t - 1) * B + ecolc) * N + j];
            if (!FUSE) dhy = DHy[((size_t)t * B + ecolc) * N + j];
        }
        if (FUSE && w >= 4 && t >= 2) output_layer_fetch(t - 1);
        if (has_next && w == 0) {
            const unsigned *cpn = cnt + (size_t)((t + 1) * NG + g) * CNT_SLOTS * CNT_STRIDE;
            if (!wait_arrivals<BWD_SH>(cpn, NBK, epoch * EW, abortp, l) && l == 0) s_abort = 1;
            if (XCD_LOCAL && t == S - 2) { // every workgroup of the group has published its XCC id by now
                bool same = true;
                unsigned mine = 0;
                for (int i = l; i < NBK; i += 64) {
                    const unsigned v = __hip_atomic_load(xcc_tab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    mine = v;
                    same = same && (v >> 4) == epoch; // written in this launch
                }
                const unsigned first = __builtin_amdgcn_readfirstlane(mine);
                for (int i = l; i < NBK; i += 64) same = same && mine == first;
                if (NBK > 64) same = false; // one load per lane covers the grids in use; larger ones keep the sc1 path
                if ((XCD_FORCE_LOCAL || __all(same)) && l == 0) s_local = 1;
            }
        }
        __syncthreads();
        if (s_abort) return;
        if (XCD_LOCAL && t == S - 2) local_pub = s_local != 0; // decided by wave 0 just above; kept in a register
        BSTAMP(0, 1) BSTAMP(3, 9)
        float *redp = red;

        float dhn = 0.0f;
        if (M4 && has_next) {
            // 8-column groups on v_mfma_f32_4x4x1 (16 independent 4x4 outer products per instruction) instead of half-empty
            // 16x16x4 tiles: block = 8x + 4y + z (lane = 4*block + i) computes, for gate-row k(y),
            //   D[i][j] += dg[k][column 4x+i] * U^T[unit 4z+j][k]
            // CBSZ = 2 / ABID = z' makes the four z-blocks of a group read their dg from block z' of the loaded register,
            // BLGP = 1 / 2 makes both x-halves read the weights from lanes 0-31 / 32-63 of the weight register: one
            // loaded dg register feeds four instructions, one weight register two, and no lane carries padding
            // (semantics checked on gfx950 by tools/probes/mfma4x4_probe.hip).  Per wave: NR4W/2 16-byte loads with all
            // lanes active (half the instructions of the tile form) and 8*NR4W MFMAs of 8 cycles (half the pipe time).
            constexpr int NL = NR4W / 2, Kw = 16 * NR4W;
            const int lx = l >> 5, ly = (l >> 4) & 1, lz = (l >> 2) & 3, li = l & 3;
            const int col4 = COLS * g + 4 * lx + li, col4c = col4 < B ? col4 : B - 1;
            const int off = (int)((((size_t)(t + 1) * B + col4c) * G4 + Kw * w + 4 * (2 * lz + ly)) * sizeof(float));
            constexpr int PF = BWD_PF < NL ? BWD_PF : NL;
            float4 b[NL];
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#define M4_STEP(av, wq, blgp)                                                          \
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.x, c0, 2, 0, blgp);                 \
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.y, c1, 2, 1, blgp);                 \
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.z, c2, 2, 2, blgp);                 \
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(av, wq.w, c3, 2, 3, blgp);
#pragma unroll
            for (int i = 0; i < PF; i++) b[i] = ld_sc1(rDG, off + 128 * i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                if (i + PF < NL) b[i + PF] = ld_sc1(rDG, off + 128 * (i + PF));
                M4_STEP(b[i].x, a[2 * i], 1)
                M4_STEP(b[i].y, a[2 * i], 2)
                M4_STEP(b[i].z, a[2 * i + 1], 1)
                M4_STEP(b[i].w, a[2 * i + 1], 2)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef M4_STEP
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float v = (c0[r] + c1[r]) + (c2[r] + c3[r]);
                // the two k-parities y sit 16 lanes apart: fold them (ds_swizzle-free: one bpermute per register)
                redp[(w * 4 + r) * 64 + l] = v + __shfl_xor(v, 16, 64);
            }
            BSTAMP(0, 3) BSTAMP(3, 11) BSTAMP(5, 14)
            __syncthreads();
            BSTAMP(0, 4) BSTAMP(3, 12) BSTAMP(5, 15)
        } else if (has_next) {
            // Software pipeline with PF fragment loads in flight ahead of the MFMAs.  Left to itself the
            // scheduler keeps only two in flight (a fabric round trip per pair of loads: measured +220
            // cycles per load), and all of them at once measured slower still; sched_barriers pin the order.
            const int off = BF16 ? (int)((((size_t)(t + 1) * B + mcolc) * G4 + 32 * (w * NRS) + 8 * q) * sizeof(unsigned short))
                                 : (int)((((size_t)(t + 1) * B + mcolc) * G4 + 16 * (w * NR4W) + 4 * q) * sizeof(float));
            constexpr int PF = BWD_PF < NRS ? BWD_PF : NRS;
            const bool ld_lane = COLS == 16 || (l & 15) < COLS; // discarded tile columns issue no load
            float4 b[NRS];
#pragma unroll
            for (int i = 0; i < PF; i++)
                b[i] = ld_lane ? ld_sc1(rDG, off + 64 * i) : float4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NRS; i++) {
                if (i + PF < NRS)
                    b[i + PF] = ld_lane ? ld_sc1(rDG, off + 64 * (i + PF)) : float4{0.f, 0.f, 0.f, 0.f};
                if (BF16) {
                    if (i & 1)
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                       __builtin_bit_cast(bf16x8, b[i]), acc1, 0, 0, 0);
                    else
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]),
                                                                       __builtin_bit_cast(bf16x8, b[i]), acc0, 0, 0, 0);
                } else if (i & 1) {
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
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) red[(w * 4 + r) * 64 + l] = acc0[r] + acc1[r];
            __syncthreads();
        }
        if (w < EW) {
            if (FUSE) dhy = dhyb[cur][cc * 16 + jj]; // written by the followers one step ago (before barrier A)
            if (has_next) {
                // D[row = 4*(lane>>4) + reg][col = lane&15]  ->  unit jj lives in lane (jj>>2)*16 + cc, reg jj&3
                // M4: D[reg = column & 3][lane = 32*(column >> 2) + unit]
                const int src = M4 ? 32 * (cc >> 2) + jj : (jj >> 2) * 16 + cc, reg = M4 ? (cc & 3) : (jj & 3);
#pragma unroll
                for (int ww = 0; ww < 8; ww++) dhn += redp[(ww * 4 + reg) * 64 + src];
            }
            const float dh = dhy + dhn;                         // R/lstm.cc:228
            float dcv = dh * og + dcn;                          // :233
            dcv = dcv * (1.0f - cv * cv);                       // :235
            const float d_o = (dh * cv) * (og * (1.0f - og));   // :238,244
            const float d_i = (dcv * ug) * (ig * (1.0f - ig));  // :239,244
            const float d_f = (dcv * cp) * (fg * (1.0f - fg));  // :240,244
            const float d_u = (dcv * ig) * (1.0f - ug * ug);    // :241,247
            dcn = dcv * fg;                                     // :256
            if (FUSE && ecol < B) {                             // db += dg, R/lstm.cc:252
                dbacc[0] += d_i;
                dbacc[1] += d_o;
                dbacc[2] += d_f;
                dbacc[3] += d_u;
            }
            // 4x4 transpose over the four lanes of a quad by DPP (no LDS: the follower waves keep the LDS queue busy at this
            // point, and four staged writes + a read cost ~900 cycles here): lane (column cc, unit jj = 4*tq + ta) ends
            // up with gate ta of units 4*tq .. 4*tq+3, one 16-byte store
            const int ta = jj & 3, tq = jj >> 2;
            float t0 = d_i, t1 = d_o, t2 = d_f, t3 = d_u;
            {
                const float lo = (ta & 1) ? t0 : t1, hi = (ta & 1) ? t2 : t3;
                const float rlo = dpp_f<0xB1>(lo), rhi = dpp_f<0xB1>(hi); // quad_perm [1,0,3,2]
                if (ta & 1) {
                    t0 = rlo;
                    t2 = rhi;
                } else {
                    t1 = rlo;
                    t3 = rhi;
                }
                const float s0 = (ta & 2) ? t0 : t2, s1 = (ta & 2) ? t1 : t3;
                const float r0 = dpp_f<0x4E>(s0), r1 = dpp_f<0x4E>(s1); // quad_perm [2,3,0,1]
                if (ta & 2) {
                    t0 = r0;
                    t1 = r1;
                } else {
                    t2 = r0;
                    t3 = r1;
                }
            }
            BSTAMP(0, 5)
            const float4 v = {t0, t1, t2, t3};
            float4 v2 = v;
            if (BF16) { // the next quad's four units of the same gate: lane + 4 within the row of 16 (row_shl:4)
                v2.x = dpp_f<0x104>(t0);
                v2.y = dpp_f<0x104>(t1);
                v2.z = dpp_f<0x104>(t2);
                v2.w = dpp_f<0x104>(t3);
            }
            if (ecol < B) {
                if (BF16) { // fp32 copy for the dU product (read after the launch); bf16 copy is the hand-off
                    *reinterpret_cast<float4 *>(DG + ((size_t)t * B + ecol) * G4 + ta * N + 16 * kb + 4 * tq) = v;
                    if ((tq & 1) == 0) {
                        const size_t eoff = ((size_t)t * B + ecol) * G4 + ta * N + 16 * kb + 4 * tq; // in bf16 elements
                        if (XCD_LOCAL && local_pub) *reinterpret_cast<u32x4 *>(DGb + eoff) = pack_bf16x8(v, v2);
                        else __builtin_amdgcn_raw_buffer_store_b128(pack_bf16x8(v, v2), rDG, (int)(eoff * sizeof(unsigned short)), 0, 16);
                    }
                } else {
                    if (XCD_LOCAL && local_pub)
                        *reinterpret_cast<float4 *>(DG + ((size_t)t * B + ecol) * G4 + ta * N + 16 * kb + 4 * tq) = v;
                    else
                        st_sc1(v, rDG, (int)((((size_t)t * B + ecol) * G4 + ta * N + 16 * kb + 4 * tq) * sizeof(float)));
                }
            }
            if (t > 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains, then signals for itself:
                                                                 // EW arrivals per workgroup and step
                if (l == 0)
                    __hip_atomic_fetch_add(cnt + ((size_t)(t * NG + g) * CNT_SLOTS + (kb & (BWD_SH - 1))) * CNT_STRIDE, 1u, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
            if (FUSE) { // dg_t for the dW follower (read after the next workgroup barrier): off the chain
                float *sp = stage[cur] + (cc * 4) * 16 + jj;
                sp[0] = d_i;
                sp[16] = d_o;
                sp[32] = d_f;
                sp[48] = d_u;
            }
        } else if (FUSE) {
            if (w == EW && has_next) update(cur ^ 1, t + 1); // dg_{t+1}, published a step ago; off the critical path
            if (w >= 4 && t >= 2) output_layer(t - 1);       // one step ahead of its use
        }
    }
    if (FUSE) {
        __syncthreads();
        update(1, 1); // dg_1
        __syncthreads();
        float *base = gpart + (size_t)g * ((size_t)G4 * 256 + (size_t)G4 * N + G4 + (size_t)256 * N);
        // dW partial: table row r = gate*16 + unit  ->  gradient row gate*N + 16*kb + unit
        for (int i = tid; i < 256 * 64; i += 512) {
            const int x = i >> 6, r = i & 63;
            base[(size_t)x * G4 + (r >> 4) * N + 16 * kb + (r & 15)] = dWt[i];
        }
        // dWhy partial: D[row = m in tile = 4*(lane>>4)+reg][col = unit = lane&15]
        if (w >= 4) {
            float *Yp = base + (size_t)G4 * 256 + (size_t)G4 * N + G4;
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                float4 v;
                v.x = yacc[mt][0];
                v.y = yacc[mt][1];
                v.z = yacc[mt][2];
                v.w = yacc[mt][3];
                *reinterpret_cast<float4 *>(Yp + (size_t)(16 * kb + (l & 15)) * 256 + 64 * (w - 4) + 16 * mt + 4 * q) = v;
            }
        }
        // db partial: fold the columns in order
        if (tid < ETH) {
#pragma unroll
            for (int gt = 0; gt < 4; gt++) red[(cc * 4 + gt) * 16 + jj] = dbacc[gt];
        }
        __syncthreads();
        if (tid < 64) {
            const int gt = tid >> 4, rj = tid & 15;
            float sum = 0.0f;
            for (int c = 0; c < COLS; c++) sum += red[(c * 4 + gt) * 16 + rj];
            base[(size_t)G4 * 256 + (size_t)G4 * N + gt * N + 16 * kb + rj] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
size_t persistent_counter_bytes(int S, int B) {
    const int NG = (B + 3) / 4; // the narrowest column groups in use (bf16 backward recurrence at small batches)
    return (size_t)(S + 1) * NG * CNT_SLOTS * CNT_STRIDE * sizeof(unsigned);
}

template <class K> static int blocks_per_cu(K kernel, int threads, size_t dyn_lds = 0) {
    if (dyn_lds > 0)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, dyn_lds) != hipSuccess) return 0;
    return n;
}

#define FWD_CASES(X) X(1) X(2) X(4) X(8) X(16)
#define BWD_CASES(X) X(2) X(4) X(8) X(16) X(32)
constexpr size_t DW_TABLE_BYTES = 257 * 64 * sizeof(float); // dynamic LDS of the fused backward form

// which forward kernel serves a shape (one rule, no switches):
//   8-column form (k_fwd_persistent4): N = 256, 512, 1024, more than one 8-column group, one workgroup per CU fits
//   second form (k_fwd_persistent2):   N = 128, 256, 512, 1024 otherwise (e.g. the evaluator's B = 1)
//   first form (k_fwd_persistent):     every other multiple of 64
static bool fwd_second_form(int N) { return N == 128 || N == 256 || N == 512 || N == 1024; }
// Columns one launch of the fp32 two-half forms takes (N = 256, 512): as many 8-column groups as are co-resident at one
// workgroup per CU.  A wider batch runs as several launches over column ranges (the streams are independent recurrences).
// CU count of the current device (the launchers' shape rules must agree with the ones lstm_hip_create applied)
static int current_device_cus() {
    static int cached_dev = -1, cached_cus = 0; // (a process drives one device; re-queried if that ever changes)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev == cached_dev) return cached_cus;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
    cached_dev = dev, cached_cus = cus;
    return cus;
}
int two_half_launch_cols(int N, int n_cus) { return 8 * (n_cus / (N / 16)); }
// 8 columns per workgroup (two alternating halves), or 4 (one half, twice the workgroups) where the whole batch then still
// fits one launch -- the forward form only (the fused backward form's side waves are built around eight columns)
int two_half_group_cols(int N, int B, int n_cus) {
    static const int force = getenv("LSTM_HIP_FWD_GCOLS") ? atoi(getenv("LSTM_HIP_FWD_GCOLS")) : 0; // A/B: 4 or 8
    if ((N != 256 && N != 512) || force == 8 || (B + 3) / 4 > n_cus / (N / 16)) return 8;
    return 4;
}
bool two_half_wide(int N, int B, int n_cus) { return (N == 256 || N == 512) && B > two_half_launch_cols(N, n_cus) && n_cus >= N / 16; }
bool fwd_uses_8col_form(int N, int B, int n_cus) {
    // (N = 256, 512: any batch -- with one half per workgroup and a pinned launch a single stream is one group of 32 / 16
    // workgroups on one XCD, 1.8 us a step where the second form's 64 workgroups over all XCDs took 3.75; N = 1024 has the
    // 8-column one-recurrence form only, which wants more than one group)
    static const bool narrow_off = getenv("LSTM_HIP_NARROW_TWO_HALF") && atoi(getenv("LSTM_HIP_NARROW_TWO_HALF")) == 0; // A/B
    return (N == 256 || N == 512 || N == 1024) && (B > 8 || ((N == 256 || N == 512) && !narrow_off)) &&
           ((N / 16) * ((B + 7) / 8) <= n_cus || two_half_wide(N, B, n_cus));
}
// 8-column groups in the backward recurrence when that still fits one workgroup per CU (more CUs pulling fewer bytes
// each); on v_mfma_f32_4x4x1 for fp32 (N a multiple of 64)
int bwd_group_cols(int N, int B, int n_cus) { return (N / 16) * ((B + 7) / 8) <= n_cus || two_half_wide(N, B, n_cus) ? 8 : 16; }
bool bwd_uses_m4(int N, int cols, bool bf16) { return cols == 8 && !bf16 && N % 64 == 0 && N <= 1024; }
// floats in one column group's partial gradient block [dW | dU | db | dWhy]
size_t bwd_partial_floats(int N) { return (size_t)4 * N * 256 + (size_t)4 * N * N + (size_t)4 * N + (size_t)256 * N; }

bool bwd_scatter_supported(int N, int B, int n_cus, bool fused);
// All workgroups of a recurrence wait on each other, so its grid must be co-resident.  The occupancy API is asked about
// exactly the instantiation that will be launched, with its dynamic LDS; it can over-report by one block per CU
// (MI355X_MICROARCH.md, residency), so one is taken off wherever more than one is claimed.
static bool grid_fits(size_t grid, int per_cu, int n_cus) {
    if (per_cu > 1) per_cu -= 1;
    if (per_cu > 8) per_cu = 8;
    return per_cu >= 1 && grid <= (size_t)per_cu * n_cus;
}
bool persistent_supported(int N, int B, int n_cus, bool fused) {
    if (N % 64 != 0 || N > 1024) return false;
    if (two_half_wide(N, B, n_cus)) // several launches per direction: the two-half forms or nothing
        return (N == 512 ? blocks_per_cu(k_fwd_persistent6<512, false>, FWD4_THREADS) : blocks_per_cu(k_fwd_persistent6<256, false>, FWD4_THREADS)) >= 1 &&
               bwd_scatter_supported(N, B, n_cus, fused);
    int fb = 0, bb = 0;
    size_t fwd_grid = 0;
    if (fwd_uses_8col_form(N, B, n_cus)) {
        fwd_grid = (size_t)(N / 16) * ((B + 7) / 8);
        if (N == 512 && blocks_per_cu(k_fwd_persistent6<512, false>, FWD4_THREADS) < 1) return false;
        if (N == 256 && blocks_per_cu(k_fwd_persistent6<256, false>, FWD4_THREADS) < 1) return false;
        switch (N / 256) {
#define X(k) case k: fb = blocks_per_cu(k_fwd_persistent4<k, false>, FWD4_THREADS); break;
            X(1) X(2) X(4)
#undef X
        }
    } else if (fwd_second_form(N)) {
        fwd_grid = (size_t)(N / 8) * ((B + 15) / 16);
        switch (N / 128) {
#define X(k) case k: fb = blocks_per_cu(k_fwd_persistent2<k, false>, 512); break;
            X(1) X(2) X(4) X(8)
#undef X
        }
    } else {
        fwd_grid = (size_t)(N / 4) * ((B + 15) / 16);
        switch (N / 64) {
#define X(k) case k: fb = blocks_per_cu(k_fwd_persistent<k, false>, 256); break;
            FWD_CASES(X)
#undef X
            default: return false;
        }
    }
    const int cols = bwd_group_cols(N, B, n_cus);
    const bool fuse = fused && cols == 8;
    const size_t lds = fuse ? DW_TABLE_BYTES : 0;
    switch (N / 32) {
#define X(k)                                                                                                           \
    case k:                                                                                                            \
        if (cols == 8 && fuse) bb = blocks_per_cu(k_bwd_persistent<k, 8, true, false, false, true>, 512, lds);         \
        else if (cols == 8) bb = blocks_per_cu(k_bwd_persistent<k, 8, false, false, false, true>, 512, 0);             \
        else bb = blocks_per_cu(k_bwd_persistent<k, 16, false>, 512, 0);                                               \
        break;
        BWD_CASES(X)
#undef X
        default: return false;
    }
    return grid_fits(fwd_grid, fb, n_cus) && grid_fits((size_t)(N / 16) * ((B + cols - 1) / cols), bb, n_cus);
}
// the bf16 recurrence launches other instantiations on other grids: 8-column groups in the backward recurrence when they
// are co-resident (bwd_group_cols_bf16), else 16-column groups two workgroups to a CU
static int bwd_bf16_blocks(int N, int cols, bool fused) {
    int bb = 0;
    switch (N / 32) {
#define X(k)                                                                                             \
    case k:                                                                                              \
        bb = cols == 16 ? blocks_per_cu(k_bwd_persistent<k, 16, false, false, true>, 512, 0)             \
             : cols == 4 ? blocks_per_cu(k_bwd_persistent<k, 4, false, false, true>, 512, 0)             \
             : fused    ? blocks_per_cu(k_bwd_persistent<k, 8, true, false, true>, 512, DW_TABLE_BYTES)  \
                        : blocks_per_cu(k_bwd_persistent<k, 8, false, false, true>, 512, 0);             \
        break;
        X(4) X(8) X(16) X(32)
#undef X
    }
    return bb;
}
int bwd_group_cols_bf16(int N, int B, int n_cus, bool fused) {
    // 4-column groups when even 8-column groups would leave half the CUs idle (BASELINE configs[4]: hidden 1024, 16 streams
    // per GPU = 128 workgroups of 8 columns): twice the CUs, each pulling and multiplying half as much per step
    if (!fused && B % 4 == 0 && (size_t)(N / 16) * ((B + 7) / 8) * 2 <= (size_t)n_cus &&
        grid_fits((size_t)(N / 16) * (B / 4), bwd_bf16_blocks(N, 4, false), n_cus))
        return 4;
    if (grid_fits((size_t)(N / 16) * ((B + 7) / 8), bwd_bf16_blocks(N, 8, fused), n_cus)) return 8;
    return 16;
}
bool persistent_supported_bf16(int N, int B, int n_cus, bool fused) {
    if (N % 128 != 0 || N > 1024) return false;
    int fb = 0, bb = 0;
    size_t fwd_grid = 0;
    if (N % 256 == 0) {
        fwd_grid = (size_t)(N / 8) * ((B + 15) / 16);
        switch (N / 256) {
#define X(k) case k: fb = blocks_per_cu(k_fwd_persistent2_bf16<k, false>, 512); break;
            X(1) X(2) X(4)
#undef X
        }
    } else {
        fwd_grid = (size_t)(N / 4) * ((B + 15) / 16);
        switch (N / 128) {
#define X(k) case k: fb = blocks_per_cu(k_fwd_persistent_bf16<k, false>, 256); break;
            X(1) X(2) X(4) X(8)
#undef X
        }
    }
    const int cols = bwd_group_cols_bf16(N, B, n_cus, fused);
    bb = bwd_bf16_blocks(N, cols, fused && cols == 8);
    return grid_fits(fwd_grid, fb, n_cus) && grid_fits((size_t)(N / 16) * ((B + cols - 1) / cols), bb, n_cus);
}

// ------------------------------------------------------------------------------------------------
// One stream, hidden <= 128 (the reference's own default: R/lstm.cc, alice29, N = 128, S = 25, batch 1 = BASELINE configs[0]):
// the whole recurrence on ONE CU.  U (4N x N fp32 = 256 KB at N = 128) fits the register file of one 1024-thread workgroup,
// so a step needs no hand-off between CUs at all -- two or three workgroup barriers instead of an L2 round trip (the
// 16-workgroup form took 2.4 us a step at this shape).
//   forward : thread (row, k-part) keeps U[row][its N/KP values of k]; step = partial dot products -> LDS -> the N threads of
//             the units add the parts and W[:, x_t] + b, gates / cell (R/lstm.cc:176-192), h_t to LDS and H / C / G
//   backward: thread (unit j, row-part) keeps U[its 4N/RQ rows][j] (from the Ubwd tile image) and Why[its 256/RQ rows][j]; step = partial sums of
//             U^T dg_{t+1} + Why^T dy_t (R/lstm.cc:228) -> LDS -> unit threads: R/lstm.cc:233-247,256, dg_t to LDS and DG
// dW, db, dWhy, dU are the time-batched launches of the unfused path.
// ------------------------------------------------------------------------------------------------
// (workgroup barrier for LDS traffic only: __syncthreads() also waits for the wave's global stores -- H, C, G, DG of the step,
// about a microsecond each time -- which nothing in these kernels reads back)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N_, bool FAST>
__global__ __launch_bounds__(1024) void k_small_fwd(const float *__restrict__ U, const float *__restrict__ W, const float *__restrict__ bias,
                                                    float *__restrict__ H, float *__restrict__ C, float *__restrict__ G,
                                                    const int32_t *__restrict__ xi, int S) {
    constexpr int N = N_, G4 = 4 * N, KP = 1024 / G4, KW = N / KP;
    static_assert(N == 128 || N == 64, "single-CU form: hidden 64 or 128");
    __shared__ __attribute__((aligned(16))) float hs[N];
    __shared__ float part[KP][G4];
    const int tid = threadIdx.x, row = tid % G4, kp = tid / G4;
    float u[KW];
#pragma unroll
    for (int i = 0; i < KW; i++) u[i] = U[(size_t)(kp * KW + i) * G4 + row];
    float cprev = 0.0f, bs[4] = {0.f, 0.f, 0.f, 0.f}, wx[4] = {0.f, 0.f, 0.f, 0.f};
    auto gather = [&](int t) { // W[:, x_t] (R/lstm.cc:176 with a one-hot x), requested a step ahead
#pragma unroll
        for (int g = 0; g < 4; g++) wx[g] = 0.f;
        const int x = xi[t];
        if (x >= 0) {
#pragma unroll
            for (int g = 0; g < 4; g++) wx[g] = W[(size_t)x * G4 + g * N + tid];
        }
    };
    if (tid < N) {
        hs[tid] = H[tid];
        cprev = C[tid];
#pragma unroll
        for (int g = 0; g < 4; g++) bs[g] = bias[g * N + tid];
        gather(1);
    }
    __syncthreads();
    for (int t = 1; t < S; t++) {
        f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f}; // v_pk_fma_f32: two multiply-adds per lane and instruction
#pragma unroll
        for (int i = 0; i < KW; i += 4) {
            const float4 hv = *reinterpret_cast<const float4 *>(hs + kp * KW + i); // same address in every lane: a broadcast read
            a01 = __builtin_elementwise_fma(f32x2_t{u[i + 0], u[i + 1]}, f32x2_t{hv.x, hv.y}, a01);
            a23 = __builtin_elementwise_fma(f32x2_t{u[i + 2], u[i + 3]}, f32x2_t{hv.z, hv.w}, a23);
        }
        part[kp][row] = (a01[0] + a01[1]) + (a23[0] + a23[1]);
        lds_barrier();
        if (tid < N) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float uh = part[0][g * N + tid];
#pragma unroll
                for (int q = 1; q < KP; q++) uh += part[q][g * N + tid];
                pre[g] = (wx[g] + uh) + bs[g]; // R/lstm.cc:176
            }
            const float ig = p_sigm<FAST>(pre[0]), og = p_sigm<FAST>(pre[1]), fg = p_sigm<FAST>(pre[2]); // :179
            const float ug = p_tanh<FAST>(pre[3]);                                                        // :182
            const float cv = p_tanh<FAST>(ig * ug + fg * cprev);                                          // :185-189
            const float hv = og * cv;                                                                     // :192
            cprev = cv;
            hs[tid] = hv;
            if (t + 1 < S) gather(t + 1);
            H[(size_t)t * N + tid] = hv;
            C[(size_t)t * N + tid] = cv;
            float *gc = G + (size_t)t * G4 + tid;
            gc[0] = ig;
            gc[N] = og;
            gc[2 * N] = fg;
            gc[3 * N] = ug;
        }
        lds_barrier();
    }
}
// (512 threads: a thread's 128 + 64 weights at hidden 128 need the 256-register budget of eight waves)
constexpr int SMALL_BWD_THREADS = 512;
template <int N_>
__global__ __launch_bounds__(SMALL_BWD_THREADS) void k_small_bwd(const float4 *__restrict__ Ubwd, const float *__restrict__ Why, const float *__restrict__ dY,
                                                    const float *__restrict__ G, const float *__restrict__ C, float *__restrict__ DG, int S) {
    constexpr int NT = SMALL_BWD_THREADS, N = N_, G4 = 4 * N, RQ = NT / N, RW = G4 / RQ, MW = 256 / RQ;
    static_assert(N == 128 || N == 64, "single-CU form: hidden 64 or 128");
    __shared__ __attribute__((aligned(16))) float dgs[G4];
    __shared__ __attribute__((aligned(16))) float dys[256];
    __shared__ float part[RQ][N];
    const int tid = threadIdx.x, j = tid % N, rq = tid / N;
    // The thread's weights: U[its 4N/RQ rows][j] from the tile image Ubwd (k_pack_U / k_adagrad: one float4 = four consecutive
    // rows of a column, sixteen columns side by side -- 256-byte runs across a wave) and Why[its 256/RQ output rows][j]
    // (contiguous along the rows for a fixed j: 16-byte pieces, all requested at once).
    float ub[RW], wy[MW];
#pragma unroll
    for (int i = 0; i < RW; i += 4) {
        const int r = rq * RW + i;
        const float4 v = Ubwd[((size_t)(j >> 4) * (N / 4) + (r >> 4)) * 64 + (((r & 15) >> 2) << 4) + (j & 15)];
        ub[i] = v.x, ub[i + 1] = v.y, ub[i + 2] = v.z, ub[i + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < MW; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(Why + (size_t)j * 256 + rq * MW + i);
        wy[i] = v.x, wy[i + 1] = v.y, wy[i + 2] = v.z, wy[i + 3] = v.w;
    }
    for (int i = tid; i < G4; i += NT) dgs[i] = 0.0f; // dhnext = 0, R/lstm.cc:216
    float dcn = 0.0f;                                  // dcnext, :217
    // operands of a step are requested a step ahead: dy_t (256 threads), gates and cells of the unit threads
    float dyv = tid < 256 ? dY[(size_t)(S - 2) * 256 + tid] : 0.0f;
    float ig = 0.f, og = 0.f, fg = 0.f, ug = 0.f, cv = 0.f, cp = 0.f;
    auto fetch = [&](int t) {
        const float *gc = G + (size_t)t * G4 + tid;
        ig = gc[0], og = gc[N], fg = gc[2 * N], ug = gc[3 * N];
        cv = C[(size_t)t * N + tid], cp = C[(size_t)(t - 1) * N + tid];
    };
    if (tid < N) fetch(S - 1);
    for (int t = S - 1; t >= 1; t--) {
        if (tid < 256) {
            dys[tid] = dyv;
            if (t >= 2) dyv = dY[(size_t)(t - 2) * 256 + tid];
        }
        lds_barrier();
        f32x2_t a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MW; i += 4) { // Why^T dy_t
            const float4 v = *reinterpret_cast<const float4 *>(dys + rq * MW + i);
            a01 = __builtin_elementwise_fma(f32x2_t{wy[i + 0], wy[i + 1]}, f32x2_t{v.x, v.y}, a01);
            a23 = __builtin_elementwise_fma(f32x2_t{wy[i + 2], wy[i + 3]}, f32x2_t{v.z, v.w}, a23);
        }
#pragma unroll
        for (int i = 0; i < RW; i += 4) { // U^T dg_{t+1}
            const float4 v = *reinterpret_cast<const float4 *>(dgs + rq * RW + i);
            a01 = __builtin_elementwise_fma(f32x2_t{ub[i + 0], ub[i + 1]}, f32x2_t{v.x, v.y}, a01);
            a23 = __builtin_elementwise_fma(f32x2_t{ub[i + 2], ub[i + 3]}, f32x2_t{v.z, v.w}, a23);
        }
        part[rq][j] = (a01[0] + a01[1]) + (a23[0] + a23[1]);
        lds_barrier();
        if (tid < N) {
            float dh = part[0][tid]; // R/lstm.cc:228
#pragma unroll
            for (int q = 1; q < RQ; q++) dh += part[q][tid];
            float dcv = dh * og + dcn;                          // :233
            dcv = dcv * (1.0f - cv * cv);                       // :235
            const float d_o = (dh * cv) * (og * (1.0f - og));   // :238,244
            const float d_i = (dcv * ug) * (ig * (1.0f - ig));  // :239,244
            const float d_f = (dcv * cp) * (fg * (1.0f - fg));  // :240,244
            const float d_u = (dcv * ig) * (1.0f - ug * ug);    // :241,247
            dcn = dcv * fg;                                     // :256
            dgs[tid] = d_i, dgs[N + tid] = d_o, dgs[2 * N + tid] = d_f, dgs[3 * N + tid] = d_u;
            if (t >= 2) fetch(t - 1);
            float *dp = DG + (size_t)t * G4 + tid;
            dp[0] = d_i, dp[N] = d_o, dp[2 * N] = d_f, dp[3 * N] = d_u;
        }
        lds_barrier();
    }
}
bool small_recurrence_supported(int N, int B) { return B == 1 && (N == 128 || N == 64); }
void small_fwd(const float *U, const float *W, const float *bias, float *H, float *C, float *G, const int32_t *xi, int N, int S, bool fast,
               hipStream_t st) {
#define GO(n)                                                                                                     \
    do {                                                                                                          \
        if (fast) hipLaunchKernelGGL((k_small_fwd<n, true>), dim3(1), dim3(1024), 0, st, U, W, bias, H, C, G, xi, S);  \
        else hipLaunchKernelGGL((k_small_fwd<n, false>), dim3(1), dim3(1024), 0, st, U, W, bias, H, C, G, xi, S);      \
    } while (0)
    if (N == 128) GO(128);
    else GO(64);
#undef GO
}
void small_bwd(const float4 *Ubwd, const float *Why, const float *dY, const float *G, const float *C, float *DG, int N, int S, hipStream_t st) {
    if (N == 128) hipLaunchKernelGGL((k_small_bwd<128>), dim3(1), dim3(SMALL_BWD_THREADS), 0, st, Ubwd, Why, dY, G, C, DG, S);
    else hipLaunchKernelGGL((k_small_bwd<64>), dim3(1), dim3(SMALL_BWD_THREADS), 0, st, Ubwd, Why, dY, G, C, DG, S);
}

// ---- forward ---------------------------------------------------------------------------------------------------------
void fwd_persistent(const float4 *Ufwd, const float *W, const float *bias, float *H, float *C, float *G,
                    const int32_t *xi, unsigned *cnt, unsigned *abortp, unsigned epoch, int N, int S, int B, bool fast,
                    hipStream_t st) {
    if (fwd_second_form(N)) {
        const dim3 grid2(N / 8, (B + 15) / 16), block2(512);
        switch (N / 128) {
#define X(k)                                                                                                          \
    case k:                                                                                                           \
        if (fast) hipLaunchKernelGGL((k_fwd_persistent2<k, true>), grid2, block2, 0, st, Ufwd, W, bias, H, C, G, xi, cnt, abortp, epoch, S, B); \
        else hipLaunchKernelGGL((k_fwd_persistent2<k, false>), grid2, block2, 0, st, Ufwd, W, bias, H, C, G, xi, cnt, abortp, epoch, S, B);    \
        break;
            X(1) X(2) X(4) X(8)
#undef X
        }
        return;
    }
    const dim3 grid(N / 4, (B + 15) / 16), block(256);
    switch (N / 64) {
#define X(k)                                                                                                          \
    case k:                                                                                                           \
        if (fast) hipLaunchKernelGGL((k_fwd_persistent<k, true>), grid, block, 0, st, Ufwd, W, bias, H, C, G, xi, cnt, abortp, epoch, S, B); \
        else hipLaunchKernelGGL((k_fwd_persistent<k, false>), grid, block, 0, st, Ufwd, W, bias, H, C, G, xi, cnt, abortp, epoch, S, B);    \
        break;
        FWD_CASES(X)
#undef X
    }
}

size_t fwd_ring_floats(int N, int B) { return (size_t)HX_RING * N * B; }
int fwd_ring_advance(int ring_base, int S) { return (ring_base + S - 1) & (HX_RING - 1); }
void fwd_persistent4(const float4 *Ufwd4, const float *W, const float *bias, float *H, float *C, float *G, const int32_t *xi,
                     float *Hx, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N, int S, int B, bool fast,
                     int poll_cfg, hipStream_t st, unsigned long long *stamps) {
    const dim3 grid(N / 16, (B + 7) / 8), block(FWD4_THREADS);
    if (stamps != nullptr && N == 512) { // diagnostic build of the headline shape
        hipLaunchKernelGGL((k_fwd_persistent4<2, false, true>), grid, block, 0, st, Ufwd4, W, bias, H, C, G, xi, Hx, cnt, abortp,
                           epoch, ring_base, S, B, poll_cfg, stamps);
        return;
    }
    switch (N / 256) {
#define X(k)                                                                                                          \
    case k:                                                                                                           \
        if (fast) hipLaunchKernelGGL((k_fwd_persistent4<k, true>), grid, block, 0, st, Ufwd4, W, bias, H, C, G, xi, Hx, cnt, abortp, epoch, ring_base, S, B, poll_cfg, nullptr); \
        else hipLaunchKernelGGL((k_fwd_persistent4<k, false>), grid, block, 0, st, Ufwd4, W, bias, H, C, G, xi, Hx, cnt, abortp, epoch, ring_base, S, B, poll_cfg, nullptr);    \
        break;
        X(1) X(2) X(4)
#undef X
    }
}

// two-half form (k_fwd_persistent6): N = 512 or 256 on the grid of the 8-column form
bool fwd_uses_two_half_form(int N, int B, int n_cus) { return (N == 512 || N == 256) && fwd_uses_8col_form(N, B, n_cus); }
void fwd_persistent6(const float4 *Ufwd5, const float *W, const float *bias, float *H, float *C, float *G, const int32_t *xi,
                     float *Hx, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N, int S, int B, bool fast,
                     int poll_cfg, hipStream_t st, unsigned long long *stamps, int col0, int cols) {
    const int GC = two_half_group_cols(N, B, current_device_cus());
    const int NGh = ((cols > 0 ? cols : B) + GC - 1) / GC; // groups of this launch: columns [col0, col0 + cols)
    static const bool no_pin = getenv("LSTM_HIP_NO_PIN") && atoi(getenv("LSTM_HIP_NO_PIN")); // A/B
    const bool pinned = NGh < 8 && !no_pin; // one group per XCD (see the kernel)
    poll_cfg = (poll_cfg & 0xffff) | (pinned ? NGh << 16 : 0) | ((col0 / GC) << 20) | (GC == 4 ? 1 << 28 : 0);
    const dim3 grid(N / 16, pinned ? 8 : NGh), block(FWD4_THREADS);
#define F6_GO(...)                                                                                                            \
    hipLaunchKernelGGL((k_fwd_persistent6<__VA_ARGS__>), grid, block, 0, st, Ufwd5, W, bias, H, C, G, xi, Hx, cnt, abortp, epoch, \
                       ring_base, S, B, poll_cfg, stamps)
    if (N == 512) {
        if (stamps != nullptr) F6_GO(512, false, true);
        else if (fast) F6_GO(512, true);
        else F6_GO(512, false);
    } else {
        stamps = nullptr; // (stamped builds exist for the headline shape)
        if (fast) F6_GO(256, true);
        else F6_GO(256, false);
    }
#undef F6_GO
}

// 8-column groups in the bf16 forward recurrence (second form) when 16-column groups would leave half the CUs idle
static int fwd_bf16_cols(int N, int B, int n_cus) {
    return N % 256 == 0 && B % 8 == 0 && (size_t)(N / 8) * ((B + 15) / 16) * 2 <= (size_t)n_cus &&
                   (size_t)(N / 8) * (B / 8) <= (size_t)n_cus
               ? 8
               : 16;
}
// ---- bf16 two-half forward form (k_fwd_halves_bf16) ----
int fwd_halves_bf16_units(int N) { return N == 1024 ? 32 : 16; }
// Ufwd6b[kb][w][r][set][ab][l] = bf16 x 4 { U[gate (l & 3) of unit UW*kb + 16*set + (l >> 2)][k = Kw*w + 64*r + 4*ab + e], e = 0..3 }
__global__ __launch_bounds__(256) void k_pack_Ufwd6_bf16(const float *__restrict__ U, uint2 *__restrict__ img, int N, int UW) {
    const int G4 = 4 * N, Kw = N / 8, NRK = Kw >= 64 ? Kw / 64 : 1, NAB = Kw >= 64 ? 16 : Kw / 4, NSET = UW / 16;
    const size_t total = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e & 63);
        size_t q = e >> 6;
        const int ab = (int)(q % NAB);
        q /= NAB;
        const int sx = (int)(q % NSET);
        q /= NSET;
        const int r = (int)(q % NRK);
        q /= NRK;
        const int w = (int)(q % 8), kb = (int)(q / 8);
        const int row = (l & 3) * N + UW * kb + 16 * sx + (l >> 2), k = Kw * w + 64 * r + 4 * ab;
        float v[4];
#pragma unroll
        for (int x = 0; x < 4; x++) v[x] = U[(size_t)(k + x) * G4 + row];
        img[e] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}
void pack_Ufwd6_bf16(const float *U, void *img, int N, hipStream_t st) {
    const size_t n = (size_t)N * N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_pack_Ufwd6_bf16, dim3(blocks), dim3(256), 0, st, U, reinterpret_cast<uint2 *>(img), N, fwd_halves_bf16_units(N));
}
size_t fwd_halves_bf16_ring_halfwords(int N, int B) { return (size_t)HX_RING * N * B; }
#define FHB_DISPATCH(GO)                  \
    do {                                  \
        if (N == 1024) GO(1024, 32);      \
        else if (N == 512) GO(512, 16);   \
        else GO(256, 16);                 \
    } while (0)
// columns one launch takes: as many 8-column groups as are co-resident, one workgroup per CU (a wider batch runs as several
// launches over column ranges -- the streams are independent recurrences)
// Columns per workgroup: 8 (two alternating halves), or 4 -- one half per workgroup, twice the workgroups -- where the whole
// batch then still fits ONE launch.  Narrow batches leave XCDs idle (configs[4]: 16 streams = two 8-column groups on two
// XCDs), and a workgroup with one half never has that half's data waiting behind the other half's matrix phase.  Measured
// (forward / backward / window): N=1024 B=16 259 / 280 us, 0.708 ms -> 212 / 219 us, 0.601 ms; B=32 260 / 286 -> 215 / 223;
// N=512 B=16 141 / 172 -> 137 / 156; N=256 B=32 134 / 128 -> 128 / 115.  (A batch that would need a second launch with
// 4-column groups keeps 8: hidden 1024 with 64 streams is one launch of 625 us, not two of 435.)
int bf16_group_cols(int N, int B, int n_cus) {
    static const int force = getenv("LSTM_HIP_BF16_GCOLS") ? atoi(getenv("LSTM_HIP_BF16_GCOLS")) : 0; // A/B: 4 or 8
    const int fit = n_cus / (N / fwd_halves_bf16_units(N)); // groups of one launch
    if (force == 8 || B % 4 != 0 || (B + 3) / 4 > fit) return 8;
    if (force == 4) return 4;
    return BF16_SINGLE_HALF_DEFAULT ? 4 : 8;
}
int fwd_halves_bf16_launch_cols(int N, int B, int n_cus) { return bf16_group_cols(N, B, n_cus) * (n_cus / (N / fwd_halves_bf16_units(N))); }
bool fwd_halves_bf16_supported(int N, int B, int n_cus) {
    if (N != 256 && N != 512 && N != 1024) return false;
    if (B % 4 != 0) return false; // 8-byte ring pieces and Hb rows
    if (n_cus / (N / fwd_halves_bf16_units(N)) < 1) return false;
    int per_cu = 0;
#define GO(n, u)                                                                                                                  \
    per_cu = blocks_per_cu(k_fwd_halves_bf16<n, u, false>, FwdhbShape<n, u>::THREADS, FwdhbShape<n, u>::LDS) > 0 &&               \
                     blocks_per_cu(k_fwd_halves_bf16<n, u, true>, FwdhbShape<n, u>::THREADS, FwdhbShape<n, u>::LDS) > 0           \
                 ? 1                                                                                                              \
                 : 0
    FHB_DISPATCH(GO);
#undef GO
    return per_cu >= 1;
}
// one launch: columns [col0, col0 + cols) of the B, cols <= fwd_halves_bf16_launch_cols; ring_base is that column range's own
void fwd_halves_bf16(const void *Ufwd6b, const float *W, const float *bias, float *H, unsigned short *Hb, float *C, float *G,
                     const int32_t *xi, void *Hxb, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N, int S,
                     int B, int col0, int cols, bool fast, int n_cus, hipStream_t st, unsigned long long *stamps) {
    const int GC = bf16_group_cols(N, B, n_cus);
    const int NB = N / fwd_halves_bf16_units(N), NG = (cols + GC - 1) / GC;
    static const bool no_pin = getenv("LSTM_HIP_NO_PIN") && atoi(getenv("LSTM_HIP_NO_PIN")); // A/B
    const int pinned = NG < 8 && 8 * NB <= n_cus && !no_pin;
    const dim3 grid(pinned ? 8 * NB : NB * NG);
    const FwdhbArgs args = {reinterpret_cast<const uint2 *>(Ufwd6b), W, bias, H, Hb, C, G, xi, reinterpret_cast<unsigned *>(Hxb), cnt, abortp,
                            epoch, ring_base, S, B, NG, pinned, col0, GC, stamps};
#define GO(n, u)                                                                                                               \
    do {                                                                                                                       \
        if (stamps) hipLaunchKernelGGL((k_fwd_halves_bf16<n, u, false, true>), grid, dim3(FwdhbShape<n, u>::THREADS), (FwdhbShape<n, u>::LDS), st, args); \
        else if (fast) hipLaunchKernelGGL((k_fwd_halves_bf16<n, u, true>), grid, dim3(FwdhbShape<n, u>::THREADS), (FwdhbShape<n, u>::LDS), st, args); \
        else hipLaunchKernelGGL((k_fwd_halves_bf16<n, u, false>), grid, dim3(FwdhbShape<n, u>::THREADS), (FwdhbShape<n, u>::LDS), st, args);     \
    } while (0)
    FHB_DISPATCH(GO);
#undef GO
}
#undef FHB_DISPATCH

void fwd_persistent_bf16(const void *Ufwd16, const float *W, const float *bias, float *H, unsigned short *Hb, float *C,
                         float *G, const int32_t *xi, unsigned *cnt, unsigned *abortp, unsigned epoch, int N, int S, int B,
                         bool fast, hipStream_t st, int n_cus) {
    const u32x4 *U16 = reinterpret_cast<const u32x4 *>(Ufwd16);
    if (N % 256 == 0) {
        if (fwd_bf16_cols(N, B, n_cus) == 8) {
            const dim3 grid8(N / 8, (B + 7) / 8), block2(512);
            switch (N / 256) {
#define X(k)                                                                                                             \
    case k:                                                                                                              \
        if (fast) hipLaunchKernelGGL((k_fwd_persistent2_bf16<k, true, 8>), grid8, block2, 0, st, U16, W, bias, H, Hb, C, G, xi, cnt, abortp, epoch, S, B); \
        else hipLaunchKernelGGL((k_fwd_persistent2_bf16<k, false, 8>), grid8, block2, 0, st, U16, W, bias, H, Hb, C, G, xi, cnt, abortp, epoch, S, B);    \
        break;
                X(1) X(2) X(4)
#undef X
            }
            return;
        }
        const dim3 grid2(N / 8, (B + 15) / 16), block2(512);
        switch (N / 256) {
#define X(k)                                                                                                             \
    case k:                                                                                                              \
        if (fast) hipLaunchKernelGGL((k_fwd_persistent2_bf16<k, true>), grid2, block2, 0, st, U16, W, bias, H, Hb, C, G, xi, cnt, abortp, epoch, S, B); \
        else hipLaunchKernelGGL((k_fwd_persistent2_bf16<k, false>), grid2, block2, 0, st, U16, W, bias, H, Hb, C, G, xi, cnt, abortp, epoch, S, B);    \
        break;
            X(1) X(2) X(4)
#undef X
        }
        return;
    }
    const dim3 grid(N / 4, (B + 15) / 16), block(256);
    switch (N / 128) {
#define X(k)                                                                                                             \
    case k:                                                                                                              \
        if (fast) hipLaunchKernelGGL((k_fwd_persistent_bf16<k, true>), grid, block, 0, st, U16, W, bias, H, Hb, C, G, xi, cnt, abortp, epoch, S, B); \
        else hipLaunchKernelGGL((k_fwd_persistent_bf16<k, false>), grid, block, 0, st, U16, W, bias, H, Hb, C, G, xi, cnt, abortp, epoch, S, B);    \
        break;
        X(1) X(2) X(4) X(8)
#undef X
    }
}

// ---- backward --------------------------------------------------------------------------------------------------------
// (one buffer serves whichever form the handle runs: the dg ring of the two-half form or the partial-sum ring of the scatter form)
size_t bwd_ring_floats(int N, int B) {
    const size_t dg = (size_t)HX_RING * 4 * N * B, q = (N == 512 || N == 256) ? bwds_ring_floats(N, B) : 0;
    return dg > q ? dg : q;
}
int bwd_ring_advance(int ring_base, int S) { return (ring_base - (S - 1)) & (HX_RING - 1); }
// columns per group of the fp32 scatter form for this shape (4 = one half per workgroup): the fused partial gradient blocks
// are one per group
int bwd_scatter_group_cols(int N, int B, int n_cus) {
    static const bool force8 = getenv("LSTM_HIP_BWD_GCOLS") && atoi(getenv("LSTM_HIP_BWD_GCOLS")) == 8; // A/B
    return two_half_group_cols(N, B, n_cus) == 4 && !force8 ? 4 : 8;
}
int bwds_ring_advance(int ring_base, int S) {
    if (BWDS_TAGGED) return (ring_base + (S > 2 ? S - 2 : 0)) & 7; // publication number: slot = low two bits, parity = bit 2
    return (ring_base - (S - 2)) & (HX_RING - 1);
}

// scatter form of the backward recurrence (k_bwd_scatter): the shapes of the two-half form
bool bwd_scatter_supported(int N, int B, int n_cus, bool fused) {
    if ((N != 512 && N != 256) || bwd_group_cols(N, B, n_cus) != 8) return false;
    const int lg = two_half_launch_cols(N, n_cus) / 8, ng = (B + 7) / 8;
    const size_t grid = (size_t)(N / 16) * (ng < lg ? ng : lg); // of one launch
    int per_cu = 0;
    if (N == 512)
        per_cu = fused ? blocks_per_cu(k_bwd_scatter<512, true>, BWDH_THREADS, bwdh_lds_bytes(true))
                       : blocks_per_cu(k_bwd_scatter<512, false>, BWDH_THREADS, bwdh_lds_bytes(false));
    else
        per_cu = fused ? blocks_per_cu(k_bwd_scatter<256, true>, BWDH_THREADS, bwdh_lds_bytes(true))
                       : blocks_per_cu(k_bwd_scatter<256, false>, BWDH_THREADS, bwdh_lds_bytes(false));
    return per_cu >= 1 && grid <= (size_t)n_cus;
}
void bwd_scatter(const float4 *Ubwd6, float *DG, const float *Why, const float *dY, const float *G, const float *C, const float *H,
                 const int32_t *xi, float *gpart, float *Qx, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N,
                 int S, int B, int cfg, hipStream_t st, unsigned long long *stamps, int col0, int cols) {
    // fewer than 8 groups: pinned launch, one group per XCD (see BWDH_COMMON); cfg bit 16 (spread mapping, tests) keeps the plain one
    const int GC = bwd_scatter_group_cols(N, B, current_device_cus());
    const int NGh = ((cols > 0 ? cols : B) + GC - 1) / GC; // groups of this launch: columns [col0, col0 + cols)
    cfg = (cfg & 0xffff) | ((col0 / GC) << 20) | (GC == 4 ? 1 << 28 : 0);
    static const bool no_pin = getenv("LSTM_HIP_NO_PIN") && atoi(getenv("LSTM_HIP_NO_PIN")); // A/B
    const bool pinned = NGh < 8 && !(cfg & 16) && !no_pin;
    if (pinned) cfg |= NGh << 16;
    const dim3 grid(N / 16, pinned ? 8 : NGh), block(BWDH_THREADS);
    const bool fuse = gpart != nullptr;
    const size_t lds = bwdh_lds_bytes(fuse);
#define BS_GO(...)                                                                                                                   \
    do {                                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bwd_scatter<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds);                                                                                         \
        hipLaunchKernelGGL((k_bwd_scatter<__VA_ARGS__>), grid, block, lds, st, args);                                                \
    } while (0)
    const BwdhArgs args = {Ubwd6, DG, Why, dY, G, C, H, xi, gpart, Qx, cnt, abortp, epoch, ring_base, S, B, cfg,
                           N == 512 ? stamps : nullptr};
    if (N == 256) {
        if (fuse) BS_GO(256, true, false);
        else BS_GO(256, false, false);
    } else if (stamps != nullptr) {
        if (fuse) BS_GO(512, true, true);
        else BS_GO(512, false, true);
    } else if (fuse)
        BS_GO(512, true, false);
    else
        BS_GO(512, false, false);
#undef BS_GO
}

// bf16 scatter form (k_bwd_scatter_bf16): N = 256 / 512 / 1024, 8-column groups.  Hidden 1024 runs 32 units to a workgroup so
// that a group (32 workgroups) fits one XCD; with fewer than 8 groups the launch is pinned: 8 * NB workgroups, group g = the
// ones the dispatcher's round robin puts on XCD g (workgroup i -> XCD i % 8), the others leave at once.  (The kernel checks
// the placement it really got, HW_REG_XCC_ID, and keeps to device-scope stores where it is not what was asked for.)
int bwd_scatter_bf16_units(int N) { return N == 1024 ? 32 : 16; }
// publication number of the next launch's first hand-off (slot = low two bits, parity = bit 2)
int bwd_scatter_bf16_ring_advance(int base, int S) { return (base + (S > 2 ? S - 2 : 0)) & 7; }
int bf16_group_cols(int N, int B, int n_cus);
int bwd_scatter_bf16_launch_cols(int N, int B, int n_cus) { return bf16_group_cols(N, B, n_cus) * (n_cus / (N / bwd_scatter_bf16_units(N))); }
size_t bwd_scatter_bf16_ring_floats(int N, int B, int n_cus) {
    (void)n_cus;
    return bwdsb_ring_floats(N, bwd_scatter_bf16_units(N), B); // a region per column group of the whole batch
}
bool bwd_scatter_bf16_supported(int N, int B, int n_cus) {
    if (N != 256 && N != 512 && N != 1024) return false;
    (void)B;
    if (n_cus / (N / bwd_scatter_bf16_units(N)) < 1) return false;
    int per_cu = 0;
    if (N == 1024) per_cu = blocks_per_cu(k_bwd_scatter_bf16<1024, 32>, BwdsbShape<1024, 32>::THREADS);
    else if (N == 512) per_cu = blocks_per_cu(k_bwd_scatter_bf16<512, 16>, BwdsbShape<512, 16>::THREADS);
    else per_cu = blocks_per_cu(k_bwd_scatter_bf16<256, 16>, BwdsbShape<256, 16>::THREADS);
    return per_cu >= 1;
}
// one launch: columns [col0, col0 + cols) of the B, cols <= bwd_scatter_bf16_launch_cols
void bwd_scatter_bf16(const void *Ubwd6b, float *DG, const float *DHy, const float *G, const float *C, float *Qx, unsigned *cnt,
                      unsigned *abortp, unsigned epoch, int ring_base, int N, int S, int B, int col0, int cols, int n_cus, hipStream_t st,
                      unsigned long long *stamps, unsigned short *DGt_b, int Tpad) {
    const int GC = bf16_group_cols(N, B, n_cus);
    const int NB = N / bwd_scatter_bf16_units(N), NG = (cols + GC - 1) / GC;
    static const int spread = getenv("LSTM_HIP_BWD_SPREAD") && atoi(getenv("LSTM_HIP_BWD_SPREAD")) ? 1 : 0;
    const int pinned = NG < 8 && 8 * NB <= n_cus && !spread;
    const dim3 grid(pinned ? 8 * NB : NB * NG);
    const BwdsbArgs args = {reinterpret_cast<const uint2 *>(Ubwd6b), DG, DHy, G, C, Qx, cnt, abortp, epoch, ring_base, S, B, NG, pinned, col0, GC, DGt_b, Tpad, stamps};
    if (N == 1024 && stamps) hipLaunchKernelGGL((k_bwd_scatter_bf16<1024, 32, true>), grid, dim3(BwdsbShape<1024, 32>::THREADS), 0, st, args);
    else if (N == 1024) hipLaunchKernelGGL((k_bwd_scatter_bf16<1024, 32>), grid, dim3(BwdsbShape<1024, 32>::THREADS), 0, st, args);
    else if (N == 512 && stamps) hipLaunchKernelGGL((k_bwd_scatter_bf16<512, 16, true>), grid, dim3(BwdsbShape<512, 16>::THREADS), 0, st, args);
    else if (N == 512) hipLaunchKernelGGL((k_bwd_scatter_bf16<512, 16>), grid, dim3(BwdsbShape<512, 16>::THREADS), 0, st, args);
    else hipLaunchKernelGGL((k_bwd_scatter_bf16<256, 16>), grid, dim3(BwdsbShape<256, 16>::THREADS), 0, st, args);
}

void bwd_persistent(const float4 *Ubwd, float *DG, const float *DHy, const float *G, const float *C, const float *H,
                    const int32_t *xi, float *gpart, const float *Why, const float *dY, unsigned *cnt, unsigned *abortp,
                    unsigned epoch, int N, int S, int B, int cols, hipStream_t st, unsigned long long *stamps,
                    unsigned short *DGb) {
    const dim3 grid(N / 16, (B + cols - 1) / cols), block(512);
    const bool fuse = gpart != nullptr;
    const size_t lds = fuse ? DW_TABLE_BYTES : 0;
    // test hook: LSTM_HIP_BWD_SPREAD=1 keeps the dispatch-order workgroup mapping (column groups spread over all XCDs)
    static const int spread = getenv("LSTM_HIP_BWD_SPREAD") && atoi(getenv("LSTM_HIP_BWD_SPREAD")) ? 1 : 0;
#define BWD_GO(...)                                                                                                    \
    do {                                                                                                               \
        if (fuse) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_bwd_persistent<__VA_ARGS__>),           \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)DW_TABLE_BYTES);          \
        hipLaunchKernelGGL((k_bwd_persistent<__VA_ARGS__>), grid, block, lds, st, Ubwd, DG, DHy, G, C, H, xi, gpart, Why, dY, cnt, \
                           abortp, epoch, S, B, spread, stamps, DGb);                                                  \
    } while (0)
    if (DGb != nullptr) { // bf16 recurrence: 8- or 16-column groups, 16x16x32 tiles, counter hand-off
        switch (N / 32) {
#define X(k)                                                  \
    case k:                                                   \
        if (cols == 16) BWD_GO(k, 16, false, false, true);    \
        else if (cols == 4) BWD_GO(k, 4, false, false, true); \
        else if (fuse) BWD_GO(k, 8, true, false, true);       \
        else BWD_GO(k, 8, false, false, true);                \
        break;
            X(4) X(8) X(16) X(32)
#undef X
        }
        return;
    }
    if (bwd_uses_m4(N, cols, false)) { // Ubwd is the 4x4x1 image here (the caller packs it when bwd_uses_m4 says so)
        if (stamps != nullptr && N == 512) { // diagnostic build of the headline shape
            if (fuse) BWD_GO(16, 8, true, true, false, true);
            else BWD_GO(16, 8, false, true, false, true);
            return;
        }
        stamps = nullptr;
        switch (N / 32) {
#define X(k)                                                                   \
    case k:                                                                    \
        if (fuse) BWD_GO(k, 8, true, false, false, true);                      \
        else BWD_GO(k, 8, false, false, false, true);                          \
        break;
            X(2) X(4) X(8) X(16) X(32)
#undef X
        }
        return;
    }
    stamps = nullptr;
    switch (N / 32) { // 16-column groups on 16x16x4 tiles (batches too large for 8-column groups on one workgroup per CU)
#define X(k) case k: BWD_GO(k, 16, false); break;
        BWD_CASES(X)
#undef X
    }
#undef BWD_GO
}

} // namespace lstmk
