// lstm_hip_api.cpp -- the C ABI of include/lstm_hip.h over the gfx950 kernels.
//
// One lstm_hip_ctx = what the reference keeps in cuParameters p, d, m and cuLSTM<S>
// (OV/lstm_eigen_class_CUDA/cu_lstm.h:20-304), laid out for one MI355X:
//   P, dP, mem : flat [W|U|b|Why|by] blocks (dP is also the RCCL all-reduce payload)
//   H, C       : [S][B][N]   (= N x (S*B) column-major; columns t*B.. are step t)
//   G, DG      : [S][B][4N]  post-activation gates / their gradients
//   Y          : [S][B][256] logits, overwritten in place by dY;  Pr: probs
//   DHy        : [S][B][N]   Why^T * dY for every step at once
// so the time-batched products see plain column-major matrices with T = (S-1)*B columns.
#include "../../include/lstm_hip.h"
#include "kernels.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

using namespace lstmk;

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(LSTM_HIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

enum KernelId {
    K_PACK_U, K_FWD_STEP, K_GEMM_Y, K_SOFTMAX, K_LOSS, K_GEMM_DHY, K_BWD_STEP, K_GEMM_DWHY, K_GEMM_DU, K_DW_DB,
    K_DBY, K_ADAGRAD, K_SLIDE, K_ALLREDUCE, K_FWD_PERSIST, K_BWD_PERSIST, K_COUNT
};
const char *const kKernelNames[K_COUNT] = {
    "pack_U", "fwd_step", "gemm_Y", "softmax_loss_dy", "loss_reduce", "gemm_DHy", "bwd_step", "gemm_dWhy", "gemm_dU",
    "dW_db", "loss_dby", "adagrad", "slide", "allreduce", "fwd_persistent", "bwd_persistent"};

// ---- RCCL, loaded on first use so single-GPU users never touch it --------------------------
struct UniqueId {
    char internal[LSTM_HIP_UNIQUE_ID_BYTES];
};
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr; // optional
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
int rccl_load() {
    if (g_rccl.lib) return 0;
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(LSTM_HIP_ERCCL, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(lib, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(LSTM_HIP_ERCCL, "librccl lacks a required symbol");
    g_rccl.lib = lib;
    return 0;
}

} // namespace

struct lstm_hip_ctx {
    lstm_hip_config cfg{};
    ParamLayout pl{};
    int T = 0; // (S-1)*B columns in the time-batched matrices
    hipStream_t st = nullptr;
    hipStream_t st2 = nullptr; // the early part of the gradient all-reduce runs here, beside the dU product on `st`
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr, evt0 = nullptr, evt1 = nullptr;
    bool fold_pending = false;  // the backward pass left the gradient in pieces for Adagrad to sum (single-GPU loop)
    int n_slabs_dU = 0;         // ... with this many dU slabs (0: dU is final in dP)
    bool in_loop = false;       // inside lstm_hip_train_windows: nobody reads the gradient block between backward and Adagrad
    size_t dU_reduced = 0;      // ... and so are this many leading floats of the dU range (its first column half)
    bool early_reduced = false; // [dW] and [db | dWhy | dby] are already being all-reduced on st2 (ev_join marks the end)
    float *slabs_dU = nullptr; // split-K slabs of dU
    bool bf16 = false;         // LSTM_HIP_BF16_RECURRENCE
    bool packed16 = false;
    unsigned short *Hb = nullptr, *DGb = nullptr; // bf16 hand-off copies of h and dg
    void *Ufwd16 = nullptr, *Ubwd16 = nullptr;    // bf16 fragment images of U
    void *Ubwd6b = nullptr;                       // ... and the scatter-form backward's (bwd_scatter16)
    bool bwd_scatter16 = false;                   // bf16 backward recurrence in its scatter form (k_bwd_scatter_bf16)
    void *Ufwd6b = nullptr, *Hxb = nullptr;       // two-half bf16 forward form: weights image, bf16 hand-off ring
    bool fwd_halves16 = false;
    bool slide_in_adagrad = true;                 // LSTM_HIP_SLIDE_IN_ADAGRAD=0 (per handle): A/B
    bool carry_slide = false, pre_slid = false;   // window loop: this Adagrad launch carries the next window's slide / it has been done
    bool small = false;                           // one stream, hidden <= 128: both recurrences on one CU (k_small_fwd / k_small_bwd)
    bool dgt_written = false;                     // the backward recurrence wrote the transposed bf16 image of dg itself
    bool packed6b = false;                        // Ubwd6b is current (written by the Adagrad launch)
    bool packedf6b = false;                       // ... and Ufwd6b
    // bf16 operands of the four time-batched products, k contiguous (kernels.h, gemm_bf16): Why^T and Why; per window
    // h^T [N][SBpad], dy^T [256][Tpad], dg^T [4N][Tpad] and dy [T][256]
    unsigned short *WhyT_b = nullptr, *Why_b = nullptr, *Ht_b = nullptr, *dYt_b = nullptr, *DGt_b = nullptr, *dYb = nullptr;
    int Tpad = 0, SBpad = 0;
    bool why_packed = false;
    float *gpart = nullptr;    // per-column-group partial [dW|dU|db|dWhy] blocks of the fused backward recurrence
    int bwd_cols = 16;         // batch columns per backward-recurrence workgroup (8 or 16)

    float *P = nullptr, *dP = nullptr, *mem = nullptr;
    float4 *Ufwd = nullptr, *Ubwd = nullptr;
    float4 *Ubwd4 = nullptr; // weight image of the 4x4x1 backward form (kernels.hip, k_pack_U), when bwd_uses_m4
    float4 *Ufwd4 = nullptr; // ... of the 8-column forward kernel (fwd_uses_8col_form) or, fwd_cols4, of the two-half one
    int n_cus = 0;           // compute units of the device (grid choices)
    bool side_stream = true; // LSTM_HIP_NO_SIDE_STREAM=1 (per handle): keep the whole window on one stream
    int probe_overlap = 0;   // LSTM_HIP_PROBE_OVERLAP=k (per handle, timing probe): k Y-sized products on st2 beside the forward recurrence
    int bwd_halves = 0;      // != 0: backward recurrence likewise, scatter form (k_bwd_scatter); bits above bit 0: its cfg word
    int half_forms() const { return fwd_cols4 | (bwd_halves ? 4 : 0); } // which U images are live (kernels.h)
    int fwd_cols4 = 0;       // 1: forward recurrence as two alternating 4-column halves per workgroup (k_fwd_persistent6)
    float *Hx = nullptr;     // 8-column forward form: ring of hand-off slots (data-as-flag), sentinel-filled
    int ring_base = 0;       // slot of step 0 in the next launch
    float *DGx = nullptr;    // backward recurrence: the same kind of ring for dg
    int ring_base_b = 0;
    int gpart_cols = 8;      // columns per fused partial gradient block (4 where the scatter form runs one half per workgroup)
    size_t DGx_floats = 0;   // size of the backward hand-off ring
    int poll_cfg = 0;        // LSTM_HIP_FWD_POLL: bits 0-7 s_sleep between polls, 8-15 first delay of the non-gating waves
    bool packed = false;
    float *H = nullptr, *C = nullptr, *G = nullptr, *DG = nullptr, *Y = nullptr, *Pr = nullptr, *DHy = nullptr;
    float *dcnext = nullptr, *colloss = nullptr, *dby_part = nullptr, *slabs = nullptr;
    char *dw_scratch = nullptr;
    int n_dby_parts = 0;
    int splits_dWhy = 1, splits_dU = 1;
    int32_t *xi = nullptr, *ti = nullptr;            // flat [S][B] indices the kernels read
    int32_t *Xr = nullptr, *Tr = nullptr, *head = nullptr; // ring form kept by the device-side slide
    double *d_loss = nullptr;
    double *d_losses = nullptr;
    double *h_losses = nullptr; // pinned twin of d_losses: the caller's (pageable) array is filled from it after the sync.
                                // (The runtime's own staging path for a first large pageable copy cost the NEXT 20 windows
                                // 0.4 ms of device time -- tools/train_windows_probe.py.)
    int64_t losses_cap = 0;
    uint8_t *text = nullptr;
    uint64_t text_len = 0;
    uint64_t *pos = nullptr;
    int32_t global_B = 0;
    lstm_hip_ctx *eval_h = nullptr; // internal B = 1 handle used by lstm_hip_eval_bits
    int stride = 1, carry_col = 1; // window advance per iteration and the column that becomes the carry
    bool fwd_done = false;
    bool dby_done = false;       // dby already produced by the loss launch of this window
    bool persistent = false;     // default engine; false = one launch per timestep
    unsigned *cnt = nullptr;     // [2][persistent_counter_bytes]: fwd region, bwd region
    unsigned *abortp = nullptr;  // set by a timed-out spin inside a persistent kernel
    size_t cnt_bytes = 0;
    unsigned long long *stamps = nullptr; // LSTM_HIP_DEBUG_STAMPS: [fwd, bwd][2 workgroups][S][16] s_memtime values
    int loss_mode = 0; // LSTM_HIP_LOSS_*
    unsigned fwd_epoch = 0, bwd_epoch = 0; // launches so far on the cumulative hand-off counters

    void *comm = nullptr;
    int nranks = 1, rank = 0;

    bool profiling = false;
    int64_t launches[K_COUNT] = {};
    double total_ms[K_COUNT] = {};
};

namespace {

// Every launch is checked: a kernel that could not be launched (resources, an attribute that was refused) must not leave the
// window to complete "successfully" on stale buffers.
int launch_status(int id) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LSTM_HIP_EHIP, "launch of %s failed: %s", kKernelNames[id], hipGetErrorString(e));
    return 0;
}
template <class F> int timed(lstm_hip_ctx *h, int id, F &&launch) {
    if (!h->profiling) {
        launch();
        return launch_status(id);
    }
    HIP_TRY(hipEventRecord(h->ev0, h->st));
    launch();
    if (int rc = launch_status(id)) return rc;
    HIP_TRY(hipEventRecord(h->ev1, h->st));
    HIP_TRY(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->launches[id]++;
    h->total_ms[id] += ms;
    return 0;
}
#define RUN(id, ...)                                               \
    do {                                                           \
        int rc_ = timed(h, id, [&]() { __VA_ARGS__; });            \
        if (rc_) return rc_;                                       \
    } while (0)

template <class T> int dalloc(T **p, size_t count) {
    HIP_TRY(hipMalloc((void **)p, count * sizeof(T)));
    HIP_TRY(hipMemset(*p, 0, count * sizeof(T)));
    return 0;
}
#define ALLOC(p, n)                  \
    do {                             \
        int rc_ = dalloc(&(p), (n)); \
        if (rc_) return rc_;         \
    } while (0)

int check(lstm_hip_ctx *h) {
    if (!h) return fail(LSTM_HIP_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    return 0;
}
#define CHECK(h)             \
    do {                     \
        int rc_ = check(h);  \
        if (rc_) return rc_; \
    } while (0)

// after a synchronisation point: did a persistent kernel give up on a hand-off?
int check_abort(lstm_hip_ctx *h) {
    if (!h->persistent) return 0;
    unsigned flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, h->abortp, sizeof(unsigned), hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    if (flag != 0) {
        HIP_TRY(hipMemsetAsync(h->abortp, 0, sizeof(unsigned), h->st));
        HIP_TRY(hipMemsetAsync(h->cnt, 0, 2 * h->cnt_bytes, h->st)); // counters are inconsistent after an abort
        h->fwd_epoch = h->bwd_epoch = 0;
        if (h->Hx) { // and so are the hand-off rings
            HIP_TRY(hipMemsetAsync(h->Hx, 0xff, sizeof(float) * fwd_ring_floats(h->cfg.N, h->cfg.B), h->st));
            h->ring_base = 0;
        }
        if (h->Hxb) {
            HIP_TRY(hipMemsetAsync(h->Hxb, 0xff, sizeof(unsigned short) * fwd_halves_bf16_ring_halfwords(h->cfg.N, h->cfg.B), h->st));
            h->ring_base = 0;
        }
        if (h->DGx) {
            HIP_TRY(hipMemsetAsync(h->DGx, 0xff, sizeof(float) * h->DGx_floats, h->st));
            h->ring_base_b = 0;
        }
        return fail(LSTM_HIP_ESTATE, "a persistent recurrence kernel timed out waiting for a hand-off (results invalid)");
    }
    return 0;
}

// reported loss: all S-1 steps in bits (R/lstm.cc:204-207), or the last step only -- in nats
// (OV/lstm_eigen_class_CUDA/lstm.h:200-221) or in bits (cuLSTM::calculate_loss, cu_lstm.h:203-215);
// colloss holds -log2 p(target) per (step, column)
bool loss_last_step(const lstm_hip_ctx *h) { return h->loss_mode != LSTM_HIP_LOSS_ALL_STEPS_BITS; }
const float *loss_src(const lstm_hip_ctx *h) {
    return loss_last_step(h) ? h->colloss + (size_t)(h->cfg.S - 2) * h->cfg.B : h->colloss;
}
int loss_steps(const lstm_hip_ctx *h) { return loss_last_step(h) ? 1 : h->cfg.S - 1; }
float loss_scale(const lstm_hip_ctx *h) { return h->loss_mode == LSTM_HIP_LOSS_LAST_STEP_NATS ? 0.693147180559945f : 1.0f; }

int launch_fwd_recurrence(lstm_hip_ctx *h) {
    const int N = h->cfg.N, B = h->cfg.B, S = h->cfg.S;
    const bool fast = (h->cfg.flags & LSTM_HIP_FAST_MATH) != 0;
    if (h->fwd_epoch >= (1u << 26)) { // keep epoch * arrivals inside 32 bits
        HIP_TRY(hipMemsetAsync(h->cnt, 0, h->cnt_bytes, h->st));
        h->fwd_epoch = 0;
    }
    h->fwd_epoch++;
    if (h->small) {
        RUN(K_FWD_PERSIST, small_fwd(h->P + h->pl.U, h->P + h->pl.W, h->P + h->pl.b, h->H, h->C, h->G, h->xi, N, S, fast, h->st));
        return 0;
    }
    if (h->bf16) {
        if (!h->packed16) {
            // (the one-recurrence forms' images only where one of them runs)
            RUN(K_PACK_U, (h->fwd_halves16 && h->bwd_scatter16 ? (void)0 : pack_U_bf16(h->P + h->pl.U, h->Ufwd16, h->Ubwd16, N, h->st),
                           h->bwd_scatter16 && !h->packed6b ? pack_U6_bf16(h->P + h->pl.U, h->Ubwd6b, N, h->st) : (void)0,
                           h->fwd_halves16 && !h->packedf6b ? pack_Ufwd6_bf16(h->P + h->pl.U, h->Ufwd6b, N, h->st) : (void)0));
            h->packed16 = true;
        }
        if (h->fwd_halves16) { // as many 8-column groups per launch as are co-resident; the streams are independent
            const int lc = fwd_halves_bf16_launch_cols(N, B, h->n_cus);
            for (int c0 = 0; c0 < B; c0 += lc) {
                if (c0 > 0) h->fwd_epoch++;
                RUN(K_FWD_PERSIST, fwd_halves_bf16(h->Ufwd6b, h->P + h->pl.W, h->P + h->pl.b, h->H, h->Hb, h->C, h->G, h->xi, h->Hxb,
                                                   h->cnt, h->abortp, h->fwd_epoch, h->ring_base, N, S, B, c0, B - c0 < lc ? B - c0 : lc,
                                                   fast, h->n_cus, h->st, h->stamps));
            }
            h->ring_base = fwd_ring_advance(h->ring_base, S); // (every column range has made the same S - 1 hand-offs on its part of the ring)
            return 0;
        }
        RUN(K_FWD_PERSIST, fwd_persistent_bf16(h->Ufwd16, h->P + h->pl.W, h->P + h->pl.b, h->H, h->Hb, h->C, h->G, h->xi,
                                               h->cnt, h->abortp, h->fwd_epoch, N, S, B, fast, h->st, h->n_cus));
    } else if (h->Hx && h->fwd_cols4) { // as many 8-column groups per launch as are co-resident (one launch unless the batch is wide)
        const int lc = two_half_launch_cols(N, h->n_cus);
        for (int c0 = 0; c0 < B; c0 += lc) {
            if (c0 > 0) h->fwd_epoch++;
            RUN(K_FWD_PERSIST, fwd_persistent6(h->Ufwd4, h->P + h->pl.W, h->P + h->pl.b, h->H, h->C, h->G, h->xi, h->Hx, h->cnt,
                                               h->abortp, h->fwd_epoch, h->ring_base, N, S, B, fast, h->poll_cfg, h->st, h->stamps, c0,
                                               B - c0 < lc ? B - c0 : lc));
        }
        h->ring_base = fwd_ring_advance(h->ring_base, S); // (every column has made the same S - 1 hand-offs on its part of the ring)
    } else if (h->Hx) {
        RUN(K_FWD_PERSIST, fwd_persistent4(h->Ufwd4, h->P + h->pl.W, h->P + h->pl.b, h->H, h->C, h->G, h->xi, h->Hx, h->cnt,
                                           h->abortp, h->fwd_epoch, h->ring_base, N, S, B, fast, h->poll_cfg, h->st, h->stamps));
        h->ring_base = fwd_ring_advance(h->ring_base, S);
    } else {
        RUN(K_FWD_PERSIST, fwd_persistent(h->Ufwd, h->P + h->pl.W, h->P + h->pl.b, h->H, h->C, h->G, h->xi, h->cnt, h->abortp,
                                          h->fwd_epoch, N, S, B, fast, h->st));
    }
    return 0;
}

int do_forward(lstm_hip_ctx *h) {
    const int N = h->cfg.N, B = h->cfg.B, S = h->cfg.S, G4 = 4 * N;
    const bool fast = (h->cfg.flags & LSTM_HIP_FAST_MATH) != 0;
    if (!h->packed && !h->bf16) { // (the bf16 path packs its own images, launch_fwd_recurrence)
        RUN(K_PACK_U, pack_U(h->P + h->pl.U, h->Ufwd4 ? nullptr : h->Ufwd, h->Ubwd4 ? nullptr : h->Ubwd, N, h->st, h->Ubwd4,
                             h->Ufwd4, h->half_forms())); // one image per direction is live
        h->packed = true;
    }
    h->n_dby_parts = softmax_parts(h->T);
    // Timing probe (LSTM_HIP_PROBE_OVERLAP=1, tools/ab_kernels.py; results unaffected): the Y product of the PREVIOUS window's
    // H is launched on st2 beside the forward recurrence, into the Pr buffer (overwritten by the softmax later), to measure
    // what a time-batched product costs the recurrence when both share the chip ("overlap").
    const int probe_overlap = h->probe_overlap;
    if (probe_overlap && h->persistent && !h->bf16 && !h->profiling) {
        HIP_TRY(hipEventRecord(h->ev_fork, h->st));
        HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
    }
    if (h->persistent) {
        int rc = launch_fwd_recurrence(h);
        if (rc) return rc;
        if (probe_overlap && !h->bf16 && !h->profiling) {
            for (int rep = 0; rep < (probe_overlap & 15); rep++) {
                if (probe_overlap & 16) // 64 x 64 tiles: can share a compute unit with a workgroup of the recurrence
                    gemm_probe_small_kfast(256, h->T, N, h->P + h->pl.Why, 256, h->H + (size_t)N * B, N, h->Pr + (size_t)256 * B, 256, h->st2);
                else // the library's 128 x 64 tiles (128 KB of LDS): starts only where a recurrence workgroup has left
                    gemm(false, false, 256, h->T, N, h->P + h->pl.Why, 256, h->H + (size_t)N * B, N, h->Pr + (size_t)256 * B, 256, 1,
                         nullptr, h->st2);
            }
            HIP_TRY(hipEventRecord(h->ev_join, h->st2));
            HIP_TRY(hipStreamWaitEvent(h->st, h->ev_join, 0));
        }
    } else {
        for (int t = 1; t < S; t++) {
            RUN(K_FWD_STEP, fwd_step(h->Ufwd, h->P + h->pl.W, h->P + h->pl.b, h->H + (size_t)(t - 1) * N * B,
                                     h->C + (size_t)(t - 1) * N * B, h->H + (size_t)t * N * B, h->C + (size_t)t * N * B,
                                     h->G + (size_t)t * G4 * B, h->xi + (size_t)t * B, N, B, fast, h->st));
        }
    }
    // Y = Why * H[1..S-1]   (R/lstm.cc:195 for every step at once)
    if (h->bf16) { // bf16 operands (Why rounded once per update, the recurrence's own bf16 copy of h), fp32 accumulate
        if (!h->why_packed) {
            RUN(K_PACK_U, (transpose_pack_bf16(h->P + h->pl.Why, N, 256, 256, h->WhyT_b, N, h->st),
                           pack_bf16(h->P + h->pl.Why, (size_t)256 * N, h->Why_b, h->st)));
            h->why_packed = true;
        }
        RUN(K_GEMM_Y, gemm_bf16(256, h->T, N, h->WhyT_b, N, h->Hb + (size_t)N * B, N, h->Y + (size_t)256 * B, 256, 1, nullptr,
                                h->st));
    } else
    RUN(K_GEMM_Y, gemm(false, false, 256, h->T, N, h->P + h->pl.Why, 256, h->H + (size_t)N * B, N,
                       h->Y + (size_t)256 * B, 256, 1, nullptr, h->st));
    RUN(K_SOFTMAX, softmax_loss_dy(h->Y + (size_t)256 * B, h->Pr + (size_t)256 * B, h->P + h->pl.by, h->ti + B,
                                   h->colloss, h->dby_part, 0, h->T, h->st));
    h->fwd_done = true;
    return 0;
}

int do_backward(lstm_hip_ctx *h) {
    const int N = h->cfg.N, B = h->cfg.B, S = h->cfg.S, G4 = 4 * N, T = h->T;
    if (!h->fwd_done) return fail(LSTM_HIP_ESTATE, "backward called before forward");
    float *dY = h->Y + (size_t)256 * B;
    // dby = rowsum(dY) (R/lstm.cc:227): folded with the loss when the loop runs on the device
    if (!h->dby_done)
        RUN(K_DBY, loss_reduce(loss_src(h), loss_steps(h), B, h->global_B, h->d_loss, h->dby_part, h->n_dby_parts,
                               h->dP + h->pl.by, h->st, loss_scale(h)));
    h->dby_done = false;
    // fused mode: the backward recurrence produces DHy = Why^T * dY (R/lstm.cc:228) itself and accumulates dW, db, dWhy
    const bool fused = h->persistent && h->gpart != nullptr && h->bwd_cols == 8;
    if (h->bf16) { // DHy = Why^T * dY on bf16 operands: both already have the contraction index m contiguous
        RUN(K_GEMM_DHY, (pack_bf16(dY, (size_t)T * 256, h->dYb, h->st),
                         gemm_bf16(N, T, 256, h->Why_b, 256, h->dYb, 256, h->DHy + (size_t)N * B, N, 1, nullptr, h->st)));
    } else if (!fused && !h->bwd_halves && !h->small) // (the two-half and the single-CU backward forms compute Why^T dy themselves)
        RUN(K_GEMM_DHY, gemm(true, false, N, T, 256, h->P + h->pl.Why, 256, dY, 256, h->DHy + (size_t)N * B, N, 1, nullptr,
                             h->st));
    // Unfused two-half form, single GPU: the sums that do not feed the recurrence run on st2 beside it -- the column sort of
    // the dW pass and dWhy = dY H^T while the recurrence runs (one workgroup per CU leaves room), the dW / db sums beside
    // the dU product.  (Profiling runs keep everything on `st`, one timed launch after the other.)
    const bool side = h->bwd_halves && !fused && h->side_stream && !h->comm && !h->profiling;
    // (Not for the bf16 scatter form, although its pinned launch leaves most of the chip idle -- configs[4]: 64 of 256 CUs.
    // Measured, kernel trace: of the side stream's launches only the one-workgroup column sort ran beside the recurrence; the
    // next one started and then sat until the recurrence ended, because its workgroups are dealt to the XCDs in turn and the
    // two XCDs the recurrence fills have no room for the ones they are dealt.  Window 0.6993 -> 0.6955 ms: dropped.  The
    // sort alone, queued beside the FORWARD recurrence: 0.7094 against 0.7094.)
    if (side) {
        HIP_TRY(hipEventRecord(h->ev_fork, h->st));
        HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
        dW_sort(h->xi + B, T, G4, h->dw_scratch, h->st2);
        gemm(false, true, 256, N, T, dY, 256, h->H + (size_t)N * B, N, h->dP + h->pl.Why, 256, h->splits_dWhy, h->slabs, h->st2);
    }
    unsigned *cb = h->cnt + h->cnt_bytes / sizeof(unsigned);
    if (h->persistent) {
        if (h->bwd_epoch >= (1u << 26)) {
            HIP_TRY(hipMemsetAsync(cb, 0, h->cnt_bytes, h->st));
            h->bwd_epoch = 0;
        }
        h->bwd_epoch++;
        h->dgt_written = false;
        if (h->small) {
            RUN(K_BWD_PERSIST, small_bwd(h->Ubwd, h->P + h->pl.Why, dY, h->G, h->C, h->DG, N, S, h->st));
        } else if (h->bf16 && h->bwd_scatter16) {
            const int lc = bwd_scatter_bf16_launch_cols(N, B, h->n_cus); // one launch per co-resident range of columns
            static const bool no_direct = getenv("LSTM_HIP_NO_DIRECT_DGT") && atoi(getenv("LSTM_HIP_NO_DIRECT_DGT")); // A/B
            // The recurrence writes the k-contiguous bf16 image of dg for the dU product itself (2-byte stores, off the chain)
            // where that is cheaper than the transposing pass behind it: measured at N=1024 with 16 streams 0.5776 -> 0.5701 ms
            // (recurrence +4 us, dU launch -9); with 64 streams the scattered stores cost the recurrence what the pass costs
            // (N=512: +19 / -18 us) and with 128 more (+76 / -71 us per window), so only for the narrow batches.
            const bool direct_dgt = !no_direct && N == 1024 && B <= 32;
            h->dgt_written = direct_dgt;
            for (int c0 = 0; c0 < B; c0 += lc) {
                if (c0 > 0) h->bwd_epoch++;
                RUN(K_BWD_PERSIST, bwd_scatter_bf16(h->Ubwd6b, h->DG, h->DHy, h->G, h->C, h->DGx, cb, h->abortp, h->bwd_epoch,
                                                    h->ring_base_b, N, S, B, c0, B - c0 < lc ? B - c0 : lc, h->n_cus, h->st,
                                                    h->stamps ? h->stamps + (size_t)2 * S * 16 : nullptr, direct_dgt ? h->DGt_b : nullptr, h->Tpad));
            }
            h->ring_base_b = bwd_scatter_bf16_ring_advance(h->ring_base_b, S); // (every group's region has had its S - 2 publications)
        } else if (h->bf16) {
            RUN(K_BWD_PERSIST, bwd_persistent(reinterpret_cast<const float4 *>(h->Ubwd16), h->DG, h->DHy, h->G, h->C, h->H,
                                              h->xi, fused ? h->gpart : nullptr, h->P + h->pl.Why, dY, cb, h->abortp,
                                              h->bwd_epoch, N, S, B, h->bwd_cols, h->st, nullptr, h->DGb));
        } else if (h->bwd_halves) {
            const int lc = two_half_launch_cols(N, h->n_cus); // one launch per co-resident range of columns; every group of the
            for (int c0 = 0; c0 < B; c0 += lc) {               // batch has its own ring region and partial gradient block
                if (c0 > 0) h->bwd_epoch++;
                RUN(K_BWD_PERSIST, bwd_scatter(h->Ubwd4, h->DG, h->P + h->pl.Why, dY, h->G, h->C, h->H, h->xi, fused ? h->gpart : nullptr,
                                               h->DGx, cb, h->abortp, h->bwd_epoch, h->ring_base_b, N, S, B, h->bwd_halves >> 1, h->st,
                                               h->stamps ? h->stamps + (size_t)2 * S * 16 : nullptr, c0, B - c0 < lc ? B - c0 : lc));
            }
            h->ring_base_b = bwds_ring_advance(h->ring_base_b, S);
        } else {
            RUN(K_BWD_PERSIST, bwd_persistent(h->Ubwd4 ? h->Ubwd4 : h->Ubwd, h->DG, h->DHy, h->G, h->C, h->H, h->xi,
                                              fused ? h->gpart : nullptr, h->P + h->pl.Why, dY, cb, h->abortp, h->bwd_epoch, N, S,
                                              B, h->bwd_cols, h->st, h->stamps ? h->stamps + (size_t)2 * S * 16 : nullptr, nullptr));
        }
    } else {
        HIP_TRY(hipMemsetAsync(h->dcnext, 0, sizeof(float) * N * B, h->st)); // R/lstm.cc:216-217
        for (int t = S - 1; t >= 1; t--) {
            RUN(K_BWD_STEP, bwd_step(h->Ubwd, t < S - 1 ? h->DG + (size_t)(t + 1) * G4 * B : nullptr,
                                     h->DHy + (size_t)t * N * B, h->G + (size_t)t * G4 * B, h->C + (size_t)t * N * B,
                                     h->C + (size_t)(t - 1) * N * B, h->dcnext, h->DG + (size_t)t * G4 * B, N, B, h->st));
        }
    }
    // dWhy = dY * H[1..]^T             R/lstm.cc:226
    if (h->bf16) {
        // the contraction runs over the window's columns: k-contiguous bf16 images of dy, h and dg first (zero-padded to
        // Tpad); dy_t pairs with h_t, i.e. column (t-1)*B+b of dY with column t*B+b of H: the h image shifted by B columns
        RUN(K_GEMM_DWHY, (transpose_pack_bf16(dY, T, 256, 256, h->dYt_b, h->Tpad, h->st),
                          transpose_pack_bf16(h->H, S * B, N, N, h->Ht_b, h->SBpad, h->st),
                          gemm_bf16(256, N, h->Tpad, h->dYt_b, h->Tpad, h->Ht_b + B, h->SBpad, h->dP + h->pl.Why, 256,
                                    h->splits_dWhy, h->slabs, h->st)));
    } else if (!fused && !side)
        RUN(K_GEMM_DWHY, gemm(false, true, 256, N, T, dY, 256, h->H + (size_t)N * B, N, h->dP + h->pl.Why, 256,
                              h->splits_dWhy, h->slabs, h->st));
    // dW, db                           R/lstm.cc:251-252
    // Inside the single-GPU loop nobody reads the gradient block between here and Adagrad: the folds of the group
    // partials and of the dU slabs are left to the Adagrad launch (do_adagrad), three launches fewer per window.
    const bool defer_fold = fused && h->in_loop && !h->comm;
    if (defer_fold) {
        h->fold_pending = true;
    } else if (fused) { // accumulated per column group inside the recurrence: fold the groups in order
        const int NGb = (B + h->gpart_cols - 1) / h->gpart_cols;
        const size_t psz = bwd_partial_floats(N);
        // b and Why are adjacent both in the flat block and in the partial blocks: one fold covers both.  With the early
        // all-reduce below the folds go to st2 with it, so the dU product on `st` does not wait for them.
        const bool on_st2 = h->comm && h->in_loop && !h->profiling;
        hipStream_t fs = h->st;
        if (on_st2) {
            HIP_TRY(hipEventRecord(h->ev_fork, h->st));
            HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
            fs = h->st2;
        }
        RUN(K_DW_DB, (gemm_fold(h->gpart, NGb, G4 * 256, 1, h->dP + h->pl.W, G4 * 256, fs, psz),
                      gemm_fold(h->gpart + (size_t)G4 * 256 + (size_t)G4 * N, NGb, G4 + 256 * N, 1, h->dP + h->pl.b,
                                G4 + 256 * N, fs, psz)));
    } else if (side) {
        HIP_TRY(hipEventRecord(h->ev_mid, h->st)); // DG is complete
        HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_mid, 0));
        dW_sums(h->DG + (size_t)G4 * B, T, G4, h->dP + h->pl.W, h->dP + h->pl.b, h->dw_scratch, h->st2);
        HIP_TRY(hipEventRecord(h->ev_join, h->st2));
    } else {
        RUN(K_DW_DB, dW_db(h->DG + (size_t)G4 * B, h->xi + B, T, G4, h->dP + h->pl.W, h->dP + h->pl.b, h->dw_scratch, h->st));
    }
    // Everything but dU is final here.  Inside the device-resident loop their all-reduce starts now, on st2, beside the dU
    // product: ranges [dW] and [db | dWhy | dby] of the flat block (one group).  The dU range follows on `st` behind the
    // product, ordered after ev_join, so the communicator never runs two collectives at once (do_allreduce).
    if (h->comm && h->in_loop && !h->profiling) {
        if (!fused) { // (fused: st2 already follows `st` from the end of the recurrence, with the folds queued on it)
            HIP_TRY(hipEventRecord(h->ev_fork, h->st));
            HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_fork, 0));
        }
        int rc = 0;
        if (g_rccl.GroupStart && g_rccl.GroupEnd) g_rccl.GroupStart();
        rc = g_rccl.AllReduce(h->dP, h->dP, h->pl.U, /*ncclFloat*/ 7, /*ncclSum*/ 0, h->comm, h->st2);
        if (rc == 0)
            rc = g_rccl.AllReduce(h->dP + h->pl.b, h->dP + h->pl.b, h->pl.total - h->pl.b, 7, 0, h->comm, h->st2);
        if (g_rccl.GroupStart && g_rccl.GroupEnd) {
            const int rc2 = g_rccl.GroupEnd();
            if (rc == 0) rc = rc2;
        }
        if (rc != 0)
            return fail(LSTM_HIP_ERCCL, "ncclAllReduce (early ranges): %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
        // With LSTM_HIP_DU_SPLIT=1 the dU product runs as two column halves and the first half's all-reduce also goes on st2,
        // behind the early ranges and beside the second half's product; only the second half is left for `st`.  Off by
        // default: two half-size products cost more than the whole one, about what hiding half of the dU all-reduce can
        // win back (measured with a 1-rank communicator, tools/comm_overhead_probe.py); to be decided on a multi-GPU node.
        // The switch is read once per process and the path depends on nothing else, so every rank of a job (same
        // environment) posts the same sequence of collectives.
        h->dU_reduced = 0;
        static const bool du_split = getenv("LSTM_HIP_DU_SPLIT") && atoi(getenv("LSTM_HIP_DU_SPLIT")) != 0;
        if (!h->bf16 && du_split) {
            int n1 = (N / 2) / 64 * 64; // whole 64-column tiles in the first half
            if (n1 == 0) n1 = N / 2;
            gemm(false, true, G4, n1, T, h->DG + (size_t)G4 * B, G4, h->H, N, h->dP + h->pl.U, G4, h->splits_dU, h->slabs_dU, h->st);
            HIP_TRY(hipEventRecord(h->ev_mid, h->st));
            gemm(false, true, G4, N - n1, T, h->DG + (size_t)G4 * B, G4, h->H + n1, N, h->dP + h->pl.U + (size_t)G4 * n1, G4,
                 h->splits_dU, h->slabs_dU, h->st);
            HIP_TRY(hipStreamWaitEvent(h->st2, h->ev_mid, 0));
            h->dU_reduced = (size_t)G4 * n1;
            rc = g_rccl.AllReduce(h->dP + h->pl.U, h->dP + h->pl.U, h->dU_reduced, 7, 0, h->comm, h->st2);
            if (rc != 0)
                return fail(LSTM_HIP_ERCCL, "ncclAllReduce (dU, first half): %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
            h->n_slabs_dU = 0;
        }
        HIP_TRY(hipEventRecord(h->ev_join, h->st2));
        h->early_reduced = true;
        if (h->dU_reduced) return 0; // the product is done
    }
    // dU = DG * H[0..S-2]^T            R/lstm.cc:250
    if (h->bf16) { // dg_t pairs with h_{t-1}: column (t-1)*B+b of both images
        h->n_slabs_dU = 0;
        RUN(K_GEMM_DU, (h->dgt_written ? (void)0 : transpose_pack_bf16(h->DG + (size_t)G4 * B, T, G4, G4, h->DGt_b, h->Tpad, h->st),
                        gemm_bf16(G4, N, h->Tpad, h->DGt_b, h->Tpad, h->Ht_b, h->SBpad, h->dP + h->pl.U, G4, h->splits_dU,
                                  h->slabs_dU, h->st)));
    } else if (defer_fold && h->splits_dU > 1)
        RUN(K_GEMM_DU, h->n_slabs_dU = gemm_slabs(false, true, G4, N, T, h->DG + (size_t)G4 * B, G4, h->H, N, h->slabs_dU,
                                                   h->splits_dU, h->st));
    else {
        h->n_slabs_dU = 0;
        RUN(K_GEMM_DU, gemm(false, true, G4, N, T, h->DG + (size_t)G4 * B, G4, h->H, N, h->dP + h->pl.U, G4, h->splits_dU,
                            h->slabs_dU, h->st));
    }
    if (side) HIP_TRY(hipStreamWaitEvent(h->st, h->ev_join, 0)); // the gradient block is complete on `st` from here
    return 0;
}

// SUM all-reduce of the flat gradient block [dW | dU | db | dWhy | dby] over the ranks (SUM, not mean: the reference's
// weight gradients are sums over batch columns, OV/lstm_eigen_opt/lstm.cc:271,297-299).  When do_backward has already
// started the ranges that were final before the dU product (early_reduced), only dU is left: it goes on `st` behind the
// product and behind ev_join, i.e. after the early ranges have finished on st2.
int do_allreduce(lstm_hip_ctx *h) {
    if (!h->comm) return 0;
    int rc = 0;
    if (h->early_reduced) {
        h->early_reduced = false;
        HIP_TRY(hipStreamWaitEvent(h->st, h->ev_join, 0));
        const size_t done = h->dU_reduced;
        h->dU_reduced = 0;
        RUN(K_ALLREDUCE, rc = g_rccl.AllReduce(h->dP + h->pl.U + done, h->dP + h->pl.U + done, h->pl.b - h->pl.U - done, 7, 0, h->comm,
                                               h->st));
    } else
    RUN(K_ALLREDUCE, rc = g_rccl.AllReduce(h->dP, h->dP, h->pl.total, /*ncclFloat*/ 7, /*ncclSum*/ 0, h->comm, h->st));
    if (rc != 0)
        return fail(LSTM_HIP_ERCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    return 0;
}

int do_adagrad(lstm_hip_ctx *h, double lr) {
    const SlideJob job{h->text, h->text_len, h->pos, h->Xr, h->Tr, h->head, h->xi, h->ti, h->H, h->C,
                       h->cfg.S, h->cfg.B, h->cfg.N, h->stride, h->carry_col};
    const SlideJob *sj = h->carry_slide ? &job : nullptr;
    if (h->fold_pending) {
        h->fold_pending = false;
        const int NGb = (h->cfg.B + h->gpart_cols - 1) / h->gpart_cols;
        RUN(K_ADAGRAD, adagrad(h->P, h->dP, h->mem, h->pl.total, (float)lr, h->pl.U, h->cfg.N, h->Ufwd4 ? nullptr : h->Ufwd,
                               h->Ubwd4 ? nullptr : h->Ubwd, h->st, h->Ubwd4, h->Ufwd4, h->gpart, NGb, bwd_partial_floats(h->cfg.N),
                               h->pl.by, h->n_slabs_dU > 0 ? h->slabs_dU : nullptr, h->n_slabs_dU,
                               (size_t)4 * h->cfg.N * h->cfg.N, h->half_forms(), nullptr, 0, nullptr, nullptr, 0, sj));
    } else if (h->bf16) // the fp32 fragment images are not used by the bf16 path (its own are repacked by pack_U_bf16)
    {   // ... except the scatter-form backward image, whose 8-byte elements are the four rows an Adagrad thread holds
        RUN(K_ADAGRAD, adagrad(h->P, h->dP, h->mem, h->pl.total, (float)lr, h->pl.U, h->cfg.N, nullptr, nullptr, h->st, nullptr, nullptr,
                               nullptr, 0, 0, 0, nullptr, 0, 0, 0, h->bwd_scatter16 ? h->Ubwd6b : nullptr,
                               bwd_scatter_bf16_units(h->cfg.N), h->Why_b, h->WhyT_b, h->pl.Why, sj,
                               h->fwd_halves16 ? h->Ufwd6b : nullptr, fwd_halves_bf16_units(h->cfg.N)));
        h->packed6b = h->bwd_scatter16;
        static const bool quad_off = getenv("LSTM_HIP_ADAGRAD_QUAD") && atoi(getenv("LSTM_HIP_ADAGRAD_QUAD")) == 0; // (A/B switch of adagrad())
        h->packedf6b = h->fwd_halves16 && !quad_off;
    }
    else
    RUN(K_ADAGRAD, adagrad(h->P, h->dP, h->mem, h->pl.total, (float)lr, h->pl.U, h->cfg.N, h->Ufwd4 ? nullptr : h->Ufwd,
                           h->Ubwd4 ? nullptr : h->Ubwd, h->st, h->Ubwd4, h->Ufwd4, nullptr, 0, 0, 0, nullptr, 0, 0, h->half_forms(), nullptr, 0,
                           nullptr, nullptr, 0, sj));
    if (sj) h->pre_slid = true;
    h->carry_slide = false;
    h->packed = true; // the fp32 U images were refreshed by the same launch (the bf16 path has none)
    h->packed16 = false;
    h->why_packed = h->bf16; // (the bf16 path's Adagrad launch has just rewritten both bf16 copies of Why)
    return 0;
}

} // namespace

static int create_body(lstm_hip_ctx *h, const lstm_hip_config *cfg, const hipDeviceProp_t &prop);

extern "C" {

const char *lstm_hip_last_error(void) { return g_err; }

size_t lstm_hip_param_count(int32_t N, int32_t M) { return ParamLayout::make(N, M).total; }

int lstm_hip_device_info(int32_t device, char name[64], int32_t *cus, int32_t *clock_mhz) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name) snprintf(name, 64, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    return 0;
}

int lstm_hip_create(const lstm_hip_config *cfg, lstm_hip_t **out) {
    if (!cfg || !out) return fail(LSTM_HIP_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->M != LSTM_HIP_VOCAB) return fail(LSTM_HIP_EINVAL, "M must be %d (got %d)", LSTM_HIP_VOCAB, cfg->M);
    if (cfg->N < 16 || cfg->N % 16 != 0) return fail(LSTM_HIP_EINVAL, "N must be a positive multiple of 16 (got %d)", cfg->N);
    if (cfg->S < 2) return fail(LSTM_HIP_EINVAL, "S must be >= 2 (got %d)", cfg->S);
    if (cfg->B < 1) return fail(LSTM_HIP_EINVAL, "B must be >= 1 (got %d)", cfg->B);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LSTM_HIP_ENODEV, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(LSTM_HIP_ENODEV, "device %d out of range (%d visible)", cfg->device, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LSTM_HIP_ENODEV, "device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(cfg->device));
    // everything that can be refused is refused before the first allocation
    if (cfg->flags & LSTM_HIP_BF16_RECURRENCE) {
        if ((cfg->flags & LSTM_HIP_STEP_KERNELS) || cfg->N % 128 != 0 || cfg->N > 1024)
            return fail(LSTM_HIP_EINVAL, "LSTM_HIP_BF16_RECURRENCE needs the persistent engine and N a multiple of 128, <= 1024 (N=%d, B=%d)", cfg->N, cfg->B);
        if (cfg->B % 8 != 0)
            return fail(LSTM_HIP_EINVAL, "LSTM_HIP_BF16_RECURRENCE needs a multiple of 8 streams (16-byte aligned bf16 operand rows); B=%d", cfg->B);
        // the two-half forms (one workgroup per CU, 8-column groups) or, where a shape has none, the one-recurrence forms
        if (!(fwd_halves_bf16_supported(cfg->N, cfg->B, prop.multiProcessorCount) && bwd_scatter_bf16_supported(cfg->N, cfg->B, prop.multiProcessorCount)) &&
            !persistent_supported_bf16(cfg->N, cfg->B, prop.multiProcessorCount, false))
            return fail(LSTM_HIP_EINVAL, "LSTM_HIP_BF16_RECURRENCE: the bf16 recurrence grids for N=%d, B=%d are not co-resident on %d CUs",
                        cfg->N, cfg->B, prop.multiProcessorCount);
    }

    lstm_hip_ctx *h = new lstm_hip_ctx();
    const int rc = create_body(h, cfg, prop);
    if (rc != 0) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep)); // destroy must not overwrite the reason
        (void)lstm_hip_destroy(h);         // frees whatever had been allocated (every member is null-checked)
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out = h;
    return 0;
}

} // extern "C"

static int create_body(lstm_hip_ctx *h, const lstm_hip_config *cfg, const hipDeviceProp_t &prop) {
    h->cfg = *cfg;
    h->pl = ParamLayout::make(cfg->N, cfg->M);
    const size_t N = cfg->N, B = cfg->B, S = cfg->S, G4 = 4 * N;
    h->T = (int)((S - 1) * B);
    h->global_B = cfg->B;
    HIP_TRY(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
    HIP_TRY(hipEventCreate(&h->evt0)); // train_windows' elapsed time (ev0 / ev1 belong to the per-kernel profiling)
    HIP_TRY(hipEventCreate(&h->evt1));
    { // first use of timed events on the stream, here rather than inside somebody's measurement
        float ms = 0.0f;
        HIP_TRY(hipEventRecord(h->evt0, h->st));
        HIP_TRY(hipEventRecord(h->evt1, h->st));
        HIP_TRY(hipEventSynchronize(h->evt1));
        HIP_TRY(hipEventElapsedTime(&ms, h->evt0, h->evt1));
    }
    ALLOC(h->P, h->pl.total);
    ALLOC(h->dP, h->pl.total);
    ALLOC(h->mem, h->pl.total);
    ALLOC(h->Ufwd, N * N);
    ALLOC(h->Ubwd, N * N);
    ALLOC(h->H, N * B * S);
    ALLOC(h->C, N * B * S);
    ALLOC(h->G, G4 * B * S);
    ALLOC(h->DG, G4 * B * S);
    ALLOC(h->Y, 256 * B * S);
    ALLOC(h->Pr, 256 * B * S);
    ALLOC(h->DHy, N * B * S);
    ALLOC(h->dcnext, N * B);
    ALLOC(h->colloss, B * S);
    ALLOC(h->dby_part, (size_t)256 * softmax_parts(h->T));
    h->splits_dWhy = gemm_pick_splits(false, true, 256, (int)N, h->T, prop.multiProcessorCount);
    h->splits_dU = gemm_pick_splits(false, true, (int)G4, (int)N, h->T, prop.multiProcessorCount);
    if (cfg->flags & LSTM_HIP_BF16_RECURRENCE) {
        h->Tpad = (h->T + 63) / 64 * 64;
        h->SBpad = ((int)B + h->Tpad + 63) / 64 * 64;
        h->splits_dWhy = gemm_bf16_pick_splits(256, (int)N, h->Tpad);
        h->splits_dU = gemm_bf16_pick_splits((int)G4, (int)N, h->Tpad);
    }
    ALLOC(h->slabs, (size_t)h->splits_dWhy * 256 * N);
    ALLOC(h->slabs_dU, (size_t)h->splits_dU * G4 * N);
    ALLOC(h->dw_scratch, dW_scratch_bytes(h->T, (int)G4));
    ALLOC(h->xi, S * B);
    ALLOC(h->ti, S * B);
    ALLOC(h->Xr, S * B);
    ALLOC(h->Tr, S * B);
    ALLOC(h->head, 1);
    HIP_TRY(hipMemset(h->xi, 0xff, sizeof(int32_t) * S * B)); // -1: all-zero columns (opt:122,125)
    HIP_TRY(hipMemset(h->ti, 0xff, sizeof(int32_t) * S * B));
    HIP_TRY(hipMemset(h->Xr, 0xff, sizeof(int32_t) * S * B));
    HIP_TRY(hipMemset(h->Tr, 0xff, sizeof(int32_t) * S * B));
    ALLOC(h->d_loss, 1);
    ALLOC(h->pos, B);
    h->cnt_bytes = persistent_counter_bytes((int)S, (int)B);
    ALLOC(h->cnt, 2 * h->cnt_bytes / sizeof(unsigned));
    ALLOC(h->abortp, 4);
    // bf16 path: every product is a bf16 GEMM of its own, nothing is fused into the recurrence
    // one stream at hidden <= 128 (the reference's default shape): single-CU recurrences, unfused sums ("0": the multi-CU forms, A/B)
    const bool small_ok = !(cfg->flags & (LSTM_HIP_STEP_KERNELS | LSTM_HIP_BF16_RECURRENCE)) && small_recurrence_supported(cfg->N, cfg->B) &&
                          !(getenv("LSTM_HIP_SMALL") && atoi(getenv("LSTM_HIP_SMALL")) == 0);
    const bool want_fused = !(cfg->flags & (LSTM_HIP_NO_FUSED_GRADS | LSTM_HIP_BF16_RECURRENCE)) && cfg->N <= 512 && !small_ok; // larger N: one workgroup per CU no longer holds
    h->persistent = !(cfg->flags & LSTM_HIP_STEP_KERNELS) &&                          // the dW table beside the weights
                    persistent_supported(cfg->N, cfg->B, prop.multiProcessorCount, want_fused);
    h->n_cus = prop.multiProcessorCount;
    h->slide_in_adagrad = !(getenv("LSTM_HIP_SLIDE_IN_ADAGRAD") && atoi(getenv("LSTM_HIP_SLIDE_IN_ADAGRAD")) == 0);
    h->small = h->persistent && small_ok;
    h->bwd_cols = bwd_group_cols(cfg->N, cfg->B, prop.multiProcessorCount);
    if (cfg->flags & LSTM_HIP_BF16_RECURRENCE) { // refusals: lstm_hip_create, before anything is allocated
        h->persistent = true;                    // ... where the bf16 kernels' own grids were checked
        h->bf16 = true;
        h->bwd_cols = bwd_group_cols_bf16(cfg->N, cfg->B, prop.multiProcessorCount, false);
        ALLOC(h->Hb, S * B * N);
        ALLOC(h->DGb, S * B * G4);
        HIP_TRY(hipMalloc(&h->Ufwd16, (size_t)8 * N * N));
        HIP_TRY(hipMalloc(&h->Ubwd16, (size_t)8 * N * N));
        ALLOC(h->WhyT_b, 256 * N);
        ALLOC(h->Why_b, 256 * N);
        ALLOC(h->Ht_b, N * (size_t)h->SBpad);
        ALLOC(h->dYt_b, (size_t)256 * h->Tpad);
        ALLOC(h->DGt_b, G4 * (size_t)h->Tpad);
        ALLOC(h->dYb, (size_t)h->T * 256);
        // "0": the one-recurrence forms (A/B; per handle) -- where the shape has them
        const bool older = persistent_supported_bf16(cfg->N, cfg->B, prop.multiProcessorCount, false);
        const char *bh = getenv("LSTM_HIP_BWD_HALVES");
        h->bwd_scatter16 = !(older && bh && atoi(bh) == 0) && bwd_scatter_bf16_supported((int)N, (int)B, prop.multiProcessorCount);
        const char *fh = getenv("LSTM_HIP_FWD_HALVES");
        h->fwd_halves16 = !(older && fh && atoi(fh) == 0) && fwd_halves_bf16_supported((int)N, (int)B, prop.multiProcessorCount);
        if (h->fwd_halves16) {
            HIP_TRY(hipMalloc(&h->Ufwd6b, (size_t)8 * N * N));
            HIP_TRY(hipMalloc(&h->Hxb, sizeof(unsigned short) * fwd_halves_bf16_ring_halfwords((int)N, (int)B)));
            HIP_TRY(hipMemset(h->Hxb, 0xff, sizeof(unsigned short) * fwd_halves_bf16_ring_halfwords((int)N, (int)B)));
        }
        if (h->bwd_scatter16) {
            HIP_TRY(hipMalloc(&h->Ubwd6b, (size_t)8 * N * N));
            h->DGx_floats = bwd_scatter_bf16_ring_floats((int)N, (int)B, prop.multiProcessorCount);
            ALLOC(h->DGx, h->DGx_floats);
            HIP_TRY(hipMemset(h->DGx, 0xff, sizeof(float) * h->DGx_floats));
        }
    }
    if (h->persistent && bwd_uses_m4((int)N, h->bwd_cols, h->bf16) && !h->small) { // (the single-CU form reads the tile image Ubwd)
        ALLOC(h->Ubwd4, N * N);
        const char *bh = getenv("LSTM_HIP_BWD_HALVES");
        h->side_stream = !(getenv("LSTM_HIP_NO_SIDE_STREAM") && atoi(getenv("LSTM_HIP_NO_SIDE_STREAM")));
        // two-half (scatter) form wherever it exists; "0" selects the one-recurrence form (A/B), other values carry test /
        // tuning bits for the kernel (value >> 1 = its cfg word)
        const int bhv = bh ? atoi(bh) : 1;
        const bool wide = two_half_wide((int)N, (int)B, prop.multiProcessorCount); // (no one-recurrence form there)
        h->bwd_halves = bhv == 0 && !wide ? 0 : (bhv | 1) * (int)bwd_scatter_supported((int)N, (int)B, prop.multiProcessorCount, want_fused);
        if (h->bwd_halves) { // hand-off through a sentinel ring
            h->DGx_floats = bwd_ring_floats((int)N, (int)B);
            ALLOC(h->DGx, h->DGx_floats);
            HIP_TRY(hipMemset(h->DGx, 0xff, sizeof(float) * h->DGx_floats));
        }
    }
    if (h->persistent && !h->bf16 && fwd_uses_8col_form((int)N, (int)B, prop.multiProcessorCount)) {
        const char *pp = getenv("LSTM_HIP_FWD_HALVES"); // "0": one 8-column recurrence per workgroup (A/B; per handle)
        h->fwd_cols4 = fwd_uses_two_half_form((int)N, (int)B, prop.multiProcessorCount) &&
                       (!(pp && atoi(pp) == 0) || two_half_wide((int)N, (int)B, prop.multiProcessorCount));
        ALLOC(h->Ufwd4, N * N);
        ALLOC(h->Hx, fwd_ring_floats((int)N, (int)B));
        HIP_TRY(hipMemset(h->Hx, 0xff, sizeof(float) * fwd_ring_floats((int)N, (int)B)));
        // tuning knob (flat from 0 to 4 for the one-recurrence form; the two-half form polls without pause)
        h->poll_cfg = getenv("LSTM_HIP_FWD_POLL") ? atoi(getenv("LSTM_HIP_FWD_POLL")) : (h->fwd_cols4 ? 0 : 1);
    }
    if (h->persistent && want_fused && h->bwd_cols == 8)
    {
        h->gpart_cols = h->bwd_halves ? bwd_scatter_group_cols((int)N, (int)B, prop.multiProcessorCount) : h->bwd_cols;
        ALLOC(h->gpart, (size_t)((B + h->gpart_cols - 1) / h->gpart_cols) * bwd_partial_floats(cfg->N));
    }
    h->probe_overlap = getenv("LSTM_HIP_PROBE_OVERLAP") ? atoi(getenv("LSTM_HIP_PROBE_OVERLAP")) : 0;
    HIP_TRY(hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming));
    if (h->persistent && (cfg->flags & LSTM_HIP_DEBUG_STAMPS) &&
        ((cfg->N == 512 && h->Hx && h->Ubwd4) || (h->fwd_halves16 && h->bwd_scatter16)))
        ALLOC(h->stamps, 4 * S * 16);
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

extern "C" {

int lstm_hip_destroy(lstm_hip_t *h) {
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    if (h->eval_h) (void)lstm_hip_destroy(h->eval_h);
    if (h->st) (void)hipStreamSynchronize(h->st);
    if (h->st2) (void)hipStreamSynchronize(h->st2);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    void *bufs[] = {h->P, h->dP, h->mem, h->Ufwd, h->Ubwd, h->Ubwd4, h->Ufwd4, h->Hx, h->DGx, h->H, h->C, h->G, h->DG, h->Y, h->Pr, h->DHy, h->dcnext,
                    h->colloss, h->dby_part, h->slabs, h->slabs_dU, h->gpart, h->Hb, h->DGb, h->Ufwd16, h->Ubwd16, h->Ubwd6b, h->Ufwd6b, h->Hxb, h->WhyT_b, h->Why_b, h->Ht_b, h->dYt_b, h->DGt_b, h->dYb, h->dw_scratch, h->xi, h->ti, h->Xr, h->Tr, h->head, h->cnt, h->abortp, h->stamps, h->d_loss, h->d_losses, h->text, h->pos};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->h_losses) (void)hipHostFree(h->h_losses);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->evt0) (void)hipEventDestroy(h->evt0);
    if (h->evt1) (void)hipEventDestroy(h->evt1);
    for (hipEvent_t e : {h->ev_fork, h->ev_join, h->ev_mid})
        if (e) (void)hipEventDestroy(e);
    if (h->st2) (void)hipStreamDestroy(h->st2);
    if (h->st) (void)hipStreamDestroy(h->st);
    delete h;
    return 0;
}

static float *block_of(lstm_hip_ctx *h, int which) { return which == 0 ? h->P : which == 1 ? h->dP : which == 2 ? h->mem : nullptr; }

int lstm_hip_set_params(lstm_hip_t *h, int which, const float *host_block) {
    CHECK(h);
    float *dst = block_of(h, which);
    if (!dst || !host_block) return fail(LSTM_HIP_EINVAL, "set_params: bad block id %d or null pointer", which);
    HIP_TRY(hipMemcpyAsync(dst, host_block, sizeof(float) * h->pl.total, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    if (which == 0) h->packed = h->packed16 = h->packed6b = h->packedf6b = h->why_packed = false;
    return 0;
}
int lstm_hip_get_params(lstm_hip_t *h, int which, float *host_block) {
    CHECK(h);
    float *src = block_of(h, which);
    if (!src || !host_block) return fail(LSTM_HIP_EINVAL, "get_params: bad block id %d or null pointer", which);
    HIP_TRY(hipMemcpyAsync(host_block, src, sizeof(float) * h->pl.total, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return check_abort(h);
}

int lstm_hip_set_state(lstm_hip_t *h, int32_t t, const float *h_t, const float *c_t) {
    if (h) h->pre_slid = false; // (a window slid ahead by an interrupted loop is not the caller's window any more)
    CHECK(h);
    if (t < 0 || t >= h->cfg.S) return fail(LSTM_HIP_EINVAL, "set_state: t=%d outside [0,%d)", t, h->cfg.S);
    const size_t n = (size_t)h->cfg.N * h->cfg.B;
    if (h_t) HIP_TRY(hipMemcpyAsync(h->H + t * n, h_t, sizeof(float) * n, hipMemcpyHostToDevice, h->st));
    if (c_t) HIP_TRY(hipMemcpyAsync(h->C + t * n, c_t, sizeof(float) * n, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}
int lstm_hip_get_state(lstm_hip_t *h, int32_t t, float *h_t, float *c_t) {
    CHECK(h);
    if (t < 0 || t >= h->cfg.S) return fail(LSTM_HIP_EINVAL, "get_state: t=%d outside [0,%d)", t, h->cfg.S);
    const size_t n = (size_t)h->cfg.N * h->cfg.B;
    if (h_t) HIP_TRY(hipMemcpyAsync(h_t, h->H + t * n, sizeof(float) * n, hipMemcpyDeviceToHost, h->st));
    if (c_t) HIP_TRY(hipMemcpyAsync(c_t, h->C + t * n, sizeof(float) * n, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}
int lstm_hip_get_activations(lstm_hip_t *h, int32_t t, float *g_t, float *probs_t) {
    CHECK(h);
    if (t < 1 || t >= h->cfg.S) return fail(LSTM_HIP_EINVAL, "get_activations: t=%d outside [1,%d)", t, h->cfg.S);
    const size_t B = h->cfg.B, G4 = 4 * (size_t)h->cfg.N;
    if (g_t) HIP_TRY(hipMemcpyAsync(g_t, h->G + t * G4 * B, sizeof(float) * G4 * B, hipMemcpyDeviceToHost, h->st));
    if (probs_t) HIP_TRY(hipMemcpyAsync(probs_t, h->Pr + t * 256 * B, sizeof(float) * 256 * B, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}

int lstm_hip_set_window(lstm_hip_t *h, const int32_t *xi, const int32_t *ti) {
    if (h) h->pre_slid = false; // (a window slid ahead by an interrupted loop is not the caller's window any more)
    CHECK(h);
    if (!xi || !ti) return fail(LSTM_HIP_EINVAL, "set_window: null pointer");
    const size_t n = (size_t)h->cfg.S * h->cfg.B;
    for (size_t i = 0; i < n; i++)
        if (xi[i] >= LSTM_HIP_VOCAB || ti[i] >= LSTM_HIP_VOCAB)
            return fail(LSTM_HIP_EINVAL, "set_window: index %d/%d at %zu is >= %d", xi[i], ti[i], i, LSTM_HIP_VOCAB);
    HIP_TRY(hipMemcpyAsync(h->xi, xi, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemcpyAsync(h->ti, ti, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemcpyAsync(h->Xr, xi, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->st)); // rings, head = 0
    HIP_TRY(hipMemcpyAsync(h->Tr, ti, sizeof(int32_t) * n, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemsetAsync(h->head, 0, sizeof(int32_t), h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}
int lstm_hip_set_inputs_dense(lstm_hip_t *h, const float *h0, const float *c0, const float *x, const float *target) {
    if (h) h->pre_slid = false; // (a window slid ahead by an interrupted loop is not the caller's window any more)
    CHECK(h);
    if (!x || !target) return fail(LSTM_HIP_EINVAL, "set_inputs_dense: null x or target");
    const size_t cols = (size_t)h->cfg.S * h->cfg.B;
    std::vector<int32_t> xi(cols), ti(cols);
    for (int which = 0; which < 2; which++) {
        const float *m = which ? target : x;
        std::vector<int32_t> &out = which ? ti : xi;
        for (size_t c = 0; c < cols; c++) {
            int32_t idx = -1;
            for (int r = 0; r < LSTM_HIP_VOCAB; r++) {
                const float v = m[c * LSTM_HIP_VOCAB + r];
                if (v == 0.0f) continue;
                if (v != 1.0f || idx >= 0)
                    return fail(LSTM_HIP_EINVAL, "set_inputs_dense: column %zu of %s is not one-hot (row %d holds %g)", c,
                                which ? "target" : "x", r, (double)v);
                idx = r;
            }
            out[c] = idx;
        }
    }
    int rc = lstm_hip_set_window(h, xi.data(), ti.data());
    if (rc) return rc;
    if (h0 || c0) return lstm_hip_set_state(h, 0, h0, c0);
    return 0;
}
int lstm_hip_get_window(lstm_hip_t *h, int32_t *xi, int32_t *ti) {
    CHECK(h);
    const size_t n = (size_t)h->cfg.S * h->cfg.B;
    if (xi) HIP_TRY(hipMemcpyAsync(xi, h->xi, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->st));
    if (ti) HIP_TRY(hipMemcpyAsync(ti, h->ti, sizeof(int32_t) * n, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}
int lstm_hip_reset_window(lstm_hip_t *h) {
    CHECK(h);
    const size_t n = (size_t)h->cfg.S * h->cfg.B;
    HIP_TRY(hipMemsetAsync(h->xi, 0xff, sizeof(int32_t) * n, h->st));
    HIP_TRY(hipMemsetAsync(h->ti, 0xff, sizeof(int32_t) * n, h->st));
    HIP_TRY(hipMemsetAsync(h->Xr, 0xff, sizeof(int32_t) * n, h->st));
    HIP_TRY(hipMemsetAsync(h->Tr, 0xff, sizeof(int32_t) * n, h->st));
    HIP_TRY(hipMemsetAsync(h->head, 0, sizeof(int32_t), h->st));
    return 0;
}

static int slide_state(lstm_hip_ctx *h) {
    const size_t n = (size_t)h->cfg.N * h->cfg.B;
    if (h->cfg.S < 2) return 0;
    HIP_TRY(hipMemcpyAsync(h->H, h->H + n, sizeof(float) * n, hipMemcpyDeviceToDevice, h->st));
    HIP_TRY(hipMemcpyAsync(h->C, h->C + n, sizeof(float) * n, hipMemcpyDeviceToDevice, h->st));
    return 0;
}
int lstm_hip_slide_state(lstm_hip_t *h) {
    CHECK(h);
    return slide_state(h);
}

int lstm_hip_forward(lstm_hip_t *h) {
    CHECK(h);
    return do_forward(h);
}
int lstm_hip_loss(lstm_hip_t *h, double *loss_bits) {
    CHECK(h);
    if (!loss_bits) return fail(LSTM_HIP_EINVAL, "loss: null pointer");
    if (!h->fwd_done) return fail(LSTM_HIP_ESTATE, "loss called before forward");
    RUN(K_LOSS, loss_reduce(loss_src(h), loss_steps(h), h->cfg.B, h->global_B, h->d_loss, nullptr, 0, nullptr, h->st, loss_scale(h)));
    HIP_TRY(hipMemcpyAsync(loss_bits, h->d_loss, sizeof(double), hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return check_abort(h);
}
int lstm_hip_backward(lstm_hip_t *h) {
    CHECK(h);
    return do_backward(h);
}
int lstm_hip_adagrad(lstm_hip_t *h, double learning_rate) {
    CHECK(h);
    return do_adagrad(h, learning_rate);
}

int lstm_hip_comm_unique_id(uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES]) {
    int rc = rccl_load();
    if (rc) return rc;
    UniqueId u;
    rc = g_rccl.GetUniqueId(&u);
    if (rc != 0) return fail(LSTM_HIP_ERCCL, "ncclGetUniqueId failed (%d)", rc);
    memcpy(id, u.internal, LSTM_HIP_UNIQUE_ID_BYTES);
    return 0;
}
int lstm_hip_comm_init(lstm_hip_t *h, const uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank) {
    CHECK(h);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(LSTM_HIP_EINVAL, "comm_init: rank %d of %d", rank, nranks);
    int rc = rccl_load();
    if (rc) return rc;
    UniqueId u;
    memcpy(u.internal, id, LSTM_HIP_UNIQUE_ID_BYTES);
    rc = g_rccl.CommInitRank(&h->comm, nranks, u, rank);
    if (rc != 0) {
        h->comm = nullptr;
        return fail(LSTM_HIP_ERCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    }
    h->nranks = nranks;
    h->rank = rank;
    return 0;
}
int lstm_hip_allreduce_grads(lstm_hip_t *h) {
    CHECK(h);
    return do_allreduce(h);
}

int lstm_hip_set_text(lstm_hip_t *h, const uint8_t *text, size_t len) {
    if (h) h->pre_slid = false; // (a window slid ahead by an interrupted loop is not the caller's window any more)
    CHECK(h);
    if (!text || len <= (size_t)h->cfg.S) return fail(LSTM_HIP_EINVAL, "set_text: need more than S=%d bytes (got %zu)", h->cfg.S, len);
    HIP_TRY(hipStreamSynchronize(h->st));
    if (h->text) HIP_TRY(hipFree(h->text));
    h->text = nullptr;
    HIP_TRY(hipMalloc((void **)&h->text, len));
    HIP_TRY(hipMemcpy(h->text, text, len, hipMemcpyHostToDevice));
    h->text_len = len;
    return 0;
}
int lstm_hip_set_cursors(lstm_hip_t *h, const uint64_t *pos) {
    if (h) h->pre_slid = false; // (a window slid ahead by an interrupted loop is not the caller's window any more)
    CHECK(h);
    if (!pos) return fail(LSTM_HIP_EINVAL, "set_cursors: null pointer");
    if (!h->text) return fail(LSTM_HIP_ESTATE, "set_cursors before set_text");
    for (int b = 0; b < h->cfg.B; b++)
        if (pos[b] >= h->text_len) return fail(LSTM_HIP_EINVAL, "set_cursors: pos[%d]=%llu >= len %llu", b, (unsigned long long)pos[b], (unsigned long long)h->text_len);
    HIP_TRY(hipMemcpyAsync(h->pos, pos, sizeof(uint64_t) * h->cfg.B, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}
int lstm_hip_get_cursors(lstm_hip_t *h, uint64_t *pos) {
    CHECK(h);
    if (!pos) return fail(LSTM_HIP_EINVAL, "get_cursors: null pointer");
    HIP_TRY(hipMemcpyAsync(pos, h->pos, sizeof(uint64_t) * h->cfg.B, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}
int lstm_hip_set_stride(lstm_hip_t *h, int32_t stride, int32_t carry_col) {
    if (h) h->pre_slid = false; // (a window slid ahead by an interrupted loop is not the caller's window any more)
    if (!h) return fail(LSTM_HIP_EINVAL, "null handle");
    if (stride < 1 || stride >= h->cfg.S || carry_col < 0 || carry_col >= h->cfg.S)
        return fail(LSTM_HIP_EINVAL, "set_stride: need 1 <= stride < S and 0 <= carry_col < S (got %d, %d)", stride, carry_col);
    h->stride = stride;
    h->carry_col = carry_col;
    return 0;
}
int lstm_hip_set_global_batch(lstm_hip_t *h, int32_t global_B) {
    if (!h || global_B < h->cfg.B) return fail(LSTM_HIP_EINVAL, "set_global_batch: %d < local B", global_B);
    h->global_B = global_B;
    return 0;
}

int lstm_hip_set_loss_mode(lstm_hip_t *h, int32_t mode) {
    if (!h || (mode != LSTM_HIP_LOSS_ALL_STEPS_BITS && mode != LSTM_HIP_LOSS_LAST_STEP_NATS && mode != LSTM_HIP_LOSS_LAST_STEP_BITS))
        return fail(LSTM_HIP_EINVAL, "set_loss_mode: unknown mode %d", mode);
    h->loss_mode = mode;
    return 0;
}

int lstm_hip_train_windows(lstm_hip_t *h, int64_t count, double learning_rate, double *losses, float *elapsed_ms) {
    CHECK(h);
    if (count < 0) return fail(LSTM_HIP_EINVAL, "train_windows: count < 0");
    if (!h->text) return fail(LSTM_HIP_ESTATE, "train_windows before set_text/set_cursors");
    if (count > h->losses_cap) {
        const int64_t cap = count < 8192 ? 8192 : count;
        HIP_TRY(hipStreamSynchronize(h->st));
        if (h->d_losses) HIP_TRY(hipFree(h->d_losses));
        if (h->h_losses) HIP_TRY(hipHostFree(h->h_losses));
        h->d_losses = nullptr;
        h->h_losses = nullptr;
        h->losses_cap = 0;
        HIP_TRY(hipMalloc((void **)&h->d_losses, sizeof(double) * cap));
        HIP_TRY(hipHostMalloc((void **)&h->h_losses, sizeof(double) * cap, hipHostMallocDefault));
        h->losses_cap = cap;
    }
    // the handle's own event pair (made and exercised once at create: the first timed record on a stream costs ~0.5 ms of
    // device time, which a 20-window measurement would carry)
    if (elapsed_ms) HIP_TRY(hipEventRecord(h->evt0, h->st));
    h->in_loop = true;
    struct LoopGuard { // leaves the loop state clean on every return path
        lstm_hip_ctx *h;
        bool completed = false;
        ~LoopGuard() {
            h->in_loop = false;
            if (completed) return;
            // error exit somewhere inside a window: nothing of that window may leak into a later standalone call
            h->fold_pending = false;
            h->early_reduced = false;
            h->dU_reduced = 0;
            h->n_slabs_dU = 0;
            h->dby_done = false;
            h->fwd_done = false;
            h->carry_slide = false;
            if (h->st2) (void)hipStreamSynchronize(h->st2); // side-stream work of the broken window (folds, early all-reduce)
            if (h->st) (void)hipStreamSynchronize(h->st);
        }
    } guard{h};
    for (int64_t i = 0; i < count; i++) {
        // (from the second window on the slide has been done by the previous window's Adagrad launch, in extra workgroups)
        if (!h->pre_slid)
            RUN(K_SLIDE, slide_window(h->text, h->text_len, h->pos, h->Xr, h->Tr, h->head, h->xi, h->ti, h->H, h->C,
                                      h->cfg.S, h->cfg.B, h->cfg.N, h->stride, h->carry_col, h->st));
        h->pre_slid = false;
        int rc = 0;
        if ((rc = do_forward(h))) return rc;
        RUN(K_LOSS, loss_reduce(loss_src(h), loss_steps(h), h->cfg.B, h->global_B, h->d_losses + i, h->dby_part,
                                h->n_dby_parts, h->dP + h->pl.by, h->st, loss_scale(h)));
        h->dby_done = true;
        if ((rc = do_backward(h))) return rc;
        if ((rc = do_allreduce(h))) return rc;
        // not behind the last window (the handle is left on the window it trained on) and not in a profiling pass
        // ... and only for windows of up to 2 048 columns: the slide's one window-building workgroup has 256 threads there
        // instead of 1 024 and outlasts the Adagrad workgroups at the headline shape (6 400 columns: 0.660 -> 0.665 ms, while
        // configs[1] gains 4 us, configs[0] 2, configs[4] 3)
        h->carry_slide = i + 1 < count && !h->profiling && h->slide_in_adagrad && (int64_t)h->cfg.S * h->cfg.B <= 2048;
        if ((rc = do_adagrad(h, learning_rate))) return rc;
    }
    if (elapsed_ms) {
        HIP_TRY(hipEventRecord(h->evt1, h->st));
        HIP_TRY(hipEventSynchronize(h->evt1));
        HIP_TRY(hipEventElapsedTime(elapsed_ms, h->evt0, h->evt1));
    }
    if (losses && count > 0)
        HIP_TRY(hipMemcpyAsync(h->h_losses, h->d_losses, sizeof(double) * count, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    if (losses && count > 0) std::memcpy(losses, h->h_losses, sizeof(double) * count);
    guard.completed = true;
    return check_abort(h);
}

// test(), OV/lstm_eigen_class_CUDA/lstm.cc:661-720: one stream from h = c = 0, bits/char over the text.
// Where the persistent forward recurrence exists for this hidden size, the text is run through it in
// chunks on an internal B = 1 handle (the carry moves from the last column of a chunk to column 0 of the
// next); otherwise by the single-workgroup kernel.
// internal B = 1 handle behind the evaluator and the sampler (its own copy of the parameters and their fragment images)
static const int AUX_S = 129; // 128 characters per evaluator chunk
static int ensure_aux_handle(lstm_hip_ctx *h) {
    if (h->eval_h) return 0;
    lstm_hip_config c = h->cfg;
    c.S = AUX_S;
    c.B = 1;
    c.flags = (h->cfg.flags & LSTM_HIP_FAST_MATH) | LSTM_HIP_NO_FUSED_GRADS;
    return lstm_hip_create(&c, &h->eval_h);
}

int lstm_hip_eval_bits(lstm_hip_t *h, const uint8_t *text, size_t len, double *bits_per_char) {
    CHECK(h);
    if (!text || len < 2 || !bits_per_char) return fail(LSTM_HIP_EINVAL, "eval_bits: need >= 2 bytes and an output pointer");
    const int N = h->cfg.N;
    if (!h->persistent || (h->cfg.flags & LSTM_HIP_STEP_KERNELS)) {
        uint8_t *d_text = nullptr;
        struct Free {
            uint8_t *&p;
            ~Free() {
                if (p) (void)hipFree(p);
            }
        } free_text{d_text};
        HIP_TRY(hipMalloc((void **)&d_text, len));
        HIP_TRY(hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, h->st));
        eval_bits(h->P, N, d_text, len, h->d_loss, nullptr, h->st);
        double sum = 0.0;
        HIP_TRY(hipMemcpyAsync(&sum, h->d_loss, sizeof(double), hipMemcpyDeviceToHost, h->st));
        HIP_TRY(hipStreamSynchronize(h->st));
        *bits_per_char = sum / (double)(len - 1);
        return 0;
    }
    HIP_TRY(hipStreamSynchronize(h->st));
    const int Se = AUX_S;
    {
        int rc = ensure_aux_handle(h);
        if (rc) return rc;
    }
    lstm_hip_ctx *e = h->eval_h;
    HIP_TRY(hipMemcpy(e->P, h->P, sizeof(float) * h->pl.total, hipMemcpyDeviceToDevice));
    e->packed = false;
    HIP_TRY(hipMemset(e->H, 0, sizeof(float) * N)); // h = c = 0 (reset_std = 0, lstm.cc:45,676-677)
    HIP_TRY(hipMemset(e->C, 0, sizeof(float) * N));
    std::vector<int32_t> xi(Se), ti(Se);
    double sum = 0.0;
    for (size_t pos = 0; pos + 1 < len; pos += Se - 1) {
        const size_t steps = std::min<size_t>(Se - 1, len - 1 - pos);
        xi[0] = ti[0] = -1;
        for (int t = 1; t < Se; t++) {
            const bool in = (size_t)t <= steps;
            xi[t] = in ? (int32_t)text[pos + t - 1] : -1; // past the end: empty columns, no loss
            ti[t] = in ? (int32_t)text[pos + t] : -1;
        }
        int rc = lstm_hip_set_window(e, xi.data(), ti.data());
        if (rc) return rc;
        if ((rc = do_forward(e))) return rc;
        double part = 0.0;
        if ((rc = lstm_hip_loss(e, &part))) return rc;
        sum += part;
        // carry: the state after the chunk's last real character becomes column 0
        HIP_TRY(hipMemcpyAsync(e->H, e->H + (size_t)steps * N, sizeof(float) * N, hipMemcpyDeviceToDevice, e->st));
        HIP_TRY(hipMemcpyAsync(e->C, e->C + (size_t)steps * N, sizeof(float) * N, hipMemcpyDeviceToDevice, e->st));
    }
    HIP_TRY(hipStreamSynchronize(e->st));
    *bits_per_char = sum / (double)(len - 1);
    return 0;
}

int lstm_hip_sample(lstm_hip_t *h, float *h0, float *c0, const double *u, int32_t count, uint8_t *out) {
    CHECK(h);
    if (!h0 || !c0 || !u || !out || count < 0) return fail(LSTM_HIP_EINVAL, "sample: null pointer or negative count");
    const int N = h->cfg.N;
    float *d_hc = nullptr;
    double *d_u = nullptr;
    uint8_t *d_out = nullptr;
    struct Scratch { // released on every return path
        float *&a;
        double *&b;
        uint8_t *&c;
        ~Scratch() {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            if (c) (void)hipFree(c);
        }
    } scratch{d_hc, d_u, d_out};
    HIP_TRY(hipMalloc((void **)&d_hc, sizeof(float) * 2 * N));
    HIP_TRY(hipMalloc((void **)&d_u, sizeof(double) * (count + 1)));
    HIP_TRY(hipMalloc((void **)&d_out, (size_t)count + 1));
    HIP_TRY(hipMemcpyAsync(d_hc, h0, sizeof(float) * N, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemcpyAsync(d_hc + N, c0, sizeof(float) * N, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemcpyAsync(d_u, u, sizeof(double) * count, hipMemcpyHostToDevice, h->st));
    if (h->persistent && !(h->cfg.flags & LSTM_HIP_STEP_KERNELS) && count > 0) {
        // Multi-workgroup path: per character one k_sample_head (probabilities + CDF walk, one workgroup) and one
        // k_fwd_step launch (the recurrent product over N/4 workgroups) on the internal B = 1 handle the evaluator uses;
        // the sampled byte never leaves the device.  (k_sample does the whole 4N x N product in ONE workgroup:
        // 395 us per character at N = 512.)
        HIP_TRY(hipStreamSynchronize(h->st));
        int rc = ensure_aux_handle(h);
        if (rc) return rc;
        lstm_hip_ctx *e = h->eval_h;
        HIP_TRY(hipMemcpyAsync(e->P, h->P, sizeof(float) * h->pl.total, hipMemcpyDeviceToDevice, e->st));
        pack_U(e->P + e->pl.U, e->Ufwd, e->Ubwd, N, e->st);
        e->packed = true;
        HIP_TRY(hipMemcpyAsync(e->H, d_hc, sizeof(float) * N, hipMemcpyDeviceToDevice, e->st));
        HIP_TRY(hipMemcpyAsync(e->C, d_hc + N, sizeof(float) * N, hipMemcpyDeviceToDevice, e->st));
        const bool fast = (h->cfg.flags & LSTM_HIP_FAST_MATH) != 0;
        int cur = 0;
        for (int i = 0; i < count; i++) {
            sample_head(e->P + e->pl.Why, e->P + e->pl.by, N, e->H + (size_t)cur * N, d_u + i, d_out + i, e->xi, e->st);
            fwd_step(e->Ufwd, e->P + e->pl.W, e->P + e->pl.b, e->H + (size_t)cur * N, e->C + (size_t)cur * N,
                     e->H + (size_t)(cur ^ 1) * N, e->C + (size_t)(cur ^ 1) * N, e->G, e->xi, N, 1, fast, e->st);
            cur ^= 1;
        }
        HIP_TRY(hipMemcpyAsync(d_hc, e->H + (size_t)cur * N, sizeof(float) * N, hipMemcpyDeviceToDevice, e->st));
        HIP_TRY(hipMemcpyAsync(d_hc + N, e->C + (size_t)cur * N, sizeof(float) * N, hipMemcpyDeviceToDevice, e->st));
        HIP_TRY(hipStreamSynchronize(e->st));
        e->fwd_done = false;
    } else
    sample(h->P, N, d_hc, d_u, count, d_out, nullptr, h->st);
    HIP_TRY(hipMemcpyAsync(h0, d_hc, sizeof(float) * N, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipMemcpyAsync(c0, d_hc + N, sizeof(float) * N, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipMemcpyAsync(out, d_out, count, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}

int lstm_hip_debug_stamps(lstm_hip_t *h, uint64_t *out, size_t count) {
    CHECK(h);
    if (!h->stamps) return fail(LSTM_HIP_ESTATE, "handle was not created with LSTM_HIP_DEBUG_STAMPS on a shape whose two-half forms carry stamps (fp32: hidden 512; bf16: any)");
    const size_t have = (size_t)4 * h->cfg.S * 16;
    HIP_TRY(hipMemcpyAsync(out, h->stamps, sizeof(uint64_t) * (count < have ? count : have), hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return 0;
}

int lstm_hip_synchronize(lstm_hip_t *h) {
    CHECK(h);
    HIP_TRY(hipStreamSynchronize(h->st));
    return check_abort(h);
}
int lstm_hip_set_profiling(lstm_hip_t *h, int32_t on) {
    if (!h) return fail(LSTM_HIP_EINVAL, "null handle");
    h->profiling = on != 0;
    return 0;
}
int lstm_hip_kernel_stat_count(lstm_hip_t *) { return K_COUNT; }
int lstm_hip_kernel_stat(lstm_hip_t *h, int32_t idx, const char **name, int64_t *launches, double *total_ms) {
    if (!h || idx < 0 || idx >= K_COUNT) return fail(LSTM_HIP_EINVAL, "kernel_stat: bad index %d", idx);
    if (name) *name = kKernelNames[idx];
    if (launches) *launches = h->launches[idx];
    if (total_ms) *total_ms = h->total_ms[idx];
    return 0;
}
int lstm_hip_reset_kernel_stats(lstm_hip_t *h) {
    if (!h) return fail(LSTM_HIP_EINVAL, "null handle");
    memset(h->launches, 0, sizeof(h->launches));
    memset(h->total_ms, 0, sizeof(h->total_ms));
    return 0;
}

} // extern "C"
