/*
 * lstm_hip.h -- C ABI of the MI355X (gfx950) LSTM forward/BPTT/Adagrad path.
 *
 * This is the drop-in boundary for the hot path of krocki/Eigen-LSTM.  The reference has exactly
 * one host<->device seam: the twelve `#ifdef __GPU__` sites of
 * OV/lstm_eigen_class_CUDA/lstm.cc (50,107,156,192,273,316,328,335,362,374,379,389), which talk to
 * the device through cuParameters / cuLSTM<S> / cuda_adagrad and six copy helpers
 * (OV/lstm_eigen_class_CUDA/cu_lstm.h).  Every entry point below names the member it replaces
 * (OV/ = /root/reference/optimized-obsfuscated_versions).  Plain pointers and sizes only; no C++ or
 * torch types.  All matrices are column-major fp32, as Eigen's MatrixXf::data() is
 * (cu_matrix.cu:93-101 copies it verbatim).
 *
 * Differences from the reference seam, on purpose:
 *   - one-hot matrices x[t], target[t] (M x B floats each) cross the boundary as int32 indices
 *     [S x B]; index < 0 is the all-zero column (OV/lstm_eigen_opt/lstm.cc:122,125);
 *   - the five parameter tensors travel as ONE flat block [W | U | b | Why | by] (also the RCCL
 *     all-reduce payload);
 *   - errors are returned (0 = ok, <0 = LSTM_HIP_E*), never printed-and-ignored
 *     (cu_matrix.cu:16-19,159-162); lstm_hip_last_error() gives the text;
 *   - the whole i-loop body can run on the device (lstm_hip_train_windows) so nothing is copied
 *     per iteration (the reference moves 7*S matrices each way, lstm.cc:274,317,375).
 *
 * Thread-safety: a handle is used by one host thread at a time; different handles are independent.
 */
#ifndef LSTM_HIP_H_
#define LSTM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSTM_HIP_OK 0
#define LSTM_HIP_EINVAL (-1)   /* bad argument / unsupported shape */
#define LSTM_HIP_EHIP (-2)     /* a HIP runtime call failed */
#define LSTM_HIP_ENODEV (-3)   /* no usable gfx950 device */
#define LSTM_HIP_ERCCL (-4)    /* RCCL missing or a collective failed */
#define LSTM_HIP_ESTATE (-5)   /* call sequence error (e.g. backward before forward) */

#define LSTM_HIP_VOCAB 256     /* M: raw bytes, R/lstm.cc:55 */

/* flags for lstm_hip_config.flags */
#define LSTM_HIP_FAST_MATH 1u      /* v_exp/v_rcp based sigmoid/tanh (the reference's --use_fast_math build,
                                      OV/lstm_eigen_class_CUDA/Makefile:53-66); default is libm-accurate */
                                   /* (bits 2u, 8u, 32u: flags of earlier versions, now ignored) */
#define LSTM_HIP_STEP_KERNELS 4u   /* one launch per timestep (baseline engine) instead of the persistent
                                      recurrence kernels */

#define LSTM_HIP_BF16_RECURRENCE 128u /* the bf16 MFMA path (BASELINE configs[4]): bf16 operands, fp32 accumulate, in the two
                                      recurrent products (U, and the h / dg hand-off) and in the four time-batched ones
                                      (Why*h, Why^T*dy, dy*h^T, dg*h^T); fp32 master weights, biases, elementwise math,
                                      dW/db/dby and Adagrad.  Needs N % 128 == 0, N <= 1024, B % 8 == 0 */
#define LSTM_HIP_NO_FUSED_GRADS 64u  /* compute dU/dW/db after the backward recurrence (GEMM + sorted segment sums)
                                      instead of accumulating them inside it */
#define LSTM_HIP_DEBUG_STAMPS 16u    /* diagnostic builds of both recurrences (N = 512, 8-column forms) that record
                                      s_memtime at marked points of every step; see lstm_hip_debug_stamps */

typedef struct lstm_hip_ctx lstm_hip_t; /* opaque: cuParameters p,d,m + cuLSTM<S> in one object */

typedef struct lstm_hip_config {
    int32_t N;       /* hidden size, multiple of 16                       R/lstm.cc:53 */
    int32_t M;       /* vocabulary, must be LSTM_HIP_VOCAB                R/lstm.cc:55 */
    int32_t S;       /* window columns; S-1 timesteps per window, S >= 2  R/lstm.cc:57 */
    int32_t B;       /* concurrent streams on THIS device                 OV/lstm_eigen_opt/lstm.cc:56 */
    int32_t device;  /* HIP device ordinal (reference: cudaSetDevice(4), lstm.cc:51) */
    uint32_t flags;  /* LSTM_HIP_* */
} lstm_hip_config;

/* ---- lifetime: cuParameters(M,N) x3 + cuLSTM<S>(M,N,B) ctor/dtor, cu_lstm.h:24-42,83-144;
 *      init_cublas/teardown_cublas, lstm.cc:50-54,389-391.  Parameters, gradients, Adagrad memory
 *      and all window state start zeroed (cuParameters::zero, cuLSTM::reset). */
int lstm_hip_create(const lstm_hip_config *cfg, lstm_hip_t **out);
int lstm_hip_destroy(lstm_hip_t *h);
const char *lstm_hip_last_error(void);
/* number of floats in the flat block: 4N*M + 4N*N + 4N + M*N + M */
size_t lstm_hip_param_count(int32_t N, int32_t M);

/* ---- copy_parameters_to_device / copy_parameters_to_host, cu_lstm.h:307-325.
 *      `which`: 0 = parameters p, 1 = gradients d, 2 = Adagrad memory m.  Host block layout
 *      [W (4N x M) | U (4N x N) | b (4N) | Why (M x N) | by (M)], each column-major. */
int lstm_hip_set_params(lstm_hip_t *h, int which, const float *host_block);
int lstm_hip_get_params(lstm_hip_t *h, int which, float *host_block);

/* ---- copy_lstm_to_device / copy_lstm_to_host / copy_context_to_host, cu_lstm.h:337-396.
 *      State column t (0 <= t < S) of h and c, each N x B column-major.  Either pointer may be NULL. */
int lstm_hip_set_state(lstm_hip_t *h, int32_t t, const float *h_t, const float *c_t);
int lstm_hip_get_state(lstm_hip_t *h, int32_t t, float *h_t, float *c_t);
/* g[t] (4N x B, post-activation gates [i;o;f;u]) and probs[t] (M x B); t in [1,S).  NULL = skip.
 * For lock-step comparison (compare_lstm_states, cu_lstm.h:398-415). */
int lstm_hip_get_activations(lstm_hip_t *h, int32_t t, float *g_t, float *probs_t);

/* ---- copy_inputs_to_device, cu_lstm.h:364-377: the window's inputs and targets as indices,
 *      xi[t*B+b], ti[t*B+b], t in [0,S) (row 0 is never read, as in the reference). */
int lstm_hip_set_window(lstm_hip_t *h, const int32_t *xi, const int32_t *ti);
/* the same call with the reference's own operands: h[0], c[0] (N x B each, may be NULL = leave as is) and the dense one-hot
 * matrices x[t], target[t] (M x B each, column-major, t = 0..S-1 back to back: S*B columns of M floats).  Every column must
 * be all-zero or exactly one 1.0f among zeros (what the reference's encoder produces, R/lstm.cc:169-170,
 * OV/lstm_eigen_opt/lstm.cc:199-212); anything else is LSTM_HIP_EINVAL.  For bit-faithful lock-step tests against code
 * that holds the dense form. */
int lstm_hip_set_inputs_dense(lstm_hip_t *h, const float *h0, const float *c0, const float *x, const float *target);
/* the device-side part of the slide (OV/lstm_eigen_opt/lstm.cc:205-206): h[0] <- h[1], c[0] <- c[1] */
int lstm_hip_slide_state(lstm_hip_t *h);

/* ---- cuLSTM::forward, cu_lstm.h:162-201 (R/lstm.cc:173-201): t = 1..S-1 */
int lstm_hip_forward(lstm_hip_t *h);
/* ---- cuLSTM::calculate_loss, cu_lstm.h:203-215, with the root file's semantics (every step
 *      counts, R/lstm.cc:204-207; /B per OV/lstm_eigen_opt/lstm.cc:249): sum_t (sum_b -log2 p)/B */
int lstm_hip_loss(lstm_hip_t *h, double *loss_bits);
/* ---- cuLSTM::backward, cu_lstm.h:216-275 (R/lstm.cc:214-257): zero d, BPTT t = S-1..1 */
int lstm_hip_backward(lstm_hip_t *h);
/* ---- cuda_adagrad, cu_lstm.h:417-432 (R/lstm.cc:261-272): m += d.*d; p -= lr*d./sqrt(m+1e-10) */
int lstm_hip_adagrad(lstm_hip_t *h, double learning_rate);

/* ---- data-parallel exchange (new; the reference is single-device).  One SUM all-reduce of the
 *      flat gradient block per window over RCCL; every rank then applies the identical Adagrad step. */
#define LSTM_HIP_UNIQUE_ID_BYTES 128
int lstm_hip_comm_unique_id(uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES]);
int lstm_hip_comm_init(lstm_hip_t *h, const uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank);
int lstm_hip_allreduce_grads(lstm_hip_t *h);

/* ---- the whole i-loop on the device (OV/lstm_eigen_opt/lstm.cc:186-318 without the host hops).
 *      set_text uploads the corpus once (rawread, R/lstm.cc:382-420); set_cursors the B read
 *      positions (opt:140-144); reset_window clears x/target to the all-zero columns (opt:122,125).
 *      train_windows runs `count` iterations of: event=text[pos]; pos++ (wrap to S); slide;
 *      forward; loss; backward; [all-reduce]; Adagrad.  losses (may be NULL) receives `count`
 *      per-window losses (what the reference adds to epoch_loss); for ranks of a communicator
 *      that is the local sum over this rank's streams divided by the GLOBAL batch.
 *      elapsed_ms (may be NULL) receives the HIP-event time of the loop on the handle's stream. */
int lstm_hip_set_text(lstm_hip_t *h, const uint8_t *text, size_t len);
int lstm_hip_set_cursors(lstm_hip_t *h, const uint64_t *pos);
int lstm_hip_get_cursors(lstm_hip_t *h, uint64_t *pos);
int lstm_hip_reset_window(lstm_hip_t *h);
int lstm_hip_get_window(lstm_hip_t *h, int32_t *xi, int32_t *ti);
int lstm_hip_train_windows(lstm_hip_t *h, int64_t count, double learning_rate, double *losses, float *elapsed_ms);
/* window stride variants (OV/lstm_eigen_class_batch/lstm_segment.cc:110,130,183-187): train_windows advances every
 * stream by `stride` bytes per iteration (default 1, the root file) and takes the carry h[0],c[0] from column
 * `carry_col` of the previous window (default 1; the segment variant uses stride = S/2, carry_col = S/2 - 1). */
int lstm_hip_set_stride(lstm_hip_t *h, int32_t stride, int32_t carry_col);
/* global batch the loss is divided by (defaults to B; set by the host when streams are sharded) */
int lstm_hip_set_global_batch(lstm_hip_t *h, int32_t global_B);
/* what lstm_hip_loss / train_windows report:
 *   ALL_STEPS_BITS  the sum over all S-1 steps of -log2 p(target) / B (R/lstm.cc:204-207; the default)
 *   LAST_STEP_NATS  step S-1 only, natural log (the CPU class of that variant: forward_loss,
 *                   OV/lstm_eigen_class_CUDA/lstm.h:200-221)
 *   LAST_STEP_BITS  step S-1 only, -log2, / B: exactly cuLSTM::calculate_loss, the boundary member lstm_hip_loss replaces
 *                   (OV/lstm_eigen_class_CUDA/cu_lstm.h:203-215 with kernel_elementwise_neglog, cu_kernels.cu:211-225)
 * The gradients do not change: that variant's backward still uses dy of every step (lstm.h:299-302, cu_lstm.h:216-300). */
#define LSTM_HIP_LOSS_ALL_STEPS_BITS 0
#define LSTM_HIP_LOSS_LAST_STEP_NATS 1
#define LSTM_HIP_LOSS_LAST_STEP_BITS 2
int lstm_hip_set_loss_mode(lstm_hip_t *h, int32_t mode);

/* ---- held-out evaluator and sampler on the device (OV/lstm_eigen_class_CUDA/lstm.cc:661-720,
 *      578-659; R/lstm.cc:293-356).  eval: bits/char of `text` from h = c = 0.  sample: `count`
 *      bytes from state (h0,c0) (N floats each, in/out) using the caller's uniform draws u[i]. */
int lstm_hip_eval_bits(lstm_hip_t *h, const uint8_t *text, size_t len, double *bits_per_char);
int lstm_hip_sample(lstm_hip_t *h, float *h0, float *c0, const double *u, int32_t count, uint8_t *out);

/* ---- measurement.  With profiling on, every kernel launch is bracketed by HIP events on the
 *      handle's stream and per-kernel totals accumulate. */
int lstm_hip_synchronize(lstm_hip_t *h);
/* [forward, backward][2 workgroups][S][16] shader-clock stamps of the last window (LSTM_HIP_DEBUG_STAMPS handles only;
 * slot meanings: persistent.hip, FSTAMP / BSTAMP) */
int lstm_hip_debug_stamps(lstm_hip_t *h, uint64_t *out, size_t count);
int lstm_hip_set_profiling(lstm_hip_t *h, int32_t on);
int lstm_hip_kernel_stat_count(lstm_hip_t *h);
int lstm_hip_kernel_stat(lstm_hip_t *h, int32_t idx, const char **name, int64_t *launches, double *total_ms);
int lstm_hip_reset_kernel_stats(lstm_hip_t *h);
/* device facts for the bench line: name (<= 63 chars), CU count, clock MHz */
int lstm_hip_device_info(int32_t device, char name[64], int32_t *cus, int32_t *clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* LSTM_HIP_H_ */
