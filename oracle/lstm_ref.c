/*
 * oracle/lstm_ref.c -- CPU restatement of the krocki/Eigen-LSTM training window.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing
 * in the product path (eigen-lstm_amd/) may include, link, dlopen or call it.
 *
 * What it restates (R/ = /root/reference, OV/ = R/optimized-obsfuscated_versions):
 *   forward  t=1..S-1           R/lstm.cc:173-209   batched form OV/lstm_eigen_opt/lstm.cc:216-251
 *   grad reset + BPTT           R/lstm.cc:214-257   batched form OV/lstm_eigen_opt/lstm.cc:256-304
 *   Adagrad                     R/lstm.cc:261-272   (eps is a double literal, R/lstm.cc:25,46-48)
 *   window slide / cursors      R/lstm.cc:155-170   batched form OV/lstm_eigen_opt/lstm.cc:140-144,190-213
 *   epoch state reset           OV/lstm_eigen_opt/lstm.cc:176-181
 *   param init + fill order     R/lstm.cc:113-129,364-380
 *   held-out evaluator test()   OV/lstm_eigen_class_CUDA/lstm.cc:661-720
 *   sampler                     R/lstm.cc:293-356
 *   finite-difference check     OV/lstm_eigen_class/lstm.h:131-170 (delta 1e-5, natural-log loss)
 *
 * Parity pinning: the reference itself cannot be built here (needs Eigen 3, absent), so this
 * restatement is pinned by (1) the reference's two known-answer checkpoints (tests/golden/
 * fixture_A_*, fixture_B_*: weights the reference saved + the bits/char its own log recorded),
 * (2) a finite-difference gradient check with the reference's thresholds, (3) an independent
 * torch-autograd model of the same recurrence.  See tests/test_oracle_*.py.
 *
 * Conventions kept from the reference (each one differs from a textbook LSTM):
 *   - gate row order in the 4N dimension is [i; o; f; u]          R/lstm.cc:77,185-192
 *   - the stored cell is tanh(i*u + f*c_prev) (already squashed)   R/lstm.cc:185-189
 *   - column 0 of the per-time buffers is carry-in only            R/lstm.cc:173,223
 *   - softmax has no max-subtraction                               R/lstm.cc:199-201
 *   - loss is -log2, divided by B in the batched variant           OV/lstm_eigen_opt/lstm.cc:246-249
 *   - weight gradients are sums (not means) over batch and time    OV/lstm_eigen_opt/lstm.cc:271,297-299
 *   - all matrices column-major (Eigen default)
 * Deviation (documented in SURVEY 8a note 5): x is zero-filled before first use, as
 *   OV/lstm_eigen_opt/lstm.cc:125 does; the root file leaves it uninitialised.
 *
 * One-hot columns are carried as indices: idx in [0,M) selects e_idx, idx < 0 is the all-zero
 * column (the state of x/target before the window has filled, OV/lstm_eigen_opt/lstm.cc:122,125).
 * W*x with x = e_k is then exactly column k of W (the dropped terms are exact zeros).
 *
 * Built as separate libraries (oracle/Makefile): liblstm_ref_f32.so (-DREAL=float -DSUF=_f32),
 * liblstm_ref_f64.so (-DREAL=double -DSUF=_f64 -DREF_DOUBLE) and liblstm_ref_f32_omp.so (f32 + -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#define SUF _f32
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#define EPS 1e-10 /* R/lstm.cc:25 -- double literal */

/* bf16 recurrence mode (BASELINE configs[4], "bf16 MFMA path"): the operands of the two recurrent products
 * U*h_prev (R/lstm.cc:176) and U^T*dg (R/lstm.cc:255) are rounded to bfloat16 (round-to-nearest-even, what
 * v_cvt_pk_bf16_f32 does); products are then exact in fp32 and the accumulation stays fp32.  Everything else
 * (W gather, biases, gates, output layer, every weight gradient, Adagrad on fp32 master weights) is unchanged. */
static int g_bf16_recurrence = 0;
void ref_set_bf16_recurrence(int on) { g_bf16_recurrence = on; }
/* Control for the trajectory tests (tests/trajectory_util.py): every contraction of the window runs over its index in
 * DESCENDING order.  The reference leaves the summation order of its products to Eigen / BLAS, so this is as
 * correct an implementation of OV/lstm_eigen_opt/lstm.cc:186-318 as the ascending one; how far the two drift apart in a
 * free-running training loop is the yardstick for a GPU implementation whose products sum in yet another order. */
static int g_desc_sums = 0;
void ref_set_descending_sums(int on) { g_desc_sums = on; }
#define KI(k, n) (g_desc_sums ? (n) - 1 - (k) : (k))
/* ... and of the four time-batched products as well (the complete "bf16 MFMA path" of configs[4]): y = Why*h, dWhy = dy*h^T,
 * Why^T*dy and dU = dg*h_prev^T take bf16-rounded operands (the SAME rounded h, dg as the recurrence; dy and Why rounded
 * once), accumulate in fp32.  Biases, dW/db/dby (sums, no products), the elementwise math and Adagrad stay fp32. */
static int g_bf16_products = 0;
void ref_set_bf16_products(int on) { g_bf16_products = on; }
static inline float bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN stays NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

/* ------------------------------------------------------------------------------------------
 * RNG.  The reference seeds a fresh mt19937 from std::random_device on every randn() call
 * (R/lstm.cc:370-372), so it has no reproducible stream to match.  The build defines one:
 * MT19937 (Matsumoto & Nishimura 1998), 53-bit uniforms (genrand_res53), Marsaglia polar
 * normals with the spare value cached.  The product's host code implements the same spec.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t mt[624];
    int idx;
    int have_spare;
    double spare;
} ref_rng;

void ref_rng_seed(ref_rng *r, uint32_t seed) {
    r->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
    r->have_spare = 0;
    r->spare = 0.0;
}
uint32_t ref_rng_u32(ref_rng *r) {
    if (r->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (r->mt[i] & 0x80000000u) | (r->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = r->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            r->mt[i] = v;
        }
        r->idx = 0;
    }
    uint32_t y = r->mt[r->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
double ref_rng_uniform(ref_rng *r) { /* [0,1) with 53 bits */
    uint32_t a = ref_rng_u32(r) >> 5, b = ref_rng_u32(r) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}
double ref_rng_normal(ref_rng *r) { /* Marsaglia polar, N(0,1) */
    if (r->have_spare) {
        r->have_spare = 0;
        return r->spare;
    }
    double u, v, s;
    do {
        u = 2.0 * ref_rng_uniform(r) - 1.0;
        v = 2.0 * ref_rng_uniform(r) - 1.0;
        s = u * u + v * v;
    } while (s >= 1.0 || s == 0.0);
    double m = sqrt(-2.0 * log(s) / s);
    r->spare = v * m;
    r->have_spare = 1;
    return u * m;
}
size_t ref_rng_sizeof(void) { return sizeof(ref_rng); }

/* randn(m, mean, stddev): row-outer, column-inner fill of a column-major matrix, double draw
 * narrowed to the element type.  R/lstm.cc:364-380 */
void FN(ref_randn)(ref_rng *r, REAL *m, int rows, int cols, double mean, double stddev) {
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) m[(size_t)j * rows + i] = (REAL)(mean + stddev * ref_rng_normal(r));
}

/* ------------------------------------------------------------------------------------------
 * Parameter block: one flat buffer [W (4N x M) | U (4N x N) | b (4N) | Why (M x N) | by (M)],
 * each matrix column-major.  Same order as Parameters{W,U,b,Why,by}
 * (OV/lstm_eigen_class_CUDA/lstm.h:43-112).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    REAL *W, *U, *b, *Why, *by;
} FN(pview);

static size_t param_count(int N, int M) {
    return (size_t)4 * N * M + (size_t)4 * N * N + (size_t)4 * N + (size_t)M * N + (size_t)M;
}
size_t FN(ref_param_count)(int N, int M) { return param_count(N, M); }

static FN(pview) FN(view)(REAL *P, int N, int M) {
    FN(pview) v;
    v.W = P;
    v.U = v.W + (size_t)4 * N * M;
    v.b = v.U + (size_t)4 * N * N;
    v.Why = v.b + (size_t)4 * N;
    v.by = v.Why + (size_t)M * N;
    return v;
}

/* R/lstm.cc:113-119: W, U, Why ~ N(0, 0.01) in that order; b = by = 0 */
void FN(ref_init_params)(ref_rng *r, REAL *P, int N, int M) {
    FN(pview) p = FN(view)(P, N, M);
    FN(ref_randn)(r, p.W, 4 * N, M, 0.0, 0.01);
    FN(ref_randn)(r, p.U, 4 * N, N, 0.0, 0.01);
    FN(ref_randn)(r, p.Why, M, N, 0.0, 0.01);
    memset(p.b, 0, sizeof(REAL) * 4 * N);
    memset(p.by, 0, sizeof(REAL) * M);
}

/* scalar helpers, R/lstm.cc:30-48 */
#if defined(REF_DOUBLE)
#define EXP exp
#define TANH tanh
#define LOG2 log2
#define LOGN log
#define SQRT sqrt
#else
#define EXP expf
#define TANH tanhf
#define LOG2 log2f
#define LOGN logf
#define SQRT sqrtf
#endif
static inline REAL logistic(REAL x) { return (REAL)1 / ((REAL)1 + EXP(-x)); }
static inline REAL tanh_prime(REAL x) { return (REAL)1 - x * x; }
static inline REAL logistic_prime(REAL x) { return x * ((REAL)1 - x); }
static inline REAL sqrt_eps(REAL x) { return SQRT((REAL)((double)x + EPS)); } /* double add, then narrowed */

/* ------------------------------------------------------------------------------------------
 * Window state.  Per-time buffers are stored [t][col-major rows x B], t = 0..S-1.
 *   h, c : N x B      g : 4N x B (post-activation gates)      probs : M x B
 *   xi, ti : S x B int32 indices (xi[t*B+b]), < 0 = zero column
 * ------------------------------------------------------------------------------------------ */

/* forward over t = 1..S-1.  OV/lstm_eigen_opt/lstm.cc:216-251 (R/lstm.cc:173-209 at B = 1).
 * Returns the reported loss (sum_t [sum_b -log2 p_target] / B) in *loss_bits and the objective
 * whose gradient backward() computes (sum_t sum_b -ln p_target) in *loss_nats. */
void FN(ref_forward)(int N, int M, int S, int B, const REAL *P, const int32_t *xi, const int32_t *ti, REAL *h,
                     REAL *c, REAL *g, REAL *probs, double *loss_bits, double *loss_nats) {
    FN(pview) p = FN(view)((REAL *)P, N, M);
    const int G = 4 * N;
    double lb = 0.0, ln_ = 0.0;
    REAL *surpr = (REAL *)malloc(sizeof(REAL) * B);
    double *surpn = (double *)malloc(sizeof(double) * B);
    REAL *Ub = NULL; /* bf16 mode: U rounded once */
    if (g_bf16_recurrence) {
        Ub = (REAL *)malloc(sizeof(REAL) * (size_t)G * N);
        for (size_t i = 0; i < (size_t)G * N; i++) Ub[i] = (REAL)bf16_rne((float)p.U[i]);
    }
    const REAL *Urec = Ub ? Ub : p.U;
    REAL *Whyb = NULL; /* bf16 products mode: Why rounded once */
    if (g_bf16_products) {
        Whyb = (REAL *)malloc(sizeof(REAL) * (size_t)M * N);
        for (size_t i = 0; i < (size_t)M * N; i++) Whyb[i] = (REAL)bf16_rne((float)p.Why[i]);
    }
    const REAL *Whyf = Whyb ? Whyb : p.Why;
    for (int t = 1; t < S; t++) {
        REAL *gt = g + (size_t)t * G * B, *ht = h + (size_t)t * N * B, *ct = c + (size_t)t * N * B;
        const REAL *hp = h + (size_t)(t - 1) * N * B, *cp = c + (size_t)(t - 1) * N * B;
        REAL *pt = probs + (size_t)t * M * B;
        /* OpenMP pragmas (active only in the -fopenmp build used as the all-cores CPU baseline)
         * split work over independent columns/rows; every output element keeps its serial
         * summation order, so all builds give identical results. */
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; b++) {
            REAL *gc = gt + (size_t)b * G;
            /* g = W*x + U*h_prev + b   (opt:219) */
            /* loops run k-outer / r-inner for cache locality; each gc[r] still accumulates its
             * products in ascending k, so the rounding sequence is that of a plain dot product */
            int xk = xi[t * B + b];
            for (int r = 0; r < G; r++) gc[r] = 0;
            for (int k0 = 0; k0 < N; k0++) {
                const int k = KI(k0, N);
                const REAL hk = Ub ? (REAL)bf16_rne((float)hp[(size_t)b * N + k]) : hp[(size_t)b * N + k];
                const REAL *Uk = Urec + (size_t)k * G;
                for (int r = 0; r < G; r++) gc[r] += Uk[r] * hk;
            }
            for (int r = 0; r < G; r++) {
                REAL acc = xk >= 0 ? p.W[(size_t)xk * G + r] : (REAL)0;
                gc[r] = (acc + gc[r]) + p.b[r];
            }
            /* sigmoid on i,o,f; tanh on u   (opt:222-224) */
            for (int r = 0; r < 3 * N; r++) gc[r] = logistic(gc[r]);
            for (int r = 3 * N; r < G; r++) gc[r] = TANH(gc[r]);
            /* c = tanh(i*u + f*c_prev); h = o*c   (opt:226-233) */
            for (int j = 0; j < N; j++) {
                REAL z = gc[j] * gc[3 * N + j] + gc[2 * N + j] * cp[(size_t)b * N + j];
                ct[(size_t)b * N + j] = TANH(z);
                ht[(size_t)b * N + j] = gc[N + j] * ct[(size_t)b * N + j];
            }
            /* y = Why*h + by; probs = exp(y)/sum   (opt:236-242), no max shift */
            REAL *pc = pt + (size_t)b * M;
            REAL sum = 0;
            for (int m = 0; m < M; m++) pc[m] = 0;
            for (int k0 = 0; k0 < N; k0++) {
                const int k = KI(k0, N);
                const REAL hk = Whyb ? (REAL)bf16_rne((float)ht[(size_t)b * N + k]) : ht[(size_t)b * N + k];
                const REAL *Wk = Whyf + (size_t)k * M;
                for (int m = 0; m < M; m++) pc[m] += Wk[m] * hk;
            }
            for (int m = 0; m < M; m++) {
                pc[m] = EXP(pc[m] + p.by[m]);
                sum += pc[m];
            }
            for (int m = 0; m < M; m++) pc[m] = pc[m] / sum;
            /* surprisal = -log2(p) .* target   (opt:246) */
            int tk = ti[t * B + b];
            surpr[b] = tk >= 0 ? -LOG2(pc[tk]) : (REAL)0;
            surpn[b] = tk >= 0 ? -(double)LOGN(pc[tk]) : 0.0;
        }
        REAL surpr_sum = 0; /* surprisals.sum() is a float reduction in the reference */
        for (int b = 0; b < B; b++) {
            surpr_sum += surpr[b];
            ln_ += surpn[b];
        }
        lb += (double)(surpr_sum / (REAL)B); /* opt:249 */
    }
    free(surpr); free(surpn); free(Ub); free(Whyb);
    if (loss_bits) *loss_bits = lb;
    if (loss_nats) *loss_nats = ln_;
}

/* zero grads + BPTT t = S-1..1.  OV/lstm_eigen_opt/lstm.cc:256-304 (R/lstm.cc:214-257).
 * dP has the layout of P.  scratch-free: allocates its own temporaries. */
void FN(ref_backward)(int N, int M, int S, int B, const REAL *P, const int32_t *xi, const int32_t *ti,
                      const REAL *h, const REAL *c, const REAL *g, const REAL *probs, REAL *dP) {
    FN(pview) p = FN(view)((REAL *)P, N, M);
    FN(pview) d = FN(view)(dP, N, M);
    const int G = 4 * N;
    memset(dP, 0, sizeof(REAL) * param_count(N, M));
    REAL *dy = (REAL *)malloc(sizeof(REAL) * M * B);
    REAL *dh = (REAL *)malloc(sizeof(REAL) * N * B);
    REAL *dc = (REAL *)malloc(sizeof(REAL) * N * B);
    REAL *dg = (REAL *)malloc(sizeof(REAL) * G * B);
    REAL *dhnext = (REAL *)calloc((size_t)N * B, sizeof(REAL));
    REAL *dcnext = (REAL *)calloc((size_t)N * B, sizeof(REAL));
    REAL *utmp = (REAL *)malloc(sizeof(REAL) * (size_t)N * G);
    REAL *Ub = NULL, *dgb = NULL; /* bf16 mode: rounded operands of U^T*dg */
    if (g_bf16_recurrence) {
        Ub = (REAL *)malloc(sizeof(REAL) * (size_t)G * N);
        dgb = (REAL *)malloc(sizeof(REAL) * (size_t)G * B);
        for (size_t i = 0; i < (size_t)G * N; i++) Ub[i] = (REAL)bf16_rne((float)p.U[i]);
    }
    const REAL *Urec = Ub ? Ub : p.U;
    REAL *Whyb = NULL, *dyb = NULL, *hb = NULL, *hpb = NULL; /* bf16 products mode: rounded operand copies */
    if (g_bf16_products) {
        Whyb = (REAL *)malloc(sizeof(REAL) * (size_t)M * N);
        for (size_t i = 0; i < (size_t)M * N; i++) Whyb[i] = (REAL)bf16_rne((float)p.Why[i]);
        dyb = (REAL *)malloc(sizeof(REAL) * (size_t)M * B);
        hb = (REAL *)malloc(sizeof(REAL) * (size_t)N * B);
        hpb = (REAL *)malloc(sizeof(REAL) * (size_t)N * B);
        if (!dgb) dgb = (REAL *)malloc(sizeof(REAL) * (size_t)G * B);
    }
    const REAL *Whyf = Whyb ? Whyb : p.Why;
    for (int t = S - 1; t > 0; t--) {
        const REAL *gt = g + (size_t)t * G * B, *ht = h + (size_t)t * N * B, *ct = c + (size_t)t * N * B;
        const REAL *hp = h + (size_t)(t - 1) * N * B, *cp = c + (size_t)(t - 1) * N * B;
        const REAL *pt = probs + (size_t)t * M * B;
        /* dy = probs - target   (opt:270) */
        for (int b = 0; b < B; b++) {
            int tk = ti[t * B + b];
            for (int m = 0; m < M; m++) dy[(size_t)b * M + m] = pt[(size_t)b * M + m] - (m == tk ? (REAL)1 : (REAL)0);
        }
        if (dyb) {
            for (size_t i = 0; i < (size_t)M * B; i++) dyb[i] = (REAL)bf16_rne((float)dy[i]);
            for (size_t i = 0; i < (size_t)N * B; i++) hb[i] = (REAL)bf16_rne((float)ht[i]);
            for (size_t i = 0; i < (size_t)N * B; i++) hpb[i] = (REAL)bf16_rne((float)hp[i]);
        }
        const REAL *dyp = dyb ? dyb : dy, *htp = hb ? hb : ht, *hpp = hpb ? hpb : hp; /* product operands */
        /* dWhy += dy * h^T ; dby += rowsum(dy)   (opt:271-272) */
#pragma omp parallel for schedule(static)
        for (int k = 0; k < N; k++) {
            REAL tmp[256];
            for (int m = 0; m < M; m++) tmp[m] = 0;
            for (int b0 = 0; b0 < B; b0++) {
                const int b = KI(b0, B);
                const REAL hk = htp[(size_t)b * N + k];
                for (int m = 0; m < M; m++) tmp[m] += dyp[(size_t)b * M + m] * hk;
            }
            for (int m = 0; m < M; m++) d.Why[(size_t)k * M + m] += tmp[m];
        }
        for (int m = 0; m < M; m++) {
            REAL acc = 0;
            for (int b0 = 0; b0 < B; b0++) acc += dy[(size_t)KI(b0, B) * M + m];
            d.by[m] += acc;
        }
        /* dh = Why^T * dy + dhnext   (opt:273) */
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; b++)
            for (int k = 0; k < N; k++) {
                REAL acc = 0;
                for (int m0 = 0; m0 < M; m0++) {
                    const int m = KI(m0, M);
                    acc += Whyf[(size_t)k * M + m] * dyp[(size_t)b * M + m];
                }
                dh[(size_t)b * N + k] = acc + dhnext[(size_t)b * N + k];
            }
        for (int b = 0; b < B; b++) {
            const REAL *gc = gt + (size_t)b * G;
            REAL *dgc = dg + (size_t)b * G;
            for (int j = 0; j < N; j++) {
                size_t o = (size_t)b * N + j;
                /* dc = (dh .* o + dcnext) .* (1 - c^2)   (opt:278-280) */
                REAL dcv = dh[o] * gc[N + j] + dcnext[o];
                dcv = dcv * tanh_prime(ct[o]);
                dc[o] = dcv;
                /* gates (opt:283-286) then through the nonlinearities (opt:289-294) */
                dgc[N + j] = (dh[o] * ct[o]) * logistic_prime(gc[N + j]);           /* do */
                dgc[j] = (dcv * gc[3 * N + j]) * logistic_prime(gc[j]);             /* di */
                dgc[2 * N + j] = (dcv * cp[o]) * logistic_prime(gc[2 * N + j]);     /* df */
                dgc[3 * N + j] = (dcv * gc[j]) * tanh_prime(gc[3 * N + j]);         /* du */
            }
        }
        /* the rounded dg serves the dU product (products mode) and the recurrent product (recurrence mode) alike */
        if (dgb)
            for (size_t i = 0; i < (size_t)G * B; i++) dgb[i] = (REAL)bf16_rne((float)dg[i]);
        const REAL *dgprod = g_bf16_products ? dgb : dg;
        /* dU += dg * h_prev^T ; dW += dg * x^T ; db += rowsum(dg)   (opt:297-299) */
#pragma omp parallel for schedule(static)
        for (int k = 0; k < N; k++) {
            REAL *tmp = utmp + (size_t)k * G; /* per-k scratch row (thread-private under OpenMP) */
            for (int r = 0; r < G; r++) tmp[r] = 0;
            for (int b0 = 0; b0 < B; b0++) {
                const int b = KI(b0, B);
                const REAL hk = hpp[(size_t)b * N + k];
                const REAL *dgc2 = dgprod + (size_t)b * G;
                for (int r = 0; r < G; r++) tmp[r] += dgc2[r] * hk;
            }
            for (int r = 0; r < G; r++) d.U[(size_t)k * G + r] += tmp[r];
        }
        for (int b = 0; b < B; b++) {
            int xk = xi[t * B + b];
            if (xk >= 0)
                for (int r = 0; r < G; r++) d.W[(size_t)xk * G + r] += dg[(size_t)b * G + r];
        }
        for (int r = 0; r < G; r++) {
            REAL acc = 0;
            for (int b0 = 0; b0 < B; b0++) acc += dg[(size_t)KI(b0, B) * G + r];
            d.b[r] += acc;
        }
        /* dhnext = U^T * dg ; dcnext = dc .* f   (opt:302-303) */
        const REAL *dgrec = g_bf16_recurrence ? dgb : dg;
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; b++)
            for (int k = 0; k < N; k++) {
                REAL acc = 0;
                for (int r0 = 0; r0 < G; r0++) {
                    const int r = KI(r0, G);
                    acc += Urec[(size_t)k * G + r] * dgrec[(size_t)b * G + r];
                }
                dhnext[(size_t)b * N + k] = acc;
            }
        for (int b = 0; b < B; b++)
            for (int j = 0; j < N; j++)
                dcnext[(size_t)b * N + j] = dc[(size_t)b * N + j] * gt[(size_t)b * G + 2 * N + j];
    }
    free(dy); free(dh); free(dc); free(dg); free(dhnext); free(dcnext); free(utmp); free(Ub); free(dgb);
    free(Whyb); free(dyb); free(hb); free(hpb);
}

/* m += d.*d ; p -= lr * d ./ sqrt(m + eps)   R/lstm.cc:261-272, over the whole flat block */
void FN(ref_adagrad)(size_t n, REAL *P, const REAL *dP, REAL *mem, REAL lr) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        mem[i] += dP[i] * dP[i];
        P[i] -= lr * (dP[i] / sqrt_eps(mem[i]));
    }
}

/* Central-difference gradient of the natural-log objective at `count` sampled flat indices.
 * OV/lstm_eigen_class/lstm.h:131-170 (delta = 1e-5).  Restores P. */
void FN(ref_numgrad)(int N, int M, int S, int B, REAL *P, const int32_t *xi, const int32_t *ti, const REAL *h0,
                     const REAL *c0, const int64_t *which, int count, double delta, double *out) {
    size_t nh = (size_t)N * B * S, ng = (size_t)4 * N * B * S, np = (size_t)M * B * S;
    REAL *h = (REAL *)calloc(nh, sizeof(REAL)), *c = (REAL *)calloc(nh, sizeof(REAL));
    REAL *g = (REAL *)calloc(ng, sizeof(REAL)), *pr = (REAL *)calloc(np, sizeof(REAL));
    memcpy(h, h0, sizeof(REAL) * N * B);
    memcpy(c, c0, sizeof(REAL) * N * B);
    for (int i = 0; i < count; i++) {
        REAL keep = P[which[i]];
        double lp, lm;
        P[which[i]] = (REAL)((double)keep + delta);
        FN(ref_forward)(N, M, S, B, P, xi, ti, h, c, g, pr, NULL, &lp);
        P[which[i]] = (REAL)((double)keep - delta);
        FN(ref_forward)(N, M, S, B, P, xi, ti, h, c, g, pr, NULL, &lm);
        P[which[i]] = keep;
        out[i] = (lp - lm) / (2.0 * delta);
    }
    free(h); free(c); free(g); free(pr);
}

/* Held-out evaluator: bits/char over a byte string from h = c = 0 (reset_std = 0),
 * OV/lstm_eigen_class_CUDA/lstm.cc:661-720.  The probability sum is a double there (:712). */
double FN(ref_eval_bits)(int N, int M, const REAL *P, const uint8_t *text, size_t len) {
    FN(pview) p = FN(view)((REAL *)P, N, M);
    const int G = 4 * N;
    REAL *hh = (REAL *)calloc(N, sizeof(REAL)), *cc = (REAL *)calloc(N, sizeof(REAL));
    REAL *gg = (REAL *)malloc(sizeof(REAL) * G), *pp = (REAL *)malloc(sizeof(REAL) * M);
    double err = 0.0;
    for (size_t ii = 0; ii + 1 < len; ii++) {
        int ex = text[ii], et = text[ii + 1];
        for (int r = 0; r < G; r++) {
            REAL uh = 0;
            for (int k = 0; k < N; k++) uh += p.U[(size_t)k * G + r] * hh[k];
            gg[r] = (p.W[(size_t)ex * G + r] + uh) + p.b[r];
        }
        for (int r = 0; r < 3 * N; r++) gg[r] = logistic(gg[r]);
        for (int r = 3 * N; r < G; r++) gg[r] = TANH(gg[r]);
        for (int j = 0; j < N; j++) {
            cc[j] = TANH(gg[j] * gg[3 * N + j] + gg[2 * N + j] * cc[j]);
            hh[j] = gg[N + j] * cc[j];
        }
        double sum = 0.0;
        for (int m = 0; m < M; m++) {
            REAL y = 0;
            for (int k = 0; k < N; k++) y += p.Why[(size_t)k * M + m] * hh[k];
            pp[m] = EXP(y + p.by[m]);
            sum += pp[m];
        }
        err += -log2((double)(REAL)(pp[et] / sum));
    }
    free(hh); free(cc); free(gg); free(pp);
    return err / (double)(len - 1);
}

/* Sampler, R/lstm.cc:293-356: softmax(Why*h+by) -> cdf -> first index with r < cdf -> feed back.
 * h, c are in/out (N); u[i] are the caller's uniform draws in [0,1). */
void FN(ref_sample)(int N, int M, const REAL *P, REAL *hh, REAL *cc, const double *u, int count, uint8_t *out) {
    FN(pview) p = FN(view)((REAL *)P, N, M);
    const int G = 4 * N;
    REAL *gg = (REAL *)malloc(sizeof(REAL) * G), *pp = (REAL *)malloc(sizeof(REAL) * M);
    for (int i = 0; i < count; i++) {
        REAL sum = 0;
        for (int m = 0; m < M; m++) {
            REAL y = 0;
            for (int k = 0; k < N; k++) y += p.Why[(size_t)k * M + m] * hh[k];
            pp[m] = EXP(y + p.by[m]);
            sum += pp[m];
        }
        REAL cdf = 0, r = (REAL)u[i];
        int index = 0;
        for (int m = 0; m < M; m++) {
            cdf += pp[m] / sum;
            if (r < cdf) { index = m; break; }
        }
        out[i] = (uint8_t)index;
        for (int q = 0; q < G; q++) {
            REAL uh = 0;
            for (int k = 0; k < N; k++) uh += p.U[(size_t)k * G + q] * hh[k];
            gg[q] = (p.W[(size_t)index * G + q] + uh) + p.b[q];
        }
        for (int q = 0; q < 3 * N; q++) gg[q] = logistic(gg[q]);
        for (int q = 3 * N; q < G; q++) gg[q] = TANH(gg[q]);
        for (int j = 0; j < N; j++) {
            cc[j] = TANH(gg[j] * gg[3 * N + j] + gg[2 * N + j] * cc[j]);
            hh[j] = gg[N + j] * cc[j];
        }
    }
    free(gg); free(pp);
}

/* ------------------------------------------------------------------------------------------
 * Trainer: the outer loop of OV/lstm_eigen_opt/lstm.cc:172-332 with runtime N,S,B and a seed.
 *   cursors   pos[b] = S + (b*(len-S))/B  -- deterministic stand-in for rand()%(len-S)+S (opt:140-144);
 *             `stream0`/`streams_total` let a rank own a slice of a larger batch (multi-GPU tests)
 *   per window (opt:190-213): event = text[pos]; pos++ (wrap to S); shift x,target,h,c left by one
 *             column; target[S-1] = onehot(event); x[S-1] = target[S-2]
 *   epoch start (opt:176-181): h[t], c[t] ~ N(0, 0.1) for every t, drawn h[0],c[0],h[1],c[1],...
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int N, M, S, B;
    REAL lr;
    const uint8_t *text;
    size_t len;
    size_t *pos;
    int32_t *xi, *ti; /* S x B */
    REAL *P, *dP, *mem;
    REAL *h, *c, *g, *probs;
    ref_rng rng;
    size_t np;
} FN(trainer);

FN(trainer) *FN(ref_trainer_create)(const uint8_t *text, size_t len, int N, int M, int S, int B, double lr,
                                    uint32_t seed, int stream0, int streams_total) {
    FN(trainer) *T = (FN(trainer) *)calloc(1, sizeof(FN(trainer)));
    T->N = N; T->M = M; T->S = S; T->B = B; T->lr = (REAL)lr; T->text = text; T->len = len;
    T->np = param_count(N, M);
    T->P = (REAL *)calloc(T->np, sizeof(REAL));
    T->dP = (REAL *)calloc(T->np, sizeof(REAL));
    T->mem = (REAL *)calloc(T->np, sizeof(REAL));
    T->h = (REAL *)calloc((size_t)N * B * S, sizeof(REAL));
    T->c = (REAL *)calloc((size_t)N * B * S, sizeof(REAL));
    T->g = (REAL *)calloc((size_t)4 * N * B * S, sizeof(REAL));
    T->probs = (REAL *)calloc((size_t)M * B * S, sizeof(REAL));
    T->xi = (int32_t *)malloc(sizeof(int32_t) * S * B);
    T->ti = (int32_t *)malloc(sizeof(int32_t) * S * B);
    for (int i = 0; i < S * B; i++) T->xi[i] = T->ti[i] = -1;
    T->pos = (size_t *)malloc(sizeof(size_t) * B);
    for (int b = 0; b < B; b++) T->pos[b] = (size_t)S + ((size_t)(stream0 + b) * (len - S)) / (size_t)streams_total;
    ref_rng_seed(&T->rng, seed);
    FN(ref_init_params)(&T->rng, T->P, N, M);
    return T;
}
void FN(ref_trainer_destroy)(FN(trainer) *T) {
    free(T->P); free(T->dP); free(T->mem); free(T->h); free(T->c); free(T->g); free(T->probs);
    free(T->xi); free(T->ti); free(T->pos); free(T);
}
/* opt:176-181.  With B streams of a larger batch the caller passes full-width draws through
 * ref_trainer_set_state instead (see tests). */
void FN(ref_trainer_epoch_reset)(FN(trainer) *T) {
    for (int t = 0; t < T->S; t++) {
        FN(ref_randn)(&T->rng, T->h + (size_t)t * T->N * T->B, T->N, T->B, 0.0, 0.1);
        FN(ref_randn)(&T->rng, T->c + (size_t)t * T->N * T->B, T->N, T->B, 0.0, 0.1);
    }
}
/* opt:190-213 */
void FN(ref_trainer_slide)(FN(trainer) *T) {
    const int S = T->S, B = T->B, N = T->N;
    for (int b = 0; b < B; b++) {
        int event = T->text[T->pos[b]];
        T->pos[b]++;
        if (T->pos[b] >= T->len) T->pos[b] = (size_t)S;
        for (int s = 1; s < S; s++) {
            T->xi[(s - 1) * B + b] = T->xi[s * B + b];
            T->ti[(s - 1) * B + b] = T->ti[s * B + b];
            memcpy(T->h + ((size_t)(s - 1) * B + b) * N, T->h + ((size_t)s * B + b) * N, sizeof(REAL) * N);
            memcpy(T->c + ((size_t)(s - 1) * B + b) * N, T->c + ((size_t)s * B + b) * N, sizeof(REAL) * N);
        }
        T->ti[(S - 1) * B + b] = event;
        T->xi[(S - 1) * B + b] = T->ti[(S - 2) * B + b];
    }
}
/* one iteration of the i-loop: slide, forward, backward, Adagrad.  Returns the window's loss
 * (what the reference adds to epoch_loss, opt:253). */
double FN(ref_trainer_window)(FN(trainer) *T, int do_update) {
    double loss = 0.0;
    FN(ref_trainer_slide)(T);
    FN(ref_forward)(T->N, T->M, T->S, T->B, T->P, T->xi, T->ti, T->h, T->c, T->g, T->probs, &loss, NULL);
    FN(ref_backward)(T->N, T->M, T->S, T->B, T->P, T->xi, T->ti, T->h, T->c, T->g, T->probs, T->dP);
    if (do_update) FN(ref_adagrad)(T->np, T->P, T->dP, T->mem, T->lr);
    return loss;
}
REAL *FN(ref_trainer_params)(FN(trainer) *T) { return T->P; }
REAL *FN(ref_trainer_grads)(FN(trainer) *T) { return T->dP; }
REAL *FN(ref_trainer_mem)(FN(trainer) *T) { return T->mem; }
REAL *FN(ref_trainer_h)(FN(trainer) *T) { return T->h; }
REAL *FN(ref_trainer_c)(FN(trainer) *T) { return T->c; }
REAL *FN(ref_trainer_g)(FN(trainer) *T) { return T->g; }
REAL *FN(ref_trainer_probs)(FN(trainer) *T) { return T->probs; }
int32_t *FN(ref_trainer_xi)(FN(trainer) *T) { return T->xi; }
int32_t *FN(ref_trainer_ti)(FN(trainer) *T) { return T->ti; }
ref_rng *FN(ref_trainer_rng)(FN(trainer) *T) { return &T->rng; }
