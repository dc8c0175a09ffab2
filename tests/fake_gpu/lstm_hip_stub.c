/* lstm_hip_stub.c -- TEST INFRASTRUCTURE: a GPU-less stand-in for liblstm_hip.so that exports the entry points the C++
 * host program (eigen-lstm_amd/host/lstm_main.cc) binds, so the CPU suite can drive that program's --gpus G path:
 * fork per rank before any library call, relay of the unique id and of the epoch loss, termination of the job when a
 * rank dies.  Every call appends "pid rank name ..." to the file named by LSTM_STUB_LOG.  No computation happens: a window
 * "costs" LSTM_STUB_LOSS bits per step and stream (default 2), reported the way the library reports it (local surprisal
 * sum / GLOBAL batch).  LSTM_STUB_FAIL_RANK=r makes rank r fail its second lstm_hip_train_windows call, while the other
 * ranks hang in theirs as ranks blocked in an all-reduce would.
 * Built by tests/test_host_multirank_cpu.py into a temporary directory as liblstm_hip.so; never shipped. */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/lstm_hip.h"

struct lstm_hip_ctx {
    lstm_hip_config cfg;
    int nranks, rank, global_B, calls;
    uint64_t *pos;
    uint64_t text_len;
    int stride;
};

static char g_err[256] = "";
static void logf_(const struct lstm_hip_ctx *h, const char *fmt, ...) {
    const char *path = getenv("LSTM_STUB_LOG");
    if (!path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%d %d ", (int)getpid(), h ? h->rank : -1);
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fputc('\n', f);
    fclose(f);
}

const char *lstm_hip_last_error(void) { return g_err; }
size_t lstm_hip_param_count(int32_t N, int32_t M) { return (size_t)4 * N * M + (size_t)4 * N * N + (size_t)4 * N + (size_t)M * N + M; }
int lstm_hip_create(const lstm_hip_config *cfg, lstm_hip_t **out) {
    struct lstm_hip_ctx *h = (struct lstm_hip_ctx *)calloc(1, sizeof(*h));
    h->cfg = *cfg;
    h->nranks = 1;
    h->rank = cfg->device; /* the host passes its rank as the device ordinal */
    h->global_B = cfg->B;
    h->pos = (uint64_t *)calloc((size_t)cfg->B, sizeof(uint64_t));
    h->stride = 1;
    *out = h;
    logf_(h, "create device=%d N=%d S=%d B=%d", cfg->device, cfg->N, cfg->S, cfg->B);
    return 0;
}
int lstm_hip_destroy(lstm_hip_t *h) {
    logf_(h, "destroy");
    if (h) free(h->pos), free(h);
    return 0;
}
int lstm_hip_set_params(lstm_hip_t *h, int which, const float *p) { (void)p; logf_(h, "set_params %d", which); return 0; }
int lstm_hip_get_params(lstm_hip_t *h, int which, float *p) {
    memset(p, 0, sizeof(float) * lstm_hip_param_count(h->cfg.N, h->cfg.M));
    logf_(h, "get_params %d", which);
    return 0;
}
int lstm_hip_set_state(lstm_hip_t *h, int32_t t, const float *a, const float *b) { (void)h, (void)t, (void)a, (void)b; return 0; }
int lstm_hip_set_text(lstm_hip_t *h, const uint8_t *t, size_t len) { (void)t; h->text_len = len; logf_(h, "set_text %zu", len); return 0; }
int lstm_hip_set_cursors(lstm_hip_t *h, const uint64_t *pos) {
    memcpy(h->pos, pos, sizeof(uint64_t) * (size_t)h->cfg.B);
    logf_(h, "set_cursors first=%llu", (unsigned long long)pos[0]);
    return 0;
}
int lstm_hip_get_cursors(lstm_hip_t *h, uint64_t *pos) { memcpy(pos, h->pos, sizeof(uint64_t) * (size_t)h->cfg.B); return 0; }
int lstm_hip_reset_window(lstm_hip_t *h) { (void)h; return 0; }
int lstm_hip_set_stride(lstm_hip_t *h, int32_t s, int32_t c) { (void)c; h->stride = s; return 0; }
int lstm_hip_set_global_batch(lstm_hip_t *h, int32_t gb) { h->global_B = gb; logf_(h, "set_global_batch %d", gb); return 0; }
int lstm_hip_set_loss_mode(lstm_hip_t *h, int32_t m) { (void)h, (void)m; return 0; }
int lstm_hip_comm_unique_id(uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES]) {
    for (int i = 0; i < LSTM_HIP_UNIQUE_ID_BYTES; i++) id[i] = (uint8_t)(i * 7 + 3);
    logf_(NULL, "comm_unique_id");
    return 0;
}
int lstm_hip_comm_init(lstm_hip_t *h, const uint8_t id[LSTM_HIP_UNIQUE_ID_BYTES], int32_t nranks, int32_t rank) {
    int ok = 1;
    for (int i = 0; i < LSTM_HIP_UNIQUE_ID_BYTES; i++) ok = ok && id[i] == (uint8_t)(i * 7 + 3);
    h->nranks = nranks;
    h->rank = rank;
    logf_(h, "comm_init nranks=%d rank=%d id_ok=%d", nranks, rank, ok);
    return ok ? 0 : LSTM_HIP_ERCCL;
}
int lstm_hip_train_windows(lstm_hip_t *h, int64_t count, double lr, double *losses, float *ms) {
    (void)lr;
    h->calls++;
    const char *fr = getenv("LSTM_STUB_FAIL_RANK");
    if (fr && h->calls >= 2) {
        if (atoi(fr) == h->rank) {
            snprintf(g_err, sizeof(g_err), "stub: injected failure on rank %d", h->rank);
            logf_(h, "train_windows FAIL");
            return LSTM_HIP_ESTATE;
        }
        logf_(h, "train_windows HANG"); /* a rank whose peer died blocks inside the all-reduce */
        sleep(120);
        return LSTM_HIP_ERCCL;
    }
    const double per = getenv("LSTM_STUB_LOSS") ? atof(getenv("LSTM_STUB_LOSS")) : 2.0;
    /* what the library reports per window: sum over the S-1 steps of (local surprisal sum / GLOBAL batch) */
    const double w = per * (h->cfg.S - 1) * (double)h->cfg.B / (double)h->global_B;
    for (int64_t i = 0; i < count; i++)
        if (losses) losses[i] = w;
    const uint64_t span = h->text_len - (uint64_t)h->cfg.S;
    for (int b = 0; b < h->cfg.B; b++) h->pos[b] = (uint64_t)h->cfg.S + ((h->pos[b] - h->cfg.S) + (uint64_t)count * h->stride) % span;
    if (ms) *ms = 0.0f;
    logf_(h, "train_windows %lld", (long long)count);
    return 0;
}
int lstm_hip_eval_bits(lstm_hip_t *h, const uint8_t *t, size_t len, double *bits) { (void)t, (void)len; *bits = 1.5; logf_(h, "eval_bits"); return 0; }
int lstm_hip_sample(lstm_hip_t *h, float *h0, float *c0, const double *u, int32_t n, uint8_t *out) {
    (void)h0, (void)c0, (void)u;
    memset(out, 'a', (size_t)n);
    logf_(h, "sample %d", n);
    return 0;
}
