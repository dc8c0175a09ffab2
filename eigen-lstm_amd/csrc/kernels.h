// kernels.h -- launch wrappers of the gfx950 kernels (kernels.hip), used by lstm_hip_api.cpp.
// Every wrapper enqueues on `st` and returns; no host synchronisation, no allocation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lstmk {

// flat parameter block offsets [W | U | b | Why | by], floats
struct ParamLayout {
    size_t W, U, b, Why, by, total;
    __host__ __device__ static ParamLayout make(int N, int M) {
        ParamLayout p;
        p.W = 0;
        p.U = p.W + (size_t)4 * N * M;
        p.b = p.U + (size_t)4 * N * N;
        p.Why = p.b + (size_t)4 * N;
        p.by = p.Why + (size_t)M * N;
        p.total = p.by + (size_t)M;
        return p;
    }
};

// ---- recurrent weight repack (once per window, after Adagrad) -------------------------------
// Ufwd[N/4][N/16][64] float4 : MFMA 16x16x4 A-fragments of U for the forward product
// Ubwd[N/16][N/4][64] float4 : A-fragments of U^T for the backward product
// Ubwd4 / Ufwd4 (optional): the 4x4x1 images of the 8-column backward / forward forms; half_forms: Ufwd4 receives the
// image of the two-half forward form (k_fwd_persistent6) instead
void pack_U(const float *U, float4 *Ufwd, float4 *Ubwd, int N, hipStream_t st, float4 *Ubwd4 = nullptr,
            float4 *Ufwd4 = nullptr, int half_forms = 0);
bool fwd_uses_8col_form(int N, int B, int n_cus); // forward recurrence on 8-column groups (k_fwd_persistent4, Ufwd4 image)
bool bwd_uses_m4(int N, int cols, bool bf16);     // backward recurrence on v_mfma_f32_4x4x1 (8-column groups, fp32, Ubwd4 image)

// ---- baseline engine: one launch per timestep -----------------------------------------------
// g = U*h_prev + W[:,x] + b ; gates ; c = tanh(i*u + f*c_prev) ; h = o*c      (R/lstm.cc:176-192)
void fwd_step(const float4 *Ufwd, const float *W, const float *bias, const float *Hprev, const float *Cprev,
              float *Hout, float *Cout, float *Gout, const int32_t *xi_t, int N, int B, bool fast, hipStream_t st);
// dh = DHy[t] + U^T*dg[t+1] ; dc ; dg[t] ; dcnext                              (R/lstm.cc:228-256)
void bwd_step(const float4 *Ubwd, const float *DGnext /*null at t=S-1*/, const float *DHy_t, const float *G_t,
              const float *C_t, const float *Cprev, float *dcnext, float *DG_t, int N, int B, hipStream_t st);

// ---- default engine: each recurrence of a window as ONE persistent launch (persistent.hip) ----
// Weights stay in VGPRs; the steps are chained inside the launch.  Hand-off, by form:
//   8-column forward form and (optionally) the fp32 4x4x1 backward form: data-as-flag through a ring of sentinel-filled
//     step slots (Hx / DGx; *_ring_floats() floats, filled with 0xFF bytes once and after an abort; ring_base starts at 0
//     and moves by *_ring_advance() after every launch);
//   every other form: sc1 stores + sharded device-scope counters.  `cnt` must hold persistent_counter_bytes() bytes
//     (separate regions for fwd and bwd), zeroed once; `epoch` = 1, 2, ... counts the launches that used that region
//     (counters are cumulative).  The ring forms use the step-0 slots of `cnt` for their XCD placement check.
// `abortp` is one zeroed word that a timed-out spin sets.  `stamps` (diagnostic builds, N = 512 only): [2][S][16] u64.
size_t persistent_counter_bytes(int S, int B);
bool persistent_supported(int N, int B, int n_cus, bool fused);      // fused: dW/db/DHy/dWhy inside the backward recurrence
bool persistent_supported_bf16(int N, int B, int n_cus, bool fused); // the bf16 recurrence's own kernels and grids
int bwd_group_cols_bf16(int N, int B, int n_cus, bool fused);        // 8 where that grid is co-resident, else 16
void fwd_persistent(const float4 *Ufwd, const float *W, const float *bias, float *H, float *C, float *G,
                    const int32_t *xi, unsigned *cnt, unsigned *abortp, unsigned epoch, int N, int S, int B, bool fast,
                    hipStream_t st);
size_t fwd_ring_floats(int N, int B);
int fwd_ring_advance(int ring_base, int S);
void fwd_persistent4(const float4 *Ufwd4, const float *W, const float *bias, float *H, float *C, float *G, const int32_t *xi,
                     float *Hx, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N, int S, int B, bool fast,
                     int poll_cfg, hipStream_t st, unsigned long long *stamps = nullptr);
// two-half form (N = 512): the same grid and ring, each workgroup's eight columns advanced as two alternating 4-column
// recurrences; weights in the Ufwd5 image (pack_U / adagrad with half_forms)
bool fwd_uses_two_half_form(int N, int B, int n_cus);
void fwd_persistent6(const float4 *Ufwd5, const float *W, const float *bias, float *H, float *C, float *G, const int32_t *xi,
                     float *Hx, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N, int S, int B, bool fast,
                     int poll_cfg, hipStream_t st, unsigned long long *stamps = nullptr, int col0 = 0, int cols = 0);
// columns one launch of the fp32 two-half forms takes; a wider batch (two_half_wide) runs as launches over column ranges
// [col0, col0 + cols) -- cols = 0: the whole batch
int two_half_launch_cols(int N, int n_cus);
int two_half_group_cols(int N, int B, int n_cus);     // forward: 8, or 4 (one half per workgroup) where the batch then fits one launch
int bwd_scatter_group_cols(int N, int B, int n_cus);  // backward: the same rule; the fused partial blocks are one per group
bool two_half_wide(int N, int B, int n_cus);
// two-half (scatter) form of the backward recurrence (N = 512 / 256, 8-column groups): every workgroup advances its eight
// columns as two alternating 4-column recurrences, multiplies its OWN dg_t into partial sums for all N outputs and scatters
// them to the owners of the outputs.  Ubwd6 image (pack_U / adagrad with bit 2 of half_forms), partial-sum ring Qx
// (bwd_ring_floats floats, sentinel-filled like the other rings; ring_base moves by bwds_ring_advance); computes Why^T dy
// itself; gpart != null: fused mode as below.  cfg: tuning / test bits (16: keep the dispatch-order workgroup mapping).
bool bwd_scatter_supported(int N, int B, int n_cus, bool fused);
int bwds_ring_advance(int ring_base, int S);
void bwd_scatter(const float4 *Ubwd6, float *DG, const float *Why, const float *dY, const float *G, const float *C, const float *H,
                 const int32_t *xi, float *gpart, float *Qx, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N,
                 int S, int B, int cfg, hipStream_t st, unsigned long long *stamps = nullptr, int col0 = 0, int cols = 0);
// gpart != null (8-column groups only): fused mode.  The recurrence then also produces DHy on the fly from
// Why and dY (DHy is not read), and leaves per-column-group partial blocks [dW | - | db | dWhy]
// (bwd_partial_floats(N) floats each) to be folded in group order; H and xi are read as well.
void bwd_persistent(const float4 *Ubwd, float *DG, const float *DHy, const float *G, const float *C, const float *H,
                    const int32_t *xi, float *gpart, const float *Why, const float *dY, unsigned *cnt, unsigned *abortp,
                    unsigned epoch, int N, int S, int B, int cols, hipStream_t st, unsigned long long *stamps = nullptr,
                    unsigned short *DGb = nullptr);
size_t bwd_ring_floats(int N, int B);
int bwd_ring_advance(int ring_base, int S);
// bf16 recurrence (N % 128 == 0): bf16 fragment images of U (N*N*8 bytes each), h and dg also kept as bf16
// hand-off copies Hb [S][B][N], DGb [S][B][4N]; bwd_persistent takes the Ubwd16 image as `Ubwd` and DGb != null
void pack_U_bf16(const float *U, void *Ufwd16, void *Ubwd16, int N, hipStream_t st);
void fwd_persistent_bf16(const void *Ufwd16, const float *W, const float *bias, float *H, unsigned short *Hb, float *C,
                         float *G, const int32_t *xi, unsigned *cnt, unsigned *abortp, unsigned epoch, int N, int S, int B,
                         bool fast, hipStream_t st, int n_cus);
// scatter form of the bf16 backward recurrence (N = 256 / 512 / 1024, 8-column groups co-resident): Ubwd6b image (pack_U6_bf16,
// N*N*8 bytes), partial-sum ring Qx as in bwd_scatter; reads DHy (a launch of its own in the bf16 path), writes the fp32 DG
bool bwd_scatter_bf16_supported(int N, int B, int n_cus);
void pack_U6_bf16(const float *U, void *Ubwd6b, int N, hipStream_t st);
// two-half form of the bf16 forward recurrence (k_fwd_halves_bf16): weights image Ufwd6b (N*N*8 bytes), bf16 sentinel ring Hxb
// (fwd_halves_bf16_ring_halfwords, all ones at rest; slots advance with fwd_ring_advance)
bool fwd_halves_bf16_supported(int N, int B, int n_cus);
void pack_Ufwd6_bf16(const float *U, void *img, int N, hipStream_t st);
size_t fwd_halves_bf16_ring_halfwords(int N, int B);
void fwd_halves_bf16(const void *Ufwd6b, const float *W, const float *bias, float *H, unsigned short *Hb, float *C, float *G,
                     const int32_t *xi, void *Hxb, unsigned *cnt, unsigned *abortp, unsigned epoch, int ring_base, int N, int S,
                     int B, int col0, int cols, bool fast, int n_cus, hipStream_t st, unsigned long long *stamps);
int fwd_halves_bf16_launch_cols(int N, int B, int n_cus); // a launch takes this many columns; wider batches run as several launches
size_t bwd_scatter_bf16_ring_floats(int N, int B, int n_cus);
int bwd_scatter_bf16_launch_cols(int N, int B, int n_cus);
int bwd_scatter_bf16_units(int N);
int bwd_scatter_bf16_ring_advance(int base, int S);
void bwd_scatter_bf16(const void *Ubwd6b, float *DG, const float *DHy, const float *G, const float *C, float *Qx, unsigned *cnt,
                      unsigned *abortp, unsigned epoch, int ring_base, int N, int S, int B, int col0, int cols, int n_cus, hipStream_t st,
                      unsigned long long *stamps, unsigned short *DGt_b = nullptr, int Tpad = 0);
// one stream, hidden 64 / 128: both recurrences on one CU (k_small_fwd, k_small_bwd); they read U and Why as stored, write H, C,
// G and DG (the backward one reads U from the Ubwd tile image); the backward one computes Why^T dy itself (no DHy), dW / db / dWhy / dU are the unfused path's launches
bool small_recurrence_supported(int N, int B);
void small_fwd(const float *U, const float *W, const float *bias, float *H, float *C, float *G, const int32_t *xi, int N, int S, bool fast,
               hipStream_t st);
void small_bwd(const float4 *Ubwd, const float *Why, const float *dY, const float *G, const float *C, float *DG, int N, int S, hipStream_t st);
size_t bwd_partial_floats(int N);
int bwd_group_cols(int N, int B, int n_cus); // 8 or 16 batch columns per backward workgroup

// ---- time-batched dense products (gemm.hip: fp32 MFMA 32x32x2, operands streamed into registers, K split over the waves
//      of a workgroup) --------------------------------------------------------------------------
// C[M x Nn] = op(A)[M x K] * op(B)[K x Nn], column-major; TA: A is stored K x M; TB: B is stored Nn x K.  (TA && TB is
// not a product of the window and is refused.)  splits > 1 writes partial slabs into `slabs` (each M*Nn floats, ld = M)
// and gemm_fold sums them into C in slab order (deterministic).  `slabs` must hold splits*M*Nn floats.
void gemm(bool TA, bool TB, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
          int splits, float *slabs, hipStream_t st);
// how many slabs the shape rule wants (1: no slabs); a static function of the shape and the device's compute-unit count
int gemm_pick_splits(bool TA, bool TB, int M, int Nn, int K, int n_cus);
// the ordered fold of `splits` slabs
// slab_stride: floats between consecutive slabs (0 = M*Nn, i.e. densely packed)
void gemm_fold(const float *slabs, int splits, int M, int Nn, float *C, int ldc, hipStream_t st, size_t slab_stride = 0);
// the split-K product without its fold (slabs densely packed, M*Nn floats each); returns the number of slabs written
int gemm_slabs(bool TA, bool TB, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *slabs, int splits,
               hipStream_t st);
void gemm_probe_small_kfast(int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc, hipStream_t st);
// the same by operand layout ("k fast": the contraction index is the contiguous one); gemm / gemm_slabs map onto these
int gemm_regs_splits(bool a_kfast, bool b_kfast, int M, int Nn, int K, int n_cus);
int gemm_regs(bool a_kfast, bool b_kfast, int M, int Nn, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
              int splits, float *slabs, hipStream_t st);

// ---- bf16 time-batched products (LSTM_HIP_BF16_RECURRENCE): C[m + ldc*n] = sum_k A[m][k] * B[n][k], fp32 accumulate on
//      v_mfma_f32_32x32x16_bf16.  A, B: bfloat16, k contiguous (lda, ldb in elements, multiples of 8; 16-byte aligned
//      bases), K a multiple of 64 (zero-padded images).  transpose_pack_bf16 builds such an image from a column-major fp32
//      matrix whose COLUMNS are the contraction index: dst[r][k] = bf16(src[k*ld + r]), zero for K <= k < Kpad.
void gemm_bf16(int M, int Nn, int K, const unsigned short *A, int lda, const unsigned short *B, int ldb, float *C, int ldc,
               int splits, float *slabs, hipStream_t st);
int gemm_bf16_pick_splits(int M, int Nn, int K);
void transpose_pack_bf16(const float *src, int K, int R, int ld, unsigned short *dst, int Kpad, hipStream_t st);
void pack_bf16(const float *src, size_t n, unsigned short *dst, hipStream_t st);

// ---- output layer elementwise: probs = exp(y+by)/sum ; loss ; dy = probs - onehot  (R/lstm.cc:195-207,225)
// Y is [T cols][256] (column-major 256 x T) and is overwritten by dY; probs written to P.
// colloss[col] = -log2 p[target] (0 for an empty target); dby_part[wave][256] partial row sums of dY.
// Processes columns [col0, col1) (col0 a multiple of 8) with global indexing, so a window can be done in time chunks.
int softmax_parts(int T);
void softmax_loss_dy(float *Y, float *P, const float *by, const int32_t *ti, float *colloss, float *dby_part, int col0,
                     int col1, hipStream_t st);
// window loss as the reference sums it: for each t a float sum over b, / B_global, accumulated in double;
// when dby != null a second workgroup folds the per-wave partials into dby = rowsum(dY) (R/lstm.cc:227)
void loss_reduce(const float *colloss, int steps, int B, int B_global, double *out, const float *dby_part, int n_parts,
                 float *dby, hipStream_t st, float scale = 1.0f);

// ---- dW = DG * X^T and db = rowsum(DG)                                        (R/lstm.cc:251-252)
// X is one-hot, so dW[:,v] is the sum of the DG columns whose input byte is v (bucket 256 = empty
// input column); db[r] = sum over the 257 buckets.  Stable counting sort of the column ids, ordered
// chunk sums, ordered folds: deterministic.  `scratch` must hold dW_scratch_bytes(T, G4).
constexpr int DW_CHUNK = 32;
size_t dW_scratch_bytes(int T, int G4);
void dW_db(const float *DG /*[T][G4]*/, const int32_t *xi /*[T]*/, int T, int G4, float *dW /*[256][G4]*/, float *db,
           void *scratch, hipStream_t st);
// the same in two parts: the sort needs only the input bytes, the sums need the complete DG
void dW_sort(const int32_t *xi, int T, int G4, void *scratch, hipStream_t st);
void dW_sums(const float *DG, int T, int G4, float *dW, float *db, void *scratch, hipStream_t st);

// ---- Adagrad over the flat block (R/lstm.cc:261-272; eps added in double, :25,46-48)
// When Ufwd/Ubwd are given, the U block also refreshes both MFMA fragment images (fused pack_U).
// gpart != null: the gradient is still in pieces -- `n_groups` partial blocks [dW | - | db | dWhy] (group_stride floats apart)
// and, when slabs != null, `n_slabs` split-K slabs of dU; they are summed here in the order the separate folds use and the
// sums are also stored to dP.  by_off: float offset of dby in the flat block (dby is final in dP).
// the NEXT window's slide (slide_window's arguments), carried by an Adagrad launch in extra workgroups: inside the window loop
// the slide of window i+1 needs nothing Adagrad of window i produces and touches nothing it reads
struct SlideJob {
    const uint8_t *text;
    uint64_t len;
    uint64_t *pos;
    int32_t *Xr, *Tr, *headp, *xi, *ti;
    float *H, *C;
    int S, B, N, stride, carry_col;
};
void adagrad(float *P, float *dP, float *mem, size_t n, float lr, size_t u_off, int N, float4 *Ufwd, float4 *Ubwd,
             hipStream_t st, float4 *Ubwd4 = nullptr, float4 *Ufwd4 = nullptr, const float *gpart = nullptr, int n_groups = 0,
             size_t group_stride = 0, size_t by_off = 0, const float *slabs = nullptr, int n_slabs = 0, size_t slab_stride = 0,
             int half_forms = 0, void *u6b = nullptr, int u6_uw = 0, unsigned short *why_b = nullptr,
             unsigned short *whyT_b = nullptr, size_t why_off = 0, const SlideJob *slide = nullptr, void *uf6b = nullptr,
             int uf6_uw = 0);
int fwd_halves_bf16_units(int N);

// ---- window builder on the device (OV/lstm_eigen_opt/lstm.cc:190-213): x/target rings + flat copies,
//      cursor advance, and the h/c carry (column 1 -> column 0).  Single workgroup.
void slide_window(const uint8_t *text, uint64_t len, uint64_t *pos, int32_t *Xr, int32_t *Tr, int32_t *headp,
                  int32_t *xi, int32_t *ti, float *H, float *C, int S, int B, int N, int stride, int carry_col,
                  hipStream_t st);

// ---- B = 1 recurrence for the evaluator / sampler (OV/lstm_eigen_class_CUDA/lstm.cc:578-720)
void eval_bits(const float *P, int N, const uint8_t *text, uint64_t len, double *out_bits_sum, float *scratch,
               hipStream_t st);
void sample(const float *P, int N, float *hc /*2N*/, const double *u, int count, uint8_t *out, float *scratch,
            hipStream_t st);
void sample_head(const float *Why, const float *by, int N, const float *hvec, const double *u, uint8_t *out, int32_t *x_next,
                 hipStream_t st);

} // namespace lstmk
