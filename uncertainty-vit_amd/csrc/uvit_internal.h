// Internal launch interfaces shared by the .hip translation units of libuvit (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { EPI_BF16 = 0, EPI_QKV = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_F32 = 4, EPI_PATCH = 5, EPI_DGELU = 6, EPI_QKV_ELU = 7,
       EPI_GELU_DG = 8,    // out = gelu(h), out2 = gelu'(h) (bf16): what backward needs of h, computed beside gelu (shared erf / exp)
       EPI_MULAUX = 9 };   // out = acc * aux (bf16): GELU backward against the stored gelu'(h), no transcendental work

struct GemmEpi {
    void* out = nullptr;             // bf16 or f32 [M, ldo]
    void* out2 = nullptr;            // GELU: pre-activation h (bf16); RESID: branch output before gamma (bf16)
    const float* bias = nullptr;     // [N]   (QKV: q_bias [N/3])
    const float* bias2 = nullptr;    // QKV: v_bias [N/3]
    const float* gamma = nullptr;    // RESID: LayerScale [N]
    const float* resid = nullptr;    // RESID: residual stream [M, ldo] f32
    const float* rowscale = nullptr; // RESID: per-sample drop-path multiplier [B] (null = 1)
    const void* aux = nullptr;       // DGELU: h [M, ldo] bf16
    const int64_t* mask = nullptr;   // PATCH: bool_masked_pos flattened [B*P] (null = teacher)
    const float* mask_token = nullptr;
    int ldo = 0;
    int tokens = 1;                  // RESID: rows per sample
    int patches = 1;                 // PATCH: rows per sample in the GEMM
    int row0 = 0;                    // RESID: sample index of GEMM row m is (m + row0) / tokens (row-split launches)
    int ngroup = 0;                  // 256x256 kernel: column tiles per row-tile group of the XCD-aware tile order (0 = 6)
    // RESID over a COMPACT row list (the student's last MLP on the masked rows only): GEMM row m stands for residual-stream row
    // rowmap[m] -- resid is read and out / out2 are written THERE, the drop-path sample is rowmap[m] / tokens; rows m >= *rowcount
    // are padding and write nothing
    const int* rowmap = nullptr;
    const int* rowcount = nullptr;
    // persistent 256x256 kernel: 16 zero-initialised counters (8 per-XCD tile counters, 1 exit counter) in device memory -> the workgroups
    // take their tiles from the counters instead of by a fixed stride (nullptr: fixed stride); the last workgroup to leave zeroes them again
    unsigned* tile_counter = nullptr;
};

// gemm.hip
// Launch tuning (mirrors uvit_tuning in include/uvit.h).  Passed by the caller on every launch: the launchers keep no
// mutable process-wide state, so host threads launching on different streams do not interact.
struct GemmTune {
    int nt_variant = 3;      // 0: 128x128, 1: 256x256 staggered (1 WG/CU), 5: 320x256, 3: auto by shape
    int tn_variant = 3;      // 0: 128x128 kernel, 1: 256x256 staggered kernel, 3: auto
    int tn_target = 512;     // workgroups the wgrad split-K aims for (MI355X sweep: 512 beats 256..1536 on all four wgrad shapes)
    int group_chunks = 0;    // grouped wgrad: 0 = cost model, > 0 = forced token-chunk count
    int nt_group = 0;        // 256x256 NT kernel: column tiles per row-tile group of the tile order (0 = default 6)
    int nt_persist = 1;      // 256x256 NT kernel, more tiles than CUs: persistent workgroups, operand pipeline across tiles
};
// tune == nullptr: defaults.  tail_rows_out (nullable) receives the rows that went to the row-split tail launch.
int uvit_gemm_nt_launch(int mode, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                        const GemmEpi* epi, hipStream_t s, const GemmTune* tune = nullptr, int* tail_rows_out = nullptr);
// allow_split: partial sums are combined with fp32 atomics -> C must be zero (or hold a value to add to)
int uvit_gemm_tn_launch(const void* Y, const void* X, int M, int Nn, int Kk, int ldy, int ldx, float* C,
                        int ldc, int allow_split, hipStream_t s, const GemmTune* tune = nullptr);

// grouped wgrad (all Linear weight gradients of one layer in one launch, bias column sums fused)
#define UVIT_TN_GROUP_MAX 6
struct TnProb {
    const void* Y = nullptr;         // dY [M, ldy] bf16 (rows beyond the real tokens are zero)
    const void* X = nullptr;         // layer input [M, ldx] bf16
    float* C = nullptr;              // dW [Nn, ldc] fp32, accumulated into (zeroed arena)
    float* bias = nullptr;           // column sums of Y[:, 0:bias_end) (null: none)
    float* bias2 = nullptr;          // column sums of Y[:, bias2_begin:Nn) (qkv: v_bias; null: none)
    int bias_end = 0, bias2_begin = 0;
    // stacked two-stream operands (rows [0, s1_row) = mean stream, [s1_row, M) = covariance stream) whose bias differs per stream: the
    // column sums of the rows from s1_row on go to bias_s1 / bias2_s1 (s1_row = 0: one stream); every token chunk must lie in ONE stream
    float* bias_s1 = nullptr; float* bias2_s1 = nullptr; int s1_row = 0;
    int M = 0, Nn = 0, Kk = 0, ldy = 0, ldx = 0, ldc = 0;
    // filled by the launcher
    int nm = 0, tiles_n = 0, tiles_k = 0, chunk_steps = 0, chunks = 0;
};
struct TnGroup { TnProb p[UVIT_TN_GROUP_MAX]; int nprob = 0; int max_chunks = 0; };
bool uvit_gemm_tn_group_ok(const TnProb* probs, int n, const GemmTune* tune = nullptr);
int uvit_gemm_tn_group_launch(const TnProb* probs, int n, hipStream_t s, const GemmTune* tune = nullptr);

// attention.hip
// bmap != nullptr: the launch runs a COMPACT batch (drop-path sample list); sample slot b draws the dropout of sample bmap[b]
int uvit_attn_fwd_launch(const void* qkv, const float* biasP, void* out, float* lse, int B, int H, int N, int NP,
                         float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s, const int* bmap = nullptr);
// fused backward (one recomputation of P; dS leaves as bf16 for the bias gradient): ds_ws holds uvit_attn_bwd_fused_ws_bytes()
// bytes and is written when want_ds != 0; uvit_attn_dbias_reduce_launch sums it over the batch into ONE [H][NP][NP] slab laid out
// [h][key][q] (accumulate = 0: the slab is zero-filled first).  Two launches so that the reduction can run on another stream.
size_t uvit_attn_bwd_fused_ws_bytes(int B, int H, int N);
int uvit_attn_bwd_fused_launch(const void* qkv, const void* o_fwd, const void* d_o, const float* biasP, const float* lse,
                               float* delta, void* dqkv, void* ds_ws, int want_ds, int B, int H, int N, int NP, float scale,
                               float p_drop, uint32_t seed, uint32_t layer, hipStream_t s, const int* bmap = nullptr);
int uvit_attn_dbias_reduce_launch(const void* ds_ws, float* dbias_slab, int accumulate, int B, int H, int N, int NP, hipStream_t s);

// attention2.hip (two-stream Wasserstein attention)
int uvit_attn2_fwd_launch(const void* qkv_m, const void* qkv_c, const float* biasP, void* out_m, void* out_c, float* lse,
                          int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s);
// fused backward (round 4; same contract as the base model's): dS tiles (bf16) go to ds_ws when want_ds != 0 and
// uvit_attn2_dbias_reduce_launch sums them over the batch into ONE [H][NP][NP] slab laid out [h][key][q]
size_t uvit_attn2_bwd_ws_bytes(int B, int H, int N);
int uvit_attn2_bwd_launch(const void* qkv_m, const void* qkv_c, const void* o_m, const void* o_c, const void* d_m, const void* d_c,
                          const float* biasP, const float* lse, float* delta, void* dqkv_m, void* dqkv_c, void* ds_ws, int want_ds,
                          int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s);
int uvit_attn2_dbias_reduce_launch(const void* ds_ws, float* dbias_slab, int accumulate, int B, int H, int N, int NP, hipStream_t s);

// norm.hip
int uvit_ln_fwd_launch(const float* x, const float* w, const float* b, void* y_bf16, float* mean, float* rstd,
                       int M, int C, float eps, hipStream_t s);
int uvit_ln_fwd_gather_launch(const float* x, const int* rowidx, const int* count, const float* w, const float* b,
                              void* y_bf16, float* mean, float* rstd, int Mmax, int C, float eps, hipStream_t s);
int uvit_ln_bwd_launch(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* w,
                       const float* dres, float* dx, float* dw, float* db, int M, int C, int nrep, size_t rep_stride,
                       hipStream_t s);
// drop-path sample lists (norm.hip): dense walk of the residual stream, compact branch buffers
int uvit_ln_fwd_keep_launch(const float* x, const int* pos, const float* w, const float* b, void* y, float* mean, float* rstd,
                            float* xcopy, int M, int C, int tokens, float eps, hipStream_t s);
int uvit_ln_bwd_keep_launch(const void* dy, const float* x, const int* posA, const float* mean, const float* rstd, const float* w,
                            const float* dres, float* dx, float* dw, float* db, const void* y_next, const float* gamma_next,
                            const float* rowscale_next, void* dy_next, float* dgamma_next, float* dbias_next, const int* posB,
                            const int* cntB, int tokens, int M, int C, int nrep, size_t rep_stride, hipStream_t s, int pad_base = 0,
                            void* pad2 = nullptr, int pad2_cols = 0);     // pad2: a second bf16 buffer [rows][pad2_cols] whose pad rows (same range) are zero-filled
// LayerNorm backward fused with the LayerScale + DropPath backward of the branch that consumes dx next
int uvit_ln_bwd_ls_launch(const void* dy, const float* x, const float* mean, const float* rstd, const float* w,
                          const float* dres, float* dx, float* dw, float* db, const void* y_next, const float* gamma_next,
                          const float* rowscale_next, void* dy_next, float* dgamma_next, float* dbias_next, int tokens,
                          int M, int C, int nrep, size_t rep_stride, hipStream_t s,
                          const int* rowidx = nullptr, const int* count = nullptr,    // row list: dy / mean / rstd are compact, everything else lives at rowidx[row]
                          const int* pos_next = nullptr);                              // (row list only) dy_next compact by this sample map
int uvit_ln_bwd_scatter_launch(const void* dy_bf16, const float* x, const int* rowidx, const int* count,
                               const float* mean, const float* rstd, const float* w, float* dx, float* dw, float* db,
                               int Mmax, int C, int nrep, size_t rep_stride, hipStream_t s);
int uvit_reduce_replicas_launch(const float* rep, float* out, size_t n, int nrep, size_t stride, hipStream_t s);
int uvit_target_accum_launch(const float* x, const int* rowidx, const int* count, float* acc, int first, int Mmax,
                             int C, float eps, hipStream_t s, const float* sub = nullptr,     // sub: rows subtracted before the LayerNorm
                             int ln = 1);                                                   // ln = 0: no LayerNorm, the rows are summed as they are
int uvit_variance_loss_launch(const float* out, const int* count, float w, float margin, float loss_scale, float* scratch,
                              float* loss, float* std_loss0_out, void* dout_bf16, int Mmax, int C, hipStream_t s);
int uvit_gather_patch_rows_launch(const float* x, const float* sub, float* v, int B, int P, int C, hipStream_t s);
int uvit_colnorm_launch(float* v, int groups, int rows, int C, float eps, hipStream_t s);
int uvit_axpy_rows_launch(float* acc, const float* v, int first, size_t n, hipStream_t s);
int uvit_gather_masked_rows_launch(const float* dense, const int* rowidx, const int* count, float* out, int Mmax, int P, int C,
                                   hipStream_t s);
int uvit_target_finalize_launch(float* acc, const int* count, int n_layers, int post_ln, int Mmax, int C, float eps,
                                hipStream_t s);

// elementwise.hip
int uvit_im2col_launch(const float* img, void* cols_bf16, int B, int Cin, int img_size, int patch, hipStream_t s);
int uvit_mask_compact_launch(const int64_t* mask, int* rowidx, int* count, int B, int P, hipStream_t s);
int uvit_set_cls_launch(float* x, const float* cls, const float* pos, int B, int N, int C, hipStream_t s);
int uvit_relpos_gather_launch(const float* table, const int* index, float* biasP, int H, int N, int NP, hipStream_t s);
int uvit_relpos_scatter_launch(const float* slab, int nslab, const int* index, float* dtable, int H, int N, int NP,
                               hipStream_t s);
int uvit_ls_bwd_launch(const float* dx, const void* branch_bf16, const float* gamma, const float* rowscale,
                       void* dy_bf16, float* dgamma, float* dbias, int M, int C, int tokens, int nrep, size_t rep_stride,
                       hipStream_t s, const int* rowidx = nullptr, const int* count = nullptr);   // row list: dy is compact, dx / branch live at rowidx[row]
int uvit_rows_guard_launch(const int* count, int limit, float* loss, hipStream_t s);   // *count > limit: loss <- NaN (the step is then skipped like any non-finite one)
int uvit_colsum_launch(const void* y_bf16, int ld, int col0, int ncols, int M, float* out, int nrep, size_t rep_stride,
                       hipStream_t s);
int uvit_smooth_l1_launch(const float* out, const float* target, const int* count, float beta, int l2, float loss_scale,
                          float* loss, void* dout_bf16, int Mmax, int C, hipStream_t s);
int uvit_token_bwd_launch(const float* dx, const int64_t* mask, void* dpatch_bf16, float* dcls, float* dmask_token,
                          int B, int P, int C, hipStream_t s);
int uvit_transpose_batch_launch(const void* descs_dev, int ndesc, int max_tiles, hipStream_t s);
int uvit_droppath_launch(float* scales, const float* rates_dev, int depth, int nbr, int B, uint32_t seed, uint32_t step,
                         hipStream_t s);
// drop-path sample lists of a step (elementwise.hip): per (layer, branch) list lb: pos[lb][b], bmap[lb][slot], rows[lb][r] (stride rows_stride,
// pad rows -1), cnt[lb] = kept rows, cnt[nlists + lb] = 1 when the kept SAMPLES differ from host_counts[lb], what the host sized the launches
// with; uvit_droppath_lists_guard_launch turns any such flag into a NaN loss
int uvit_droppath_lists_launch(const float* scales, int* pos, int* bmap, int* rows, int* cnt, int nlists, int B, int tokens,
                               int rows_stride, const int* host_counts, hipStream_t s);
int uvit_droppath_lists_guard_launch(const int* cnt, int nlists, float* loss, hipStream_t s);
int uvit_add_pos_launch(float* x, const float* pos, int B, int N, int C, hipStream_t s);
int uvit_pos_bwd_launch(const float* dx, float* dpos, int B, int N, int C, hipStream_t s);
int uvit_synth_batch_launch(float* images, int64_t* mask, int B, int chans, int img_size, int patches, int n_mask, uint32_t seed,
                            uint32_t it, hipStream_t s);
int uvit_poison_if_nonfinite_launch(const float* loss, float* dst, int* sticky, hipStream_t s);
int uvit_wasserstein_loss_launch(const float* out_m, const float* out_c, const float* tgt_m, const float* tgt_c, const int* count,
                                 float lam, float loss_scale, float* scratch, float* loss, void* dout_m_bf16, void* dout_c_bf16,
                                 int Mmax, int C, hipStream_t s);

// optim.hip
// guard_loss / guard_sumsq (nullable, device): the update is skipped when either holds a non-finite value
int uvit_ema_launch(float* ema, const float* p, void* ema_bf16, size_t n, float decay, hipStream_t s, const float* guard_loss = nullptr,
                    const double* guard_sumsq = nullptr);
int uvit_sumsq_launch(const float* g, size_t n, double* out, hipStream_t s);
int uvit_adamw_launch(float* p, const float* g, float* m, float* v, void* p_bf16, size_t n, size_t n_decay, float lr,
                      float wd, float b1, float b2, float eps, int step, const double* sumsq, float max_norm,
                      float grad_scale, float* gnorm_out, hipStream_t s, const float* guard_loss = nullptr,
                      float* ema = nullptr, void* ema_bf16 = nullptr, float ema_decay = 0.f,    // ema: fused EMA teacher update
                      int* sticky_poison = nullptr,    // set (and honoured) once a step's loss / norm was not finite
                      const float* sched_dev = nullptr, int sched_len = 0, int sched_idx = 0);   // device-resident {lr, wd, ema decay} tables
int uvit_zero_launch(void* dst, size_t bytes, hipStream_t s);
int uvit_cast_bf16_launch(const float* src, void* dst_bf16, size_t n, hipStream_t s);

struct TransposeDesc { const void* src; void* dst; int rows; int cols; int tile0; int pad; };
