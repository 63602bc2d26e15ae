/* libuvit -- C ABI of the MI355X-native data2vec ViT pre-training step.
 *
 * The reference (fx-erick/uncertainty-vit) is pure Python: it has no FFI / plugin boundary of
 * its own.  This header is therefore the NEW native boundary the build
 * defines; each entry point names the reference code it replaces.  Conventions:
 *   - plain pointers and sizes, no torch types; every pointer is DEVICE memory unless noted;
 *   - the caller allocates all buffers (arenas, workspace); nothing is allocated per call;
 *   - every call is asynchronous and ordered on the `stream` argument (a hipStream_t);
 *   - return value 0 = ok, negative = UVIT_ERR_*; nothing throws;
 *   - no global state besides the engine object.
 * Host-side mirror of the reference's Python API lives in uncertainty-vit_amd/ (ctypes).
 */
#ifndef UVIT_H
#define UVIT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UVIT_VERSION 100
#define UVIT_MAX_DEPTH 64

typedef struct uvit_engine uvit_engine;
typedef void* uvit_stream;   /* hipStream_t */

/* Constructor arguments of VisionTransformerForCyclicalTraining (modeling_cyclical.py:34-60)
 * plus the per-GPU batch the workspace is sized for. */
typedef struct uvit_config {
    int32_t img_size, patch_size, in_chans, embed_dim, depth, num_heads, mlp_hidden;
    int32_t use_shared_rel_pos_bias;  /* modeling_cyclical.py:84-89 */
    int32_t use_abs_pos_emb;          /* --abs_pos_emb (run_cyclical.py:55, default off): pos_embed (1, N, C) added after the cls concat */
    int32_t batch;
    float ln_eps;                     /* 1e-6, modeling_cyclical.py:294 */
    float attn_drop_rate;             /* --attn_drop_rate */
    float drop_path_rate;             /* --drop_path; per-layer linspace(0, rate, depth) */
    int32_t bias_chunk;               /* batch elements summed in registers per rel-pos-bias slab (0 = chosen to fill one round of dQ workgroups) */
    int32_t two_stream;               /* 1: DistVisionTransformerForCyclicalTraining (mean, cov) model, modeling_cyclical_dist.py:14-165 */
} uvit_config;

/* One tensor of the flat parameter arena; `name` is the reference state-dict key. */
typedef struct uvit_layout_entry {
    char name[96];
    int64_t offset;   /* in floats, multiple of 64 */
    int64_t numel;
    int32_t ndim;
    int32_t decay;    /* 1 = weight-decay group (optim_factory.py:58-97); 2 = never receives a gradient in the reference (kept constant) */
    int64_t shape[4];
} uvit_layout_entry;

/* Buffers owned by the caller.  fp32 arenas hold `uvit_arena_numel` floats; bf16 shadows the
 * same number of 2-byte elements.  Layout = [decay tensors | no-decay tensors]. */
typedef struct uvit_buffers {
    float* params;        /* student fp32 master weights */
    float* grads;         /* student gradients */
    float* adam_m;
    float* adam_v;
    float* ema;           /* EMA teacher fp32 (timm ModelEmaV2 at run_cyclical.py:503) */
    void* params_bf16;    /* bf16 shadow of params (GEMM operands) */
    void* params_bf16_t;  /* transposed bf16 copies of the 2-D Linear weights (dgrad operands) */
    void* ema_bf16;       /* bf16 shadow of the teacher */
    const int32_t* rel_index;  /* (N*N) relative_position_index as int32 (modeling_finetune.py:339-353) */
    void* workspace;
    int64_t workspace_bytes;
} uvit_buffers;

/* Per-step knobs of train_one_epoch (engine_for_cyclical.py:24-32, 47-56, 88-186). */
typedef struct uvit_step_params {
    int32_t target_layers[UVIT_MAX_DEPTH];
    int32_t n_target_layers;
    int32_t target_layer_norm_last;   /* not --no_target_layer_norm_last */
    int32_t post_target_layer_norm;
    int32_t l2_loss;
    float l1_beta;
    float loss_scale;                 /* -1 = off */
    float clip_grad;                  /* <= 0: no clipping (norm still reported) */
    float lr, weight_decay, beta1, beta2, eps;
    int32_t opt_step;                 /* 1-based AdamW step count */
    float ema_decay;                  /* cur_decay; EMA skipped when do_ema == 0 */
    int32_t do_ema;
    float grad_scale;                 /* multiplies gradients before clipping (1/world after a SUM all-reduce) */
    uint32_t seed;                    /* dropout / drop-path stream */
    uint32_t it;                      /* global iteration, decorrelates masks between steps */
    int32_t train_dropout;            /* 1: apply attn_drop_rate and drop_path_rate in the student */
    float lambda_pretraining;         /* WassersteinLoss weight (two-stream model only; --lambda_pretraining) */
    /* Device-resident schedules (utils.py:408-459 tables + the EMA decay anneal / tri-phase cut-off of
     * engine_for_cyclical.py:55-56,182-185): sched_dev = float[3][sched_len] = {lr, weight_decay, ema_decay} per iteration,
     * ema_decay < 0 meaning "skip EMA".  When non-NULL the optimizer kernels read entry `sched_index` on the device and the
     * scalar fields lr / weight_decay / ema_decay / do_ema above are ignored: the step's launch arguments then do not
     * depend on the iteration (a captured step can be replayed). */
    const float* sched_dev;
    int32_t sched_len, sched_index;
    /* flag-gated arithmetic of the step (no BASELINE config switches it on): */
    int32_t layer_results_fc;         /* --layer_results fc: targets from the teacher's MLP-branch outputs (modeling_cyclical.py:199-205) */
    float var_w0, var_margin0;        /* variance term, engine_for_cyclical.py:130-139,161 (var_w0 <= 0: off) */
    /* target-builder variants (engine_for_cyclical.py:94-118): affine-free batch norm over (B, T) per channel, instance norm
     * over T per (sample, channel) on every target layer; instance norm of the layer average.  Both models: with the two-stream model they act
     * on the MEAN targets (the covariance targets, :73-86, only know the two layer-norm flags). */
    int32_t target_batch_norm, target_instance_norm, post_target_instance_norm;
    /* Upper bound on the number of masked patches of this batch (the loader has bool_masked_pos on the host before the upload), 0 = not
     * given.  The loss reads the student at the masked rows only (modeling_cyclical.py:207,215-225), so with a bound the base model's last
     * block runs its MLP -- forward, dgrads and wgrads -- on those rows alone; results equal the all-rows step.  A bound BELOW the true count
     * makes the loss NaN (the step is then skipped like any non-finite one); the two-stream model ignores it (measured at the end of round 4:
     * the per-stream launches and the larger zero fills cost what the skipped rows save). */
    int32_t n_rows_hint;
} uvit_step_params;

int uvit_version(void);
/* First 16 hex digits of the SHA-256 over csrc/{*.hip,*.h} (sorted by name) + include/uvit.h that the library was built
 * from; the host side compares it with the sources on disk and refuses a stale build. */
const char* uvit_source_hash(void);

/* ---- arena layout (single source of truth for the host-side state-dict views) ---- */
int uvit_layout_count(const uvit_config* cfg);
int uvit_layout_get(const uvit_config* cfg, int index, uvit_layout_entry* out);
int64_t uvit_arena_numel(const uvit_config* cfg, int64_t* n_decay_out);
int64_t uvit_workspace_bytes(const uvit_config* cfg);

/* ---- engine ---- */
uvit_engine* uvit_engine_create(const uvit_config* cfg, const uvit_buffers* bufs, uvit_stream stream, int* err_out);
void uvit_engine_destroy(uvit_engine* e);

/* Refresh bf16 shadows (+ transposes) from the fp32 arenas: which = 1 student, 2 teacher, 3 both.
 * Needed after load_state_dict / init (replaces nothing in the reference; precision plumbing). */
int uvit_engine_sync_shadows(uvit_engine* e, int which, uvit_stream stream);

/* VisionTransformerForCyclicalTraining.forward_features (modeling_cyclical.py:170-207) for
 * `which` = 0 student / 1 teacher weights.  images (B,Cin,S,S) f32; mask (B,P) int64 or NULL.
 * Leaves every layer's residual stream in the workspace (see uvit_engine_ws_ptr). Batch may be
 * <= cfg.batch.  train_dropout applies attention dropout / drop-path with (seed, it). */
int uvit_engine_forward_features(uvit_engine* e, int which, const float* images, const int64_t* mask, int batch,
                                 int train_dropout, uint32_t seed, uint32_t it, uvit_stream stream);
/* norm + drop cls + (masked-row gather) + lm_head (modeling_cyclical.py:207,215-225) on the last
 * forward_features result.  all_tokens=1: out (B*P, C); else out (count, C) rows in mask order.
 * out must hold B*P*C floats; *count_dev (device int) receives the number of valid rows.
 * which: bit 0 = teacher weights, bit 1 = covariance stream (two-stream model: cov_lm_head). */
int uvit_engine_head(uvit_engine* e, int which, int all_tokens, float* out, int32_t* count_dev, uvit_stream stream);

/* Device pointers into the workspace after a forward: name in {"x" (layer 0..depth residual stream
 * (B,N,C) f32), "xm" (after the attention branch), "loss", "grad_norm", "targets", "outputs", "count"};
 * two-stream model: also "x_cov", "xm_cov", "targets_cov", "outputs_cov". */
void* uvit_engine_ws_ptr(uvit_engine* e, const char* name, int layer);
/* Rows the last training step ran its last block's MLP on (uvit_step_params.n_rows_hint): 0 = every token row. */
int uvit_engine_compact_rows(uvit_engine* e);

/* ---- the training step, engine_for_cyclical.py:58-186 ---- */
/* teacher forward (no grad) -> targets; student forward; loss; backward through lm_head + final norm */
int uvit_step_begin(uvit_engine* e, const float* images, const int64_t* mask, const uvit_step_params* hp,
                    uvit_stream stream);
/* backward of block `layer` (call depth-1 .. 0); its gradients are complete afterwards */
int uvit_step_backward_layer(uvit_engine* e, int layer, const uvit_step_params* hp, uvit_stream stream);
/* make `stream` wait until block `layer`'s weight gradients (computed on the engine's second stream) are final:
 * call on the communication stream before all-reducing that block's bucket */
int uvit_step_wait_layer_grads(uvit_engine* e, int layer, uvit_stream stream);
/* token assembly + patch embedding + relative-position table gradients */
int uvit_step_backward_embed(uvit_engine* e, uvit_stream stream);
/* clip_grad_norm_ + AdamW (utils.py:375-381) + EMA (engine_for_cyclical.py:182-185) + shadow refresh */
int uvit_step_update(uvit_engine* e, const uvit_step_params* hp, uvit_stream stream);
/* all of the above on one stream (single-GPU fast path) */
int uvit_train_step(uvit_engine* e, const float* images, const int64_t* mask, const uvit_step_params* hp,
                    uvit_stream stream);
/* Launch tuning.  It is an ARGUMENT (engine member / operator-call parameter), never process-wide state, so host
 * threads that launch on different streams cannot disturb one another.
 *   nt_variant: NT GEMM kernel for large shapes: 3 = auto by shape (default), 1 = 256x256 tile with staggered wave
 *               groups (one workgroup per CU), 5 = the same kernel with 320x256 tiles, 0 = 128x128 generic kernel,
 *               6 / 7 = ring kernel (two workgroups per CU, 3-deep 32-k ring) with 128- / 160-row tiles;
 *   tn_variant: wgrad (TN) GEMM: 3 = auto (default), 1 = 256x256 staggered kernel, 0 = 128x128;
 *   tn_split_target: workgroups the wgrad token split aims for (default 512);
 *   wgrad_group_chunks: token chunks per output tile of uvit_op_wgrad_group (0 = cost model, default);
 *   nt_group: 256x256 NT kernel: column tiles walked per row tile in the XCD-aware tile order (0 = default).
 *   nt_persist: 256x256 NT kernel with more tiles than CUs: 1 = one persistent workgroup per CU with the operand pipeline
 *               running across its tiles (default), 0 = one workgroup per tile. */
typedef struct uvit_tuning { int32_t nt_variant, tn_variant, tn_split_target, wgrad_group_chunks, nt_group, nt_persist; } uvit_tuning;
void uvit_tuning_default(uvit_tuning* out);
/* Replace the engine's tuning (values outside the lists above are refused with UVIT_ERR_ARG). */
int uvit_engine_set_tuning(uvit_engine* e, const uvit_tuning* t);
/* dual = 1 (default): teacher forward and the wgrad GEMMs run on an internal second HIP stream beside the
 * caller's stream; dual = 2: the same on a second stream of LOWER priority than the caller's (the critical chain of the
 * step gets free CUs first: faster with the whole GPU to itself, slower when other kernels -- RCCL -- hold CUs, so the
 * host mirror selects it for single-GPU runs only); dual = 0: everything on the caller's stream (used to time one
 * kernel in isolation).  Environment UVIT_SINGLE_STREAM=1 selects 0 at engine creation. */
int uvit_engine_set_streams(uvit_engine* e, int dual);
/* on = 1 (default): a training step of the base model runs every Block branch on the samples its DropPath KEPT (timm drop_path,
 * modeling_finetune.py:51-62: a dropped sample's branch is multiplied by 0 in the forward and receives no gradient), in compact rows
 * sized by the host from the same counter-based hash the device draws with; on = 0: every branch runs all samples and the dropped ones
 * are multiplied by 0, as the reference does.  Same results either way (bench.py reports both rates).  Environment UVIT_DP_ROWS=0
 * selects 0 at engine creation. */
int uvit_engine_set_drop_path_rows(uvit_engine* e, int on);
/* The host side of those lists, callable without a GPU: kept samples per (layer, draw) for one step -- out[draws_per_block * layer + k],
 * draws_per_block = 2 (base: attn, mlp) or 4 (two-stream: mean attn, mean mlp, cov attn, cov mlp) -- from the counter-based hash the device
 * draws its DropPath multipliers with (rates linspace(0, drop_path_rate, depth), modeling_cyclical.py:94-96).  Pure host arithmetic. */
int uvit_drop_path_kept_counts(int depth, float drop_path_rate, int draws_per_block, int batch, uint32_t seed, uint32_t it, int32_t* out);
/* Measurement aid for bench.py: bracket every launch of the dominant kernel (the fc1 GEMM with
 * fused bias+GELU, gemm_nt256_kernel<EPI_GELU / EPI_GELU_DG>) with HIP events on the stream it runs on.
 * profile_read: sum of the event-bracketed durations (ms), launch count, algorithmic FLOPs per launch. */
int uvit_engine_profile(uvit_engine* e, int enable, int max_launches);
int uvit_engine_profile_read(uvit_engine* e, double* total_ms, int* launches, double* flops_per_launch);
/* The brackets by kind: 0 = fc1 of the teacher (bias + GELU), 1 = fc1 of the student (also stores GELU'), 2 = attention proj and
 * 3 = fc2 (both: bias, LayerScale, DropPath, fp32 residual epilogue), -1 = kinds 0 and 1 together.  bytes_per_launch (nullable) =
 * the launch's algorithmic HBM bytes (operands read once, results written once).  Every launch of a kind is bracketed, compact ones
 * included (drop-path sample lists, the masked-row last block): FLOPs and bytes are those of the MEAN row count of the kind's launches. */
int uvit_engine_profile_read_kind(uvit_engine* e, int kind, double* total_ms, int* launches, double* flops_per_launch,
                                  double* bytes_per_launch);
/* enqueues the copy of 8 floats {loss, grad_norm, -, -, std_loss0 (the `loss_var0` meter), ...} of the last step into (pinned) host memory behind the step's kernels and returns:
 * the caller reads them after an event it records on `stream` has completed (engine_for_cyclical.py:164,186
 * sync twice per step instead).  A step whose loss or gradient norm is not finite leaves the weights untouched, and so
 * does every later step of that engine. */
int uvit_engine_read_stats_async(uvit_engine* e, float* host_out2, uvit_stream stream);
/* copies {loss, grad_norm} to host memory; synchronises the stream */
int uvit_engine_read_stats(uvit_engine* e, float* host_out2, uvit_stream stream);

/* ---- individual operators (tests, autograd wrappers) ---- */
typedef struct uvit_gemm_epilogue {
    void* out; void* out2; const float* bias; const float* bias2; const float* gamma; const float* resid;
    const float* rowscale; const void* aux; const int64_t* mask; const float* mask_token;
    int32_t ldo, tokens, patches;
    int32_t row0;   /* RESID: sample index of row m is (m + row0) / tokens */
} uvit_gemm_epilogue;
enum { UVIT_EPI_BF16 = 0, UVIT_EPI_QKV = 1, UVIT_EPI_GELU = 2, UVIT_EPI_RESID = 3, UVIT_EPI_F32 = 4,
       UVIT_EPI_PATCH = 5, UVIT_EPI_DGELU = 6, UVIT_EPI_QKV_ELU = 7,
       UVIT_EPI_GELU_DG = 8,   /* out = gelu(h), out2 = gelu'(h) (bf16), h = bf16(acc + bias): the training forward of Mlp.fc1 */
       UVIT_EPI_MULAUX = 9 };  /* out = bf16(acc * aux): GELU backward against the gelu'(h) stored by mode 8 */

/* C[M,N] = A[M,K] . W[N,K]^T with a fused epilogue: nn.Linear / F.linear sites of
 * modeling_finetune.py:75-82,151,186 and the Conv2d-as-GEMM at :317 */
int uvit_op_gemm_nt(int mode, const void* A_bf16, const void* W_bf16, int M, int N, int K, int lda, int ldw,
                    const uvit_gemm_epilogue* epi, uvit_stream stream);
/* The same with explicit tuning (NULL = defaults); *tail_rows (nullable, host) receives the number of rows the
 * launcher handed to its row-split tail launch (0 = single launch). */
int uvit_op_gemm_nt_tuned(int mode, const void* A_bf16, const void* W_bf16, int M, int N, int K, int lda, int ldw,
                          const uvit_gemm_epilogue* epi, const uvit_tuning* tune, int* tail_rows, uvit_stream stream);
/* The same with DYNAMIC tile assignment for the persistent 256x256 kernel (round 4): tile_counters = 16 uint32 in device memory, zero
 * before the first launch (every launch leaves them zero again; launches that may run concurrently need their own 16).  The workgroups
 * then take tiles from per-XCD counters instead of by a fixed stride, so a launch that finds CUs occupied by other work (RCCL's channel
 * workgroups in a data-parallel run) takes tiles / free CUs longer instead of twice as long.  NULL = fixed stride. */
int uvit_op_gemm_nt_sched(int mode, const void* A_bf16, const void* W_bf16, int M, int N, int K, int lda, int ldw,
                          const uvit_gemm_epilogue* epi, const uvit_tuning* tune, uint32_t* tile_counters, uvit_stream stream);
/* C[N,K] (f32) = Y[M,N]^T . X[M,K]: weight gradients; M must be a multiple of 64.  tune: NULL = defaults */
int uvit_op_gemm_tn(const void* Y_bf16, const void* X_bf16, int M, int N, int K, int ldy, int ldx, float* C, int ldc,
                    const uvit_tuning* tune, uvit_stream stream);
/* All weight gradients of one layer's Linears in one launch (256x256 tiles, token-chunked, fp32 atomics into C).
 * Every problem: M % 64 == 0 and M >= 512, N % 256 == 0, K % 256 == 0; C (and the bias outputs) must hold zeros or a
 * value to add to.  bias (nullable) receives the column sums of Y[:, 0:bias_end) -- the Linear's bias gradient,
 * modeling_finetune.py:149-151 for the qkv split -- and bias2 (nullable) those of Y[:, bias2_begin:N); both bounds
 * are multiples of 256.  Returns UVIT_ERR_SHAPE (nothing launched) when a problem does not qualify. */
typedef struct uvit_wgrad_problem {
    const void* Y_bf16; const void* X_bf16; float* C; float* bias; float* bias2;
    int32_t bias_end, bias2_begin, M, N, K, ldy, ldx, ldc;
} uvit_wgrad_problem;
int uvit_op_wgrad_group(const uvit_wgrad_problem* problems, int count, const uvit_tuning* tune, uvit_stream stream);
/* Attention core, modeling_finetune.py:152-185. qkv (B,N,3,H,64) bf16; biasP (H,NP,NP) f32 or NULL in the kernels'
 * private layout built by uvit_op_relpos_gather: bias * log2(e), -1e30 in padded key columns; lse is in log2 units */
int uvit_op_attn_fwd(const void* qkv, const float* biasP, void* out, float* lse, int B, int H, int N, int NP,
                     float scale, float p_drop, uint32_t seed, uint32_t layer, uvit_stream stream);
/* Backward of the same core (autograd of modeling_finetune.py:152-185), fused since round 3: dQ, dK, dV from ONE recomputation of
 * the probabilities; delta (B,H,N) = rowsum(dO o O) is an output; the score gradients leave the kernel once, as bf16, into
 * ds_workspace (uvit_op_attn_bwd_ws_bytes(B, H, N) bytes; only needed with dbias_slab) and a second kernel sums them over the
 * batch into dbias_slab = ONE (H, NP, NP) slab laid out [h][key][q] (accumulate_slab: add; NULL: no bias gradient). */
int64_t uvit_op_attn_bwd_ws_bytes(int B, int H, int N);
int uvit_op_attn_bwd(const void* qkv, const void* o_fwd, const void* d_o, const float* biasP, const float* lse,
                           float* delta, void* dqkv, float* dbias_slab, int accumulate_slab, void* ds_workspace, int B, int H,
                           int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, uvit_stream stream);
/* Two-stream Wasserstein attention core (modeling_finetune_dist.py:129-162 + uncertainty_evaluations.py:276-294).
 * qkv_m: mean-stream (B,N,3,H,64) bf16; qkv_c: covariance stream, already ELU(.)+1.  The backward returns the
 * gradient of the PRE-ELU covariance QKV (ELU' folded in).  biasP / lse units as for uvit_op_attn_fwd. */
int uvit_op_attn2_fwd(const void* qkv_m, const void* qkv_c, const float* biasP, void* out_m, void* out_c, float* lse, int B, int H,
                      int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, uvit_stream stream);
/* Backward, fused since round 4 (one recomputation of P for all six gradients), same contract as uvit_op_attn_bwd: the score
 * gradients leave once, as bf16, into ds_workspace (uvit_op_attn2_bwd_ws_bytes(B, H, N) bytes; only needed with dbias_slab) and a
 * second kernel sums them over the batch into dbias_slab = ONE (H, NP, NP) slab laid out [h][key][q]. */
int64_t uvit_op_attn2_bwd_ws_bytes(int B, int H, int N);
int uvit_op_attn2_bwd(const void* qkv_m, const void* qkv_c, const void* o_m, const void* o_c, const void* d_m, const void* d_c,
                      const float* biasP, const float* lse, float* delta, void* dqkv_m, void* dqkv_c, float* dbias_slab,
                      int accumulate_slab, void* ds_workspace, int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed,
                      uint32_t layer, uvit_stream stream);
int uvit_op_relpos_gather(const float* table, const int32_t* index, float* biasP, int H, int N, int NP, uvit_stream stream);
int uvit_op_relpos_scatter(const float* slab, int nslab, const int32_t* index, float* dtable, int H, int N, int NP,
                           uvit_stream stream);
/* nn.LayerNorm forward / backward (modeling_finetune.py:290-299) */
int uvit_op_ln_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* mean, float* rstd, int M, int C,
                   float eps, uvit_stream stream);
int uvit_op_ln_bwd(const void* dy_bf16, const float* x, const float* mean, const float* rstd, const float* w,
                   const float* dres, float* dx, float* dw, float* db, int M, int C, uvit_stream stream);
/* ModelEmaV2._update with the lambda of engine_for_cyclical.py:183 */
int uvit_op_ema(float* ema, const float* params, void* ema_bf16, int64_t n, float decay, uvit_stream stream);
int uvit_op_sumsq(const float* g, int64_t n, double* out, uvit_stream stream);
int uvit_op_adamw(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, int64_t n_decay, float lr,
                  float wd, float b1, float b2, float eps, int step, const double* sumsq, float max_norm,
                  float grad_scale, float* gnorm_out, uvit_stream stream);
/* F.smooth_l1_loss / F.mse_loss forward + gradient (engine_for_cyclical.py:147-150) */
int uvit_op_smooth_l1(const float* out, const float* target, const int32_t* count_dev, float beta, int l2,
                      float loss_scale, float* loss, void* dout_bf16, int Mmax, int C, uvit_stream stream);
/* WassersteinLoss forward + gradient (distloss.py:13-30, 73-79; engine_for_cyclical.py:152-161).  *loss += lam * loss_scale *
 * sum_r softplus(u_r) / max softplus(u); dout_m_bf16 (which already holds the SmoothL1 gradient) is ADDED to, dout_c_bf16 is
 * written.  scratch: 16 + Mmax floats.  Rows >= *count_dev are ignored (their dout_c rows are zeroed). */
int uvit_op_wasserstein_loss(const float* out_m, const float* out_c, const float* tgt_m, const float* tgt_c,
                             const int32_t* count_dev, float lam, float loss_scale, float* scratch, float* loss,
                             void* dout_m_bf16, void* dout_c_bf16, int Mmax, int C, uvit_stream stream);
/* target builder, engine_for_cyclical.py:92-122 */
int uvit_op_target_accum(const float* x, const int32_t* rowidx, const int32_t* count, float* acc, int first, int Mmax,
                         int C, float eps, uvit_stream stream);
/* variance term of the loss (engine_for_cyclical.py:130-139): z0_c = sqrt(var_r(out[r, c]) + 1e-6) over the *count_dev valid rows
 * (unbiased), std_loss0 = sum_c relu(margin - z0_c) / C.  *loss += w * loss_scale * std_loss0, *std_loss0_out = std_loss0, and
 * the gradient is ADDED to dout_bf16 (which already holds the regression gradient).  scratch: 2 * C + 16 floats. */
int uvit_op_variance_loss(const float* out, const int32_t* count_dev, float w, float margin, float loss_scale, float* scratch,
                          float* loss, float* std_loss0_out, void* dout_bf16, int Mmax, int C, uvit_stream stream);
int uvit_op_target_finalize(float* acc, const int32_t* count, int n_layers, int post_ln, int Mmax, int C, float eps,
                            uvit_stream stream);
int uvit_op_mask_compact(const int64_t* mask, int32_t* rowidx, int32_t* count, int B, int P, uvit_stream stream);
/* Synthetic batch built on the device (the loader contract of datasets.py:110-118 / engine_for_cyclical.py:45,58): images
 * (B, chans, S, S) f32 ~ N(0, 1); mask (B, patches) int64 with exactly n_mask ones per image.  Counter-based on (seed, it).
 * Either pointer may be NULL. */
int uvit_op_synth_batch(float* images, int64_t* mask, int B, int chans, int img_size, int patches, int n_mask, uint32_t seed,
                        uint32_t it, uvit_stream stream);
int uvit_op_im2col(const float* img, void* cols_bf16, int B, int Cin, int img_size, int patch, uvit_stream stream);
int uvit_op_droppath(float* scales, const float* rates_dev, int depth, int B, uint32_t seed, uint32_t step,
                     uvit_stream stream);
int uvit_op_cast_bf16(const float* src, void* dst_bf16, int64_t n, uvit_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* UVIT_H */
