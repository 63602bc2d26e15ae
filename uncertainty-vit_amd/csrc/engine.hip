// Engine: arena layout, workspace plan and the launch sequence of the data2vec ViT step
// (engine_for_cyclical.py:58-186 over modeling_cyclical.py:170-225) on one HIP stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/uvit.h"
#include "common.h"
#include "uvit_internal.h"

#define CHECK(x) do { int _rc = (x); if (_rc) return _rc; } while (0)
#define NREP 32
#define RP(off) (e->grep + ((off) - e->lo.n_decay))   // replica 0 address of a no-decay gradient
#define HIPCHECK(x) do { if ((x) != hipSuccess) return UVIT_ERR_LAUNCH; } while (0)
// every GEMM launch of the engine carries the engine's own tuning (no process-wide launcher state)
#define GEMM_NT(...) uvit_gemm_nt_launch(__VA_ARGS__, &e->tune)
#define GEMM_TN(...) uvit_gemm_tn_launch(__VA_ARGS__, &e->tune)
#define GEMM_TN_GROUP(...) uvit_gemm_tn_group_launch(__VA_ARGS__, &e->tune)

// ------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------
struct LayerOff { size_t n1w, n1b, qkvw, qb, vb, projw, projb, g1, n2w, n2b, fc1w, fc1b, fc2w, fc2b, g2;
                  size_t cqkvw, cqb, cvb, cprojw, cprojb; };   // two-stream model only
struct Layout {
    size_t cls, mask_tok, pew, peb, relt, normw, normb, lmw, lmb, pos;
    size_t ccls, cmask_tok, cpew, cpeb, clmw, clmb;          // two-stream model only
    LayerOff L[UVIT_MAX_DEPTH];
    size_t n_total, n_decay, n_live;     // [0, n_decay) decay | [n_decay, n_live) no-decay | [n_live, n_total) frozen
    std::vector<uvit_layout_entry> entries;
};

static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

static int cfg_ok(const uvit_config* c) {
    if (!c) return UVIT_ERR_ARG;
    if (c->depth < 1 || c->depth > UVIT_MAX_DEPTH || c->num_heads < 1 || c->embed_dim != c->num_heads * 64) return UVIT_ERR_SHAPE;
    if (c->embed_dim % 64 || c->mlp_hidden % 64 || c->patch_size % 8 || c->img_size % c->patch_size) return UVIT_ERR_SHAPE;
    if ((c->in_chans * c->patch_size * c->patch_size) % 64) return UVIT_ERR_SHAPE;
    const int g = c->img_size / c->patch_size;
    if (g * g + 1 > 208 || c->batch < 1) return UVIT_ERR_SHAPE;
    if (c->use_abs_pos_emb && c->two_stream) return UVIT_ERR_SHAPE;      // the two-stream model has no position embedding (modeling_cyclical_dist.py:113-130)
    return UVIT_OK;
}

static void build_layout(const uvit_config* c, Layout& lo) {
    const int64_t C = c->embed_dim, Hd = c->mlp_hidden, g = c->img_size / c->patch_size;
    const int64_t Kpe = (int64_t)c->in_chans * c->patch_size * c->patch_size;
    size_t off = 0;
    auto add = [&](const std::string& name, std::vector<int64_t> shape, int decay) -> size_t {
        uvit_layout_entry e;
        memset(&e, 0, sizeof(e));
        snprintf(e.name, sizeof(e.name), "%s", name.c_str());
        e.offset = (int64_t)off; e.ndim = (int)shape.size(); e.decay = decay; e.numel = 1;
        for (size_t i = 0; i < shape.size(); ++i) { e.shape[i] = shape[i]; e.numel *= shape[i]; }
        lo.entries.push_back(e);
        const size_t at = off;
        off = align64(off + (size_t)e.numel);
        return at;
    };
    auto blk = [](int i, const char* s) { return "blocks." + std::to_string(i) + "." + s; };
    // ---- decay group: everything that is not 1-D, not *.bias, not in {pos_embed, cls_token} ----
    const bool two = c->two_stream != 0;
    lo.mask_tok = add("mask_token", {1, 1, C}, 1);
    if (two) lo.cmask_tok = add("cov_mask_token", {1, 1, C}, 1);
    // 'cov_cls_token' is not in the no_weight_decay() skip list {'pos_embed','cls_token'} and is 3-D: decay group
    if (two) lo.ccls = add("cov_cls_token", {1, 1, C}, 1);
    if (c->use_shared_rel_pos_bias) lo.relt = add("rel_pos_bias.relative_position_bias_table", {(2 * g - 1) * (2 * g - 1) + 3, c->num_heads}, 1);
    else lo.relt = (size_t)-1;
    lo.pew = add("patch_embed.proj.weight", {C, c->in_chans, c->patch_size, c->patch_size}, 1);
    if (two) lo.cpew = add("cov_patch_embed.proj.weight", {C, c->in_chans, c->patch_size, c->patch_size}, 1);
    (void)Kpe;
    for (int i = 0; i < c->depth; ++i) {
        lo.L[i].qkvw = add(blk(i, "attn.qkv.weight"), {3 * C, C}, 1);
        lo.L[i].projw = add(blk(i, "attn.proj.weight"), {C, C}, 1);
        if (two) lo.L[i].cprojw = add(blk(i, "attn.cov_proj.weight"), {C, C}, 1);
        lo.L[i].fc1w = add(blk(i, "mlp.fc1.weight"), {Hd, C}, 1);
        lo.L[i].fc2w = add(blk(i, "mlp.fc2.weight"), {C, Hd}, 1);
    }
    lo.lmw = add("lm_head.weight", {C, C}, 1);
    if (two) lo.clmw = add("cov_lm_head.weight", {C, C}, 1);
    lo.n_decay = off;
    // ---- no-decay group ----
    lo.cls = add("cls_token", {1, 1, C}, 0);
    // 'pos_embed' (1, N, C) is 3-D but sits in the no_weight_decay() skip list (modeling_cyclical.py:163-165)
    lo.pos = c->use_abs_pos_emb ? add("pos_embed", {1, g * g + 1, C}, 0) : (size_t)-1;
    lo.peb = add("patch_embed.proj.bias", {C}, 0);
    if (two) lo.cpeb = add("cov_patch_embed.proj.bias", {C}, 0);
    for (int i = 0; i < c->depth; ++i) {
        lo.L[i].g1 = add(blk(i, "gamma_1"), {C}, 0);
        lo.L[i].g2 = add(blk(i, "gamma_2"), {C}, 0);
        lo.L[i].n1w = add(blk(i, "norm1.weight"), {C}, 0);
        lo.L[i].n1b = add(blk(i, "norm1.bias"), {C}, 0);
        lo.L[i].qb = add(blk(i, "attn.q_bias"), {C}, 0);
        lo.L[i].vb = add(blk(i, "attn.v_bias"), {C}, 0);
        if (two) { lo.L[i].cqb = add(blk(i, "attn.cov_q_bias"), {C}, 0); lo.L[i].cvb = add(blk(i, "attn.cov_v_bias"), {C}, 0); }
        lo.L[i].projb = add(blk(i, "attn.proj.bias"), {C}, 0);
        if (two) lo.L[i].cprojb = add(blk(i, "attn.cov_proj.bias"), {C}, 0);
        lo.L[i].n2w = add(blk(i, "norm2.weight"), {C}, 0);
        lo.L[i].n2b = add(blk(i, "norm2.bias"), {C}, 0);
        lo.L[i].fc1b = add(blk(i, "mlp.fc1.bias"), {Hd}, 0);
        lo.L[i].fc2b = add(blk(i, "mlp.fc2.bias"), {C}, 0);
    }
    lo.normw = add("norm.weight", {C}, 0);
    lo.normb = add("norm.bias", {C}, 0);
    lo.lmb = add("lm_head.bias", {C}, 0);
    if (two) lo.clmb = add("cov_lm_head.bias", {C}, 0);
    lo.n_live = off;
    // attn.cov_qkv.weight is never used by the reference's forward (modeling_finetune_dist.py:127 reuses qkv.weight):
    // its .grad stays None, so torch's AdamW never touches it (no weight decay either).  It lives at the END of the
    // no-decay region with a zero gradient, which leaves it exactly constant; flag 2 = "frozen".  Being last keeps its
    // 21 M floats out of the replicated column-sum accumulators (which span [n_decay, n_live) only).
    if (two)
        for (int i = 0; i < c->depth; ++i) lo.L[i].cqkvw = add(blk(i, "attn.cov_qkv.weight"), {3 * C, C}, 2);
    lo.n_total = off;
}

// ------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------
// Activation buffers are STACKED over the S streams of the model (S = 1 base model, S = 2 two-stream
// "stochastic" model: mean rows [0, M), covariance rows [Mpad, Mpad + M)).  Ops whose weights are shared by the
// streams (LayerNorms, fc1, fc2 weight gradient, qkv weight gradient, ...) run ONCE over the stacked rows.
struct LayerActs {
    bf16 *ln1, *qkv, *attn, *projout, *ln2, *h, *a, *mlpout;
    float *mean1, *rstd1, *mean2, *rstd2, *lse;
};

struct uvit_engine {
    uvit_config cfg;
    uvit_buffers buf;
    Layout lo;
    int B, P, N, NP, C, Hd, H, Kpe, M, Mpad, BP, BPpad, chunk, nchunk;
    int S;                 // streams: 1 or 2
    GemmTune tune;         // launch tuning of this engine's GEMMs (uvit_engine_set_tuning)
    int cur_B;             // batch of the last forward
    // workspace
    bf16* cols;
    float* X[UVIT_MAX_DEPTH + 1];
    float* XM[UVIT_MAX_DEPTH];
    LayerActs acts[UVIT_MAX_DEPTH];
    LayerActs tacts;       // teacher scratch (nothing saved)
    float *tX[2], *tXM;
    int *rowidx, *count;
    bf16 *normed[2], *dout[2], *dnormed[2], *dpatch[2];
    float *meanF[2], *rstdF[2], *outputs[2], *targets[2];
    float *biasP_s, *biasP_t, *slabs, *delta;
    void* ds_ws[2] = {nullptr, nullptr};   // fused attention backward: dS of one layer (bf16), by layer parity: the batch reduction runs on the second stream
    float *dXa, *dXb;
    bf16 *dY1[2], *dY2[2], *dH[2], *dLN, *dAttn, *dqkv[2];   // [layer parity]: read by the wgrad stream while the next layer runs
    float *dp_scales, *dp_rates;
    // drop-path sample lists (training step): per (layer, draw) list lb = 2 S l + k, k as in dp_ptr (base: attn, mlp; two-stream: mean attn,
    // mean mlp, cov attn, cov mlp -- only the two MLP lists are used there: the two-stream attention needs both streams of a sample)
    int *dpl_pos = nullptr, *dpl_bmap = nullptr, *dpl_rows = nullptr, *dpl_cnt = nullptr;
    size_t dpl_stride = 0;                 // ints per rows list
    int dpl_K[128] = {};                   // kept samples per list (2 S lists per layer), from the host's evaluation of the drop-path hash
    bool dpl_enable = true;                // uvit_engine_set_drop_path_rows
    bool dpl_on = false;                   // the current step runs with lists
    std::vector<float> dp_rates_host;
    float *loss, *gnorm; double* sumsq;
    float* wl_scratch;     // Wasserstein loss: scalars + per-row distances
    float *dense_v, *dense_acc; int *idrows, *idcount;   // dense target builder (batch / instance-norm target variants)
    float* var_scratch;    // variance term: column sums / centred squares (2 C + 16 floats)
    int* poisoned;         // sticky: set by the first step whose loss / gradient norm is not finite; AdamW and EMA then skip every step
    TransposeDesc* tdesc; int n_tdesc, n_ttiles;
    int64_t* mask_copy;
    unsigned* tile_cnt = nullptr;   // 2 x 16 counters (caller's stream, second stream): dynamic tile assignment of the persistent GEMMs
    float* grep;           // [NREP][no-decay region] replicated column-sum accumulators
    size_t n_nd;           // floats in the live (not frozen) part of the no-decay region
    bool slab_started;
    bool last_dropout; uint32_t last_seed, last_it;
    int ls_prefused = -1;   // layer whose MLP-branch LayerScale backward was already done by layer+1's fused LayerNorm backward
    int compact_R = 0;      // > 0: this step runs the last block's MLP on the masked rows only, in R (multiple of 64) compact rows (uvit_step_params.n_rows_hint)
    // second stream: teacher forward beside student forward; wgrad GEMMs beside the dgrad chain
    bool dual = true;
    hipStream_t aux = nullptr, aux_eq = nullptr, aux_lo = nullptr;     // aux = the one in use (uvit_engine_set_streams)
    hipEvent_t ev_fork = nullptr, ev_teacher = nullptr, ev_x[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_wdone[UVIT_MAX_DEPTH] = {};
    hipEvent_t ev_ds = nullptr;             // dS of the current layer is complete (main stream -> second stream)
    // optional HIP-event bracketing of the dominant kernel (fc1 GEMM, EPI_GELU) for bench.py's roofline
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;   // pairs
    std::vector<int> prof_kind;        // per pair: UVIT_PROF_* (which Linear of the block the bracket holds)
    std::vector<int> prof_rows;        // per pair: rows of the bracketed launch (compact launches run fewer than B x tokens)
    size_t prof_used = 0;
    // stacked-row helpers
    size_t rows_all() const { return (size_t)(S - 1) * Mpad + M; }      // rows a stacked row-wise op covers
    size_t rows_alloc() const { return (size_t)S * Mpad; }
};

struct Bump {
    char* base; size_t off, cap;
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

static size_t roundup(size_t x, size_t m) { return (x + m - 1) / m * m; }
static std::vector<float> dp_rates_linspace(int depth, float rate);

static void plan_workspace(uvit_engine* e, Bump& b) {
    const uvit_config& c = e->cfg;
    const size_t Mp = e->rows_alloc(), C = e->C, Hd = e->Hd, BPp = e->BPpad;
    auto acts = [&](LayerActs& a) {
        a.ln1 = b.take<bf16>(Mp * C); a.qkv = b.take<bf16>(Mp * 3 * C); a.attn = b.take<bf16>(Mp * C);
        a.projout = b.take<bf16>(Mp * C); a.ln2 = b.take<bf16>(Mp * C); a.h = b.take<bf16>(Mp * Hd);
        a.a = b.take<bf16>(Mp * Hd); a.mlpout = b.take<bf16>(Mp * C);
        a.mean1 = b.take<float>(Mp); a.rstd1 = b.take<float>(Mp); a.mean2 = b.take<float>(Mp); a.rstd2 = b.take<float>(Mp);
        a.lse = b.take<float>((size_t)e->B * e->H * e->N);
    };
    e->cols = b.take<bf16>(BPp * e->Kpe);
    for (int i = 0; i <= c.depth; ++i) e->X[i] = b.take<float>(Mp * C);
    for (int i = 0; i < c.depth; ++i) { e->XM[i] = b.take<float>(Mp * C); acts(e->acts[i]); }
    acts(e->tacts);
    e->tX[0] = b.take<float>(Mp * C); e->tX[1] = b.take<float>(Mp * C); e->tXM = b.take<float>(Mp * C);
    e->rowidx = b.take<int>(e->BP + 64); e->count = b.take<int>(64);
    for (int st = 0; st < e->S; ++st) {
        e->normed[st] = b.take<bf16>(BPp * C); e->dout[st] = b.take<bf16>(BPp * C); e->dnormed[st] = b.take<bf16>(BPp * C);
        e->dpatch[st] = b.take<bf16>(BPp * C);
        e->meanF[st] = b.take<float>(BPp); e->rstdF[st] = b.take<float>(BPp);
        e->outputs[st] = b.take<float>(BPp * C); e->targets[st] = b.take<float>(BPp * C);
    }
    const size_t bias_n = (size_t)e->H * e->NP * e->NP;
    e->biasP_s = b.take<float>(bias_n); e->biasP_t = b.take<float>(bias_n);
    e->slabs = b.take<float>(bias_n);       // ONE bias-gradient slab [h][key][q] (both model families since round 4)
    e->delta = b.take<float>((size_t)e->B * e->H * e->N);
    if (e->cfg.use_shared_rel_pos_bias)
        for (int k = 0; k < 2; ++k)
            e->ds_ws[k] = b.take<char>(e->S == 1 ? uvit_attn_bwd_fused_ws_bytes(e->B, e->H, e->N) : uvit_attn2_bwd_ws_bytes(e->B, e->H, e->N));
    e->dXa = b.take<float>(Mp * C); e->dXb = b.take<float>(Mp * C);
    for (int k = 0; k < 2; ++k) {
        e->dY1[k] = b.take<bf16>(Mp * C); e->dY2[k] = b.take<bf16>(Mp * C); e->dH[k] = b.take<bf16>(Mp * Hd);
        e->dqkv[k] = b.take<bf16>(Mp * 3 * C);
    }
    e->dLN = b.take<bf16>(Mp * C); e->dAttn = b.take<bf16>(Mp * C);
    e->dp_scales = b.take<float>((size_t)c.depth * 2 * e->S * e->B); e->dp_rates = b.take<float>(c.depth);
    if ((size_t)c.depth * 2 * e->S <= 128) {
        const size_t nl = (size_t)c.depth * 2 * e->S;
        e->dpl_stride = roundup((size_t)e->B * e->N, 64);
        e->dpl_pos = b.take<int>(nl * e->B); e->dpl_bmap = b.take<int>(nl * e->B);
        e->dpl_rows = b.take<int>(nl * e->dpl_stride); e->dpl_cnt = b.take<int>(2 * nl + 64);    // counts, then the host-mismatch flags
    }
    e->loss = b.take<float>(64); e->gnorm = e->loss + 1; e->sumsq = (double*)(e->loss + 2);
    e->wl_scratch = b.take<float>(16 + BPp);
    e->poisoned = b.take<int>(64);
    e->var_scratch = b.take<float>(2 * C + 16);
    e->dense_v = b.take<float>(BPp * C); e->dense_acc = b.take<float>(BPp * C);
    e->idrows = b.take<int>(e->BP + 64); e->idcount = b.take<int>(64);
    e->tdesc = b.take<TransposeDesc>(7 * c.depth + 4);
    e->mask_copy = b.take<int64_t>(e->BP + 64);
    e->grep = b.take<float>((size_t)NREP * e->n_nd);
    e->tile_cnt = b.take<unsigned>(64);          // zero from the workspace memset; every launch leaves its counters at zero
}

static void fill_dims(uvit_engine* e) {
    const uvit_config& c = e->cfg;
    const int g = c.img_size / c.patch_size;
    e->B = c.batch; e->P = g * g; e->N = e->P + 1; e->NP = 208; e->C = c.embed_dim; e->Hd = c.mlp_hidden;
    e->H = c.num_heads; e->Kpe = c.in_chans * c.patch_size * c.patch_size;
    e->S = c.two_stream ? 2 : 1;
    e->M = e->B * e->N; e->Mpad = (int)roundup(e->M, 128); e->BP = e->B * e->P; e->BPpad = (int)roundup(e->BP, 128);
    if (c.bias_chunk > 0) {
        e->chunk = c.bias_chunk;
    } else {
        // the dQ kernel runs (heads x chunks x 2 halves) workgroups, two resident per CU: take the most chunks that still
        // fit one round (ViT-B bs=128 on 256 CUs: 19 chunks of 7 samples = 456 workgroups instead of 16 x 8 = 384)
        int ncu = 256, dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ncu = prop.multiProcessorCount;
        const int nhalf = (e->N + 15) / 16 > 7 ? 2 : 1;
        const int resident = 1;                       // both dQ kernels hold four 28-KiB images (double-buffered K, V): one workgroup per CU
        int max_chunks = (resident * ncu) / (e->H * nhalf);
        if (max_chunks < 1) max_chunks = 1;
        e->chunk = (e->B + max_chunks - 1) / max_chunks;
        if (e->chunk < 1) e->chunk = 1;
    }
    e->nchunk = (e->B + e->chunk - 1) / e->chunk;
    e->cur_B = e->B;
    Layout tmp_lo; build_layout(&e->cfg, tmp_lo);
    e->n_nd = tmp_lo.n_live - tmp_lo.n_decay;      // span of the replicated accumulators: live no-decay tensors only
}


extern "C" void uvit_engine_destroy(uvit_engine* e);
extern "C" int uvit_version(void) { return UVIT_VERSION; }
#ifndef UVIT_SRC_HASH
#define UVIT_SRC_HASH "unknown"
#endif
extern "C" const char* uvit_source_hash(void) { return UVIT_SRC_HASH; }

extern "C" int uvit_layout_count(const uvit_config* cfg) {
    if (cfg_ok(cfg)) return UVIT_ERR_SHAPE;
    Layout lo; build_layout(cfg, lo);
    return (int)lo.entries.size();
}
extern "C" int uvit_layout_get(const uvit_config* cfg, int index, uvit_layout_entry* out) {
    if (cfg_ok(cfg) || !out) return UVIT_ERR_SHAPE;
    Layout lo; build_layout(cfg, lo);
    if (index < 0 || index >= (int)lo.entries.size()) return UVIT_ERR_ARG;
    *out = lo.entries[index];
    return UVIT_OK;
}
extern "C" int64_t uvit_arena_numel(const uvit_config* cfg, int64_t* n_decay_out) {
    if (cfg_ok(cfg)) return UVIT_ERR_SHAPE;
    Layout lo; build_layout(cfg, lo);
    if (n_decay_out) *n_decay_out = (int64_t)lo.n_decay;
    return (int64_t)lo.n_total;
}
extern "C" int64_t uvit_workspace_bytes(const uvit_config* cfg) {
    if (cfg_ok(cfg)) return UVIT_ERR_SHAPE;
    uvit_engine tmp;
    tmp.cfg = *cfg;
    fill_dims(&tmp);
    Bump b{nullptr, 0, 0};
    plan_workspace(&tmp, b);
    return (int64_t)roundup(b.off, 4096) + 4096;
}

extern "C" uvit_engine* uvit_engine_create(const uvit_config* cfg, const uvit_buffers* bufs, uvit_stream stream, int* err_out) {
    auto fail = [&](int rc) -> uvit_engine* { if (err_out) *err_out = rc; return nullptr; };
    if (cfg_ok(cfg) || !bufs) return fail(UVIT_ERR_SHAPE);
    if (!bufs->params || !bufs->grads || !bufs->adam_m || !bufs->adam_v || !bufs->ema || !bufs->params_bf16 ||
        !bufs->params_bf16_t || !bufs->ema_bf16 || !bufs->workspace) return fail(UVIT_ERR_ARG);
    if (cfg->use_shared_rel_pos_bias && !bufs->rel_index) return fail(UVIT_ERR_ARG);
    if (bufs->workspace_bytes < uvit_workspace_bytes(cfg)) return fail(UVIT_ERR_WORKSPACE);
    uvit_engine* e = new uvit_engine();
    e->cfg = *cfg; e->buf = *bufs;
    fill_dims(e);
    build_layout(cfg, e->lo);
    Bump b{(char*)bufs->workspace, 0, (size_t)bufs->workspace_bytes};
    plan_workspace(e, b);
    hipStream_t s = (hipStream_t)stream;
    // pad rows of every activation buffer must be (and stay) zero: TN GEMMs reduce over them
    if (hipMemsetAsync(bufs->workspace, 0, b.off, s) != hipSuccess) { delete e; return fail(UVIT_ERR_LAUNCH); }
    // drop-path rates linspace(0, rate, depth)  (modeling_cyclical.py:94-96)
    std::vector<float> rates = dp_rates_linspace(cfg->depth, cfg->drop_path_rate);
    e->dp_rates_host = rates;
    if (const char* v = getenv("UVIT_DP_ROWS")) e->dpl_enable = v[0] != '0';     // A/B switch (uvit_engine_set_drop_path_rows)
    // transposed-copy descriptors
    std::vector<TransposeDesc> td;
    int tiles = 0;
    const char* src = (const char*)bufs->params_bf16; char* dst = (char*)bufs->params_bf16_t;
    auto addT = [&](size_t off, int rows, int cols) {
        TransposeDesc d; d.src = src + off * 2; d.dst = dst + off * 2; d.rows = rows; d.cols = cols; d.tile0 = tiles; d.pad = 0;
        tiles += ((rows + 63) / 64) * ((cols + 63) / 64);
        td.push_back(d);
    };
    for (int i = 0; i < cfg->depth; ++i) {
        addT(e->lo.L[i].qkvw, 3 * e->C, e->C); addT(e->lo.L[i].projw, e->C, e->C);
        addT(e->lo.L[i].fc1w, e->Hd, e->C); addT(e->lo.L[i].fc2w, e->C, e->Hd);
        if (e->S == 2) addT(e->lo.L[i].cprojw, e->C, e->C);
    }
    addT(e->lo.lmw, e->C, e->C);
    if (e->S == 2) addT(e->lo.clmw, e->C, e->C);
    e->n_tdesc = (int)td.size(); e->n_ttiles = tiles;
    {
        std::vector<int> idr(e->BP);
        for (int i = 0; i < e->BP; ++i) idr[i] = i;
        const int cnt = e->BP;
        if (hipMemcpyAsync(e->idrows, idr.data(), idr.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(e->idcount, &cnt, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { delete e; return fail(UVIT_ERR_LAUNCH); }
    }
    if (hipMemcpyAsync(e->dp_rates, rates.data(), rates.size() * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(e->tdesc, td.data(), td.size() * sizeof(TransposeDesc), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { delete e; return fail(UVIT_ERR_LAUNCH); }
    e->slab_started = false; e->last_dropout = false; e->last_seed = 0; e->last_it = 0;
    {
        const char* env = getenv("UVIT_SINGLE_STREAM");
        e->dual = !(env && env[0] == '1');
        // Two second streams: one at the caller's priority, one BELOW it (round 4).  What the second stream carries -- teacher forward,
        // weight gradients, bias-gradient reductions -- is off the critical chain of the step (student forward -> loss -> dgrad chain), and
        // with the lower priority the dispatcher hands free CUs to the critical chain first: 25.41 -> 25.03 ms per step on all 256 CUs
        // (tools/ab.sh, 4 alternating pairs).  With CUs taken away it is the other way round (240 CUs: 28.51 -> 29.21 ms: the 216 wgrad
        // workgroups then queue behind the chain and finish late), so the host selects it for single-GPU runs only
        // (uvit_engine_set_streams(e, 2)); data-parallel runs, where RCCL's channel workgroups hold CUs during backward, keep mode 1.
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        bool ok = hipStreamCreateWithFlags(&e->aux_eq, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithPriority(&e->aux_lo, hipStreamNonBlocking, lo) == hipSuccess;
        e->aux = e->aux_eq;
        auto mk = [&](hipEvent_t* ev) { ok = ok && hipEventCreateWithFlags(ev, hipEventDisableTiming) == hipSuccess; };
        mk(&e->ev_fork); mk(&e->ev_teacher); mk(&e->ev_ds);
        for (int i = 0; i < 4; ++i) mk(&e->ev_x[i]);
        for (int i = 0; i < cfg->depth; ++i) mk(&e->ev_wdone[i]);
        if (!ok) { uvit_engine_destroy(e); return fail(UVIT_ERR_LAUNCH); }
    }
    if (err_out) *err_out = UVIT_OK;
    return e;
}

extern "C" void uvit_engine_destroy(uvit_engine* e) {
    if (!e) return;
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    if (e->aux_eq) { (void)hipStreamSynchronize(e->aux_eq); (void)hipStreamDestroy(e->aux_eq); }
    if (e->aux_lo) { (void)hipStreamSynchronize(e->aux_lo); (void)hipStreamDestroy(e->aux_lo); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_teacher) (void)hipEventDestroy(e->ev_teacher);
    if (e->ev_ds) (void)hipEventDestroy(e->ev_ds);
    for (int i = 0; i < 4; ++i) if (e->ev_x[i]) (void)hipEventDestroy(e->ev_x[i]);
    for (int i = 0; i < UVIT_MAX_DEPTH; ++i) if (e->ev_wdone[i]) (void)hipEventDestroy(e->ev_wdone[i]);
    delete e;
}

static int tune_from_abi(const uvit_tuning* t, GemmTune& g) {
    if (!t) { g = GemmTune(); return UVIT_OK; }
    if ((t->nt_variant != 0 && t->nt_variant != 1 && t->nt_variant != 3 && t->nt_variant != 5 && t->nt_variant != 6 && t->nt_variant != 7) ||
        (t->tn_variant != 0 && t->tn_variant != 1 && t->tn_variant != 3) || t->tn_split_target < 0 || t->wgrad_group_chunks < 0 || t->nt_group < 0 || t->nt_group > 64 || (t->nt_persist != 0 && t->nt_persist != 1))
        return UVIT_ERR_ARG;
    g.nt_variant = t->nt_variant; g.tn_variant = t->tn_variant;
    g.tn_target = t->tn_split_target > 0 ? t->tn_split_target : 512; g.group_chunks = t->wgrad_group_chunks; g.nt_group = t->nt_group; g.nt_persist = t->nt_persist;
    return UVIT_OK;
}

extern "C" void uvit_tuning_default(uvit_tuning* out) {
    if (!out) return;
    const GemmTune d;
    out->nt_variant = d.nt_variant; out->tn_variant = d.tn_variant; out->tn_split_target = d.tn_target; out->wgrad_group_chunks = d.group_chunks; out->nt_group = d.nt_group; out->nt_persist = d.nt_persist;
}

extern "C" int uvit_engine_set_tuning(uvit_engine* e, const uvit_tuning* t) {
    if (!e) return UVIT_ERR_ARG;
    return tune_from_abi(t, e->tune);
}

extern "C" int uvit_engine_set_streams(uvit_engine* e, int dual) {
    if (!e || dual < 0 || dual > 2) return UVIT_ERR_ARG;
    // switching streams between steps: whatever the old second stream still holds must be ordered before the new one's work
    hipStream_t next = dual == 2 ? e->aux_lo : e->aux_eq;
    if (next != e->aux && e->aux) HIPCHECK(hipStreamSynchronize(e->aux));
    e->aux = next;
    e->dual = dual != 0;
    return UVIT_OK;
}

extern "C" int uvit_engine_set_drop_path_rows(uvit_engine* e, int on) {
    if (!e) return UVIT_ERR_ARG;
    e->dpl_enable = on != 0;
    return UVIT_OK;
}

extern "C" int uvit_engine_profile(uvit_engine* e, int enable, int max_launches) {
    if (!e) return UVIT_ERR_ARG;
    if (enable && e->prof_ev.empty()) {
        e->prof_ev.resize((size_t)(max_launches > 0 ? max_launches : 4096) * 2);
        e->prof_kind.assign(e->prof_ev.size() / 2, -1);
        e->prof_rows.assign(e->prof_ev.size() / 2, 0);
        for (auto& ev : e->prof_ev) if (hipEventCreate(&ev) != hipSuccess) return UVIT_ERR_LAUNCH;
    }
    e->prof_on = enable != 0;
    if (enable) e->prof_used = 0;
    return UVIT_OK;
}

// kinds of bracketed launches (full-size forward Linears of a block; the bench line's roofline block names them)
enum { UVIT_PROF_FC1_T = 0, UVIT_PROF_FC1_S = 1, UVIT_PROF_PROJ = 2, UVIT_PROF_FC2 = 3, UVIT_PROF_KINDS = 4 };

// kind < 0: the two fc1 kinds together (the round-1 contract of uvit_engine_profile_read)
extern "C" int uvit_engine_profile_read_kind(uvit_engine* e, int kind, double* total_ms, int* launches, double* flops_per_launch,
                                             double* bytes_per_launch) {
    if (!e || !total_ms || !launches || kind >= UVIT_PROF_KINDS) return UVIT_ERR_ARG;
    double t = 0.0, rows = 0.0; int n = 0;
    for (size_t i = 0; i + 1 < e->prof_used; i += 2) {
        const int k = e->prof_kind[i / 2];
        if (kind < 0 ? (k != UVIT_PROF_FC1_T && k != UVIT_PROF_FC1_S) : k != kind) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]) != hipSuccess) return UVIT_ERR_LAUNCH;
        t += ms; ++n; rows += e->prof_rows[i / 2];
    }
    *total_ms = t; *launches = n;
    // algorithmic work of ONE launch, at the MEAN row count of the bracketed launches (rows of one stream for proj / fc2, the stacked rows
    // for fc1; compact launches -- drop-path sample lists, the masked-row last block -- run fewer rows than B x tokens)
    const double M1 = n ? rows / n : (double)e->cur_B * e->N, Mall = M1, C = e->C, Hd = e->Hd;
    double fl = 0.0, by = 0.0;
    if (kind < 0 || kind == UVIT_PROF_FC1_T || kind == UVIT_PROF_FC1_S) {
        fl = 2.0 * Mall * Hd * C;
        by = 2.0 * (Mall * C + Hd * C + Mall * Hd * (kind == UVIT_PROF_FC1_T ? 1.0 : kind == UVIT_PROF_FC1_S ? 2.0 : 1.5));
    } else if (kind == UVIT_PROF_PROJ) {
        fl = 2.0 * M1 * C * C;            // bf16 A + W, fp32 residual in and stream out (+ the student's saved bf16 branch output: not counted)
        by = 2.0 * (M1 * C + C * C) + 8.0 * M1 * C;
    } else {
        fl = 2.0 * M1 * C * Hd;
        by = 2.0 * (M1 * Hd + C * Hd) + 8.0 * M1 * C;
    }
    if (flops_per_launch) *flops_per_launch = fl;
    if (bytes_per_launch) *bytes_per_launch = by;
    return UVIT_OK;
}
extern "C" int uvit_engine_profile_read(uvit_engine* e, double* total_ms, int* launches, double* flops_per_launch) {
    return uvit_engine_profile_read_kind(e, -1, total_ms, launches, flops_per_launch, nullptr);
}

extern "C" int uvit_engine_sync_shadows(uvit_engine* e, int which, uvit_stream stream) {
    if (!e) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (which & 1) {
        CHECK(uvit_cast_bf16_launch(e->buf.params, e->buf.params_bf16, e->lo.n_total, s));
        CHECK(uvit_transpose_batch_launch(e->tdesc, e->n_tdesc, e->n_ttiles, s));
    }
    if (which & 2) CHECK(uvit_cast_bf16_launch(e->buf.ema, e->buf.ema_bf16, e->lo.n_total, s));
    return UVIT_OK;
}

// ---- forward ----
struct Weights { const float* f; const bf16* b; };

// per-stream parameter offsets (stream 1 = covariance stream of the two-stream model)
static size_t off_projw(const LayerOff& o, int st) { return st ? o.cprojw : o.projw; }
static size_t off_projb(const LayerOff& o, int st) { return st ? o.cprojb : o.projb; }
static size_t off_qb(const LayerOff& o, int st) { return st ? o.cqb : o.qb; }
static size_t off_vb(const LayerOff& o, int st) { return st ? o.cvb : o.vb; }

// drop-path multipliers: 2 draws per block (base) or 4 (two-stream: mean attn, mean mlp, cov attn, cov mlp;
// modeling_finetune_dist.py:51-55)
static const float* dp_ptr(uvit_engine* e, bool on, int l, int st, int branch, int Bc) {
    if (!on) return nullptr;
    const int nbr = 2 * e->S, k = e->S == 2 ? 2 * st + branch : branch;
    return e->dp_scales + (size_t)(nbr * l + k) * Bc;
}

// Drop-path sample lists (base model, training step with drop_path > 0; round 4).  A branch that dropped a sample adds exactly 0 for it in the
// forward and sends exactly 0 back (timm drop_path: x / keep * mask, modeling_finetune.py:51-62), so each branch of the student runs on its
// KEPT samples only: the LayerNorm in front gathers them into compact rows (and copies the dropped samples' rows to the branch's output
// stream), the Linears / the attention core run K * tokens rows, the residual epilogue scatters through the row list, and backward mirrors it.
// The host evaluates the same integer hash as droppath_kernel to size the launches; the lists themselves are built on the device.
struct DpList { const int *pos, *bmap, *rows, *cnt; int K; };
static void dp_list_get(const uvit_engine* e, int l, int st, int branch, DpList& d) {
    const int lb = 2 * e->S * l + (e->S == 2 ? 2 * st + branch : branch);
    d.pos = e->dpl_pos + (size_t)lb * e->B; d.bmap = e->dpl_bmap + (size_t)lb * e->B;
    d.rows = e->dpl_rows + (size_t)lb * e->dpl_stride; d.cnt = e->dpl_cnt + lb; d.K = e->dpl_K[lb];
}
// base model: the list of (layer, branch) when it drops somebody (nobody dropped: the dense launches; everybody dropped: dense too, 0 * branch)
static bool dp_list(const uvit_engine* e, int l, int branch, DpList& d) {
    if (!e->dpl_on || e->S != 1) return false;
    dp_list_get(e, l, 0, branch, d);
    return d.K > 0 && d.K < e->B;
}
// two-stream model: the MLP branch of a layer runs compact when either stream dropped somebody and neither dropped everybody; the kept rows of
// the mean stream, then those of the covariance stream, are stacked without a gap (fc1 / fc2 share their weights between the streams)
static bool dp_list2(const uvit_engine* e, int l, DpList (&d)[2]) {
    if (!e->dpl_on || e->S != 2) return false;
    dp_list_get(e, l, 0, 1, d[0]); dp_list_get(e, l, 1, 1, d[1]);
    return d[0].K > 0 && d[1].K > 0 && (d[0].K < e->B || d[1].K < e->B);
}
// The host's evaluation of droppath_kernel (elementwise.hip): kept samples per (layer, draw), draws = 2 (base: attn, mlp) or 4 (two-stream).
// Pure host code -- no device call -- so the CPU test suite pins it against the oracle's drop_path_scales.
static void dp_kept_counts(const float* rates, int depth, int nbr, int B, uint32_t seed, uint32_t it, int* out) {
    for (int l = 0; l < depth; ++l)
        for (int br = 0; br < nbr; ++br) {
            const float r = rates[l];
            int K = B;
            if (r > 0.f) {
                const uint32_t key = uvit_hash32(seed ^ ((it * (uint32_t)nbr * depth + (uint32_t)nbr * l + br + 1u) * 0x9E3779B9u));
                const uint32_t thr = uvit_drop_threshold(r);
                K = 0;
                for (int b = 0; b < B; ++b) K += uvit_hash32((uint32_t)b ^ key) >= thr ? 1 : 0;
            }
            out[nbr * l + br] = K;
        }
}
static std::vector<float> dp_rates_linspace(int depth, float rate) {     // drop-path rates linspace(0, rate, depth)  (modeling_cyclical.py:94-96)
    std::vector<float> rates(depth);
    for (int i = 0; i < depth; ++i) rates[i] = depth > 1 ? (float)((double)rate * i / (depth - 1)) : 0.f;
    return rates;
}
extern "C" int uvit_drop_path_kept_counts(int depth, float drop_path_rate, int draws_per_block, int B, uint32_t seed, uint32_t it,
                                          int32_t* out) {
    if (depth < 1 || depth > UVIT_MAX_DEPTH || (draws_per_block != 2 && draws_per_block != 4) || B < 1 || !out) return UVIT_ERR_ARG;
    const std::vector<float> rates = dp_rates_linspace(depth, drop_path_rate);
    dp_kept_counts(rates.data(), depth, draws_per_block, B, seed, it, out);
    return UVIT_OK;
}
static int dp_lists_begin(uvit_engine* e, uint32_t seed, uint32_t it, int Bc, hipStream_t s) {
    const int depth = e->cfg.depth, nbr = 2 * e->S;
    e->dpl_on = false;
    if (!e->dpl_pos) return UVIT_OK;
    dp_kept_counts(e->dp_rates_host.data(), depth, nbr, Bc, seed, it, e->dpl_K);
    bool any = false;
    for (int i = 0; i < nbr * depth; ++i) {
        const int K = e->dpl_K[i], br = i % nbr;
        any = any || (K > 0 && K < Bc && (e->S == 1 || (br & 1)));
    }
    e->dpl_on = any;
    if (!any) return UVIT_OK;
    return uvit_droppath_lists_launch(e->dp_scales, e->dpl_pos, e->dpl_bmap, e->dpl_rows, e->dpl_cnt, nbr * depth, Bc, e->N,
                                      (int)e->dpl_stride, e->dpl_K, s);
}

// R > 0 (base model, student's last block of a training step): the MLP branch runs on the R compact rows of the masked-patch list only --
// LN2 gathers them, fc1 / fc2 are R-row GEMMs and the residual epilogue of fc2 reads x_mid and writes x_out / the saved branch output at the
// listed rows (the other rows of x_out are never read: the head and the final-norm backward go through the same list).
static int forward_layer(uvit_engine* e, const Weights& w, int l, const float* x_in, float* x_mid, float* x_out,
                         LayerActs& a, bool save, const float* biasP, bool dp_on, float pdrop, uint32_t seed, int Bc,
                         hipStream_t s, int R = 0, bool lists = false) {
    const LayerOff& o = e->lo.L[l];
    const int M = Bc * e->N, C = e->C, Hd = e->Hd, S = e->S;
    const size_t Mp = e->Mpad;                                   // row offset of stream 1
    const int Mall = (int)((size_t)(S - 1) * Mp + M);
    // counters of the persistent GEMMs' dynamic tile assignment: one block per stream (the teacher and student forwards run side by side)
    static const bool dyn_tiles = !(getenv("UVIT_DYN_TILES") && getenv("UVIT_DYN_TILES")[0] == '0');     // A/B switch
    unsigned* const tcnt = (e->tile_cnt && dyn_tiles) ? e->tile_cnt + ((e->dual && s == e->aux) ? 16 : 0) : nullptr;
    // drop-path sample lists of the two branches of the base model: Ma / Mm rows instead of M (the masked-row last block keeps its own row list
    // for the MLP and takes the sample list for the attention branch)
    DpList da{}, dm{};
    const bool la = lists && S == 1 && Bc == e->B && dp_list(e, l, 0, da);     // (also in the masked-row last block: only its MLP keeps the row list)
    const bool lm = lists && R == 0 && S == 1 && Bc == e->B && dp_list(e, l, 1, dm);
    const int Ma = la ? da.K * e->N : M;
    if (la) CHECK(uvit_ln_fwd_keep_launch(x_in, da.pos, w.f + o.n1w, w.f + o.n1b, a.ln1, a.mean1, a.rstd1, x_mid, M, C, e->N, e->cfg.ln_eps, s));
    else CHECK(uvit_ln_fwd_launch(x_in, w.f + o.n1w, w.f + o.n1b, a.ln1, a.mean1, a.rstd1, Mall, C, e->cfg.ln_eps, s));
    for (int st = 0; st < S; ++st) {     // same qkv.weight for both streams (modeling_finetune_dist.py:121,127)
        GemmEpi q; q.out = a.qkv + st * Mp * 3 * C; q.bias = w.f + off_qb(o, st); q.bias2 = w.f + off_vb(o, st); q.ldo = 3 * C;
        q.tile_counter = tcnt;
        CHECK(GEMM_NT(st ? EPI_QKV_ELU : EPI_QKV, a.ln1 + st * Mp * C, w.b + o.qkvw, Ma, 3 * C, C, C, C, &q, s));
    }
    if (S == 1) {
        CHECK(uvit_attn_fwd_launch(a.qkv, biasP, a.attn, a.lse, la ? da.K : Bc, e->H, e->N, e->NP, 0.125f, pdrop, seed, (uint32_t)l, s,
                                   la ? da.bmap : nullptr));
    } else {
        CHECK(uvit_attn2_fwd_launch(a.qkv, a.qkv + Mp * 3 * C, biasP, a.attn, a.attn + Mp * C, a.lse, Bc, e->H, e->N, e->NP, 0.125f,
                                    pdrop, seed, (uint32_t)l, s));
    }
    // optional HIP-event brackets around the block's forward Linears, each with the rows of its launch (bench.py's roofline block)
    auto prof_begin = [&](int kind, int rows) -> bool {
        if (!(e->prof_on && e->prof_used + 2 <= e->prof_ev.size())) return false;
        e->prof_kind[e->prof_used / 2] = kind; e->prof_rows[e->prof_used / 2] = rows;
        (void)hipEventRecord(e->prof_ev[e->prof_used], s);
        return true;
    };
    auto prof_end = [&](bool on) { if (on) { (void)hipEventRecord(e->prof_ev[e->prof_used + 1], s); e->prof_used += 2; } };
    for (int st = 0; st < S; ++st) {
        GemmEpi p; p.out = x_mid + st * Mp * C; p.out2 = save ? a.projout + st * Mp * C : nullptr; p.bias = w.f + off_projb(o, st);
        p.gamma = w.f + o.g1; p.resid = x_in + st * Mp * C; p.rowscale = dp_ptr(e, dp_on, l, st, 0, Bc); p.ldo = C; p.tokens = e->N;
        if (la) { p.rowmap = da.rows; p.rowcount = da.cnt; }
        const bool pp = prof_begin(UVIT_PROF_PROJ, Ma);
        CHECK(GEMM_NT(EPI_RESID, a.attn + st * Mp * C, w.b + off_projw(o, st), Ma, C, C, C, C, &p, s));
        prof_end(pp);
    }
    DpList d2[2] = {};
    const bool lm2 = lists && S == 2 && Bc == e->B && dp_list2(e, l, d2);
    const size_t off2[2] = {0, lm2 ? (size_t)d2[0].K * e->N : Mp};       // first compact row of each stream's MLP rows
    const int Mm = lm ? dm.K * e->N : (lm2 ? (d2[0].K + d2[1].K) * e->N : (R > 0 ? R : Mall));
    if (lm2) {
        for (int st = 0; st < 2; ++st)
            CHECK(uvit_ln_fwd_keep_launch(x_mid + st * Mp * C, d2[st].pos, w.f + o.n2w, w.f + o.n2b, a.ln2 + off2[st] * C, a.mean2 + off2[st],
                                          a.rstd2 + off2[st], x_out + st * Mp * C, M, C, e->N, e->cfg.ln_eps, s));
    } else
    if (R > 0) CHECK(uvit_ln_fwd_gather_launch(x_mid, e->rowidx, e->count, w.f + o.n2w, w.f + o.n2b, a.ln2, a.mean2, a.rstd2, R, C, e->cfg.ln_eps, s));
    else if (lm) CHECK(uvit_ln_fwd_keep_launch(x_mid, dm.pos, w.f + o.n2w, w.f + o.n2b, a.ln2, a.mean2, a.rstd2, x_out, M, C, e->N, e->cfg.ln_eps, s));
    else CHECK(uvit_ln_fwd_launch(x_mid, w.f + o.n2w, w.f + o.n2b, a.ln2, a.mean2, a.rstd2, Mall, C, e->cfg.ln_eps, s));
    GemmEpi f1; f1.out = a.a; f1.out2 = save ? a.h : nullptr; f1.bias = w.f + o.fc1b; f1.ldo = Hd; f1.tile_counter = tcnt;
    const bool prof = prof_begin(save ? UVIT_PROF_FC1_S : UVIT_PROF_FC1_T, Mm);
    // student (save): a.h receives gelu'(h) -- all that backward needs of h -- computed beside gelu(h)
    CHECK(GEMM_NT(save ? EPI_GELU_DG : EPI_GELU, a.ln2, w.b + o.fc1w, Mm, Hd, C, C, C, &f1, s));
    prof_end(prof);
    for (int st = 0; st < S; ++st) {
        GemmEpi f2; f2.out = x_out + st * Mp * C; f2.out2 = save ? a.mlpout + st * Mp * C : nullptr; f2.bias = w.f + o.fc2b;
        f2.gamma = w.f + o.g2; f2.resid = x_mid + st * Mp * C; f2.rowscale = dp_ptr(e, dp_on, l, st, 1, Bc); f2.ldo = C; f2.tokens = e->N;
        if (R > 0) { f2.rowmap = e->rowidx; f2.rowcount = e->count; }
        else if (lm) { f2.rowmap = dm.rows; f2.rowcount = dm.cnt; }
        else if (lm2) { f2.rowmap = d2[st].rows; f2.rowcount = d2[st].cnt; }
        const int M2 = R > 0 ? R : (lm ? Mm : (lm2 ? d2[st].K * e->N : M));
        const bool pf2 = prof_begin(UVIT_PROF_FC2, M2);
        CHECK(GEMM_NT(EPI_RESID, a.a + (lm2 ? off2[st] : st * Mp) * Hd, w.b + o.fc2w, M2, C, Hd, Hd, Hd, &f2, s));
        prof_end(pf2);
    }
    return UVIT_OK;
}

// patch embedding + token assembly into x0 (modeling_cyclical.py:171-192; two-stream: modeling_cyclical_dist.py:106-130)
static int embed(uvit_engine* e, const Weights& w, const int64_t* mask, float* x0, int Bc, hipStream_t s) {
    for (int st = 0; st < e->S; ++st) {
        float* x = x0 + (size_t)st * e->Mpad * e->C;
        GemmEpi pe; pe.out = x; pe.bias = w.f + (st ? e->lo.cpeb : e->lo.peb); pe.mask = mask;
        pe.mask_token = w.f + (st ? e->lo.cmask_tok : e->lo.mask_tok); pe.ldo = e->C; pe.patches = e->P;
        CHECK(GEMM_NT(EPI_PATCH, e->cols, w.b + (st ? e->lo.cpew : e->lo.pew), Bc * e->P, e->C, e->Kpe, e->Kpe, e->Kpe, &pe, s));
        CHECK(uvit_set_cls_launch(x, w.f + (st ? e->lo.ccls : e->lo.cls), nullptr, Bc, e->N, e->C, s));
        // x = x + pos_embed (modeling_cyclical.py:193-194; --abs_pos_emb, off in every BASELINE config)
        if (e->cfg.use_abs_pos_emb) CHECK(uvit_add_pos_launch(x, w.f + e->lo.pos, Bc, e->N, e->C, s));
    }
    return UVIT_OK;
}

static int run_forward(uvit_engine* e, int which, const float* images, const int64_t* mask, int Bc, bool save_student,
                       bool dropout, uint32_t seed, uint32_t it, const uvit_step_params* hp_targets, bool cols_ready,
                       hipStream_t s) {
    if (Bc < 1 || Bc > e->B) return UVIT_ERR_SHAPE;
    const bool teacher = which == 1;
    Weights w{teacher ? e->buf.ema : e->buf.params, (const bf16*)(teacher ? e->buf.ema_bf16 : e->buf.params_bf16)};
    if (!cols_ready) CHECK(uvit_im2col_launch(images, e->cols, Bc, e->cfg.in_chans, e->cfg.img_size, e->cfg.patch_size, s));
    float* biasP = teacher ? e->biasP_t : e->biasP_s;
    CHECK(uvit_relpos_gather_launch(e->cfg.use_shared_rel_pos_bias ? w.f + e->lo.relt : nullptr, e->buf.rel_index, biasP,
                                    e->H, e->N, e->NP, s));
    const bool dp_on = dropout && !teacher && e->cfg.drop_path_rate > 0.f;
    const float pdrop = (dropout && !teacher) ? e->cfg.attn_drop_rate : 0.f;
    if (dp_on) CHECK(uvit_droppath_launch(e->dp_scales, e->dp_rates, e->cfg.depth, 2 * e->S, Bc, seed, it, s));
    if (!teacher) {
        e->dpl_on = false;
        if (dp_on && save_student && e->dpl_enable && Bc == e->B) CHECK(dp_lists_begin(e, seed, it, Bc, s));
    }
    const uint32_t aseed = uvit_hash32(seed ^ (it * 0x85EBCA6Bu + 0x1234567u));
    if (!teacher) { e->last_dropout = dropout; e->last_seed = aseed; e->last_it = it; }
    const bool use_saved = !teacher || !hp_targets;   // drop-in forward keeps every layer for either weight set
    float* x0 = use_saved ? e->X[0] : e->tX[0];
    CHECK(embed(e, w, mask, x0, Bc, s));
    int n_t = 0;
    for (int l = 0; l < e->cfg.depth; ++l) {
        if (use_saved) {
            const bool last_compact = save_student && !teacher && l == e->cfg.depth - 1 && e->compact_R > 0;
            CHECK(forward_layer(e, w, l, e->X[l], e->XM[l], e->X[l + 1], e->acts[l], save_student && !teacher, biasP, dp_on,
                                pdrop, aseed, Bc, s, last_compact ? e->compact_R : 0, !teacher && e->dpl_on));
        } else {
            float* xin = e->tX[l & 1]; float* xout = e->tX[(l + 1) & 1];
            // `[targets[i] for i in target_layers]` (engine_for_cyclical.py:92): a layer listed twice is summed twice and the
            // mean divides by len(target_layers); the host has already mapped negative indices and refused out-of-range ones
            const bool dense = hp_targets->target_batch_norm || hp_targets->target_instance_norm ||
                               hp_targets->post_target_instance_norm || !hp_targets->target_layer_norm_last;
            // the teacher's last block feeds nothing but the target rows: with a masked-row bound its MLP runs on those rows only
            const int Rt = (l == e->cfg.depth - 1 && !dense) ? e->compact_R : 0;
            CHECK(forward_layer(e, w, l, xin, e->tXM, xout, e->tacts, false, biasP, false, 0.f, 0, Bc, s, Rt));
            if (dense && Bc != e->B) return UVIT_ERR_ARG;
            for (int k = 0; k < hp_targets->n_target_layers; ++k) {
                if (hp_targets->target_layers[k] != l) continue;
                const float* sub = hp_targets->layer_results_fc ? e->tXM : nullptr;     // --layer_results fc: x_out - x_mid
                // Stream 0 (the only one of the base model; the MEAN targets of a stochastic step, engine_for_cyclical.py:93-118)
                // takes the dense builder when a batch- / instance-norm variant is on; the covariance targets (:73-86) only know
                // `target_layer_norm_last` and `post_target_layer_norm` and always go through the masked-row builder.
                for (int st = dense ? 1 : 0; st < e->S; ++st)
                    CHECK(uvit_target_accum_launch(xout + (size_t)st * e->Mpad * e->C, e->rowidx, e->count, e->targets[st], n_t == 0,
                                                   Bc * e->P, e->C, 1e-5f, s, sub ? sub + (size_t)st * e->Mpad * e->C : nullptr,
                                                   hp_targets->target_layer_norm_last ? 1 : 0));
                if (dense) {
                    // all patch tokens of this layer -> [batch norm] -> [instance norm] -> [LayerNorm] -> accumulate
                    CHECK(uvit_gather_patch_rows_launch(xout, sub, e->dense_v, Bc, e->P, e->C, s));
                    if (hp_targets->target_batch_norm) CHECK(uvit_colnorm_launch(e->dense_v, 1, Bc * e->P, e->C, 1e-5f, s));
                    if (hp_targets->target_instance_norm) CHECK(uvit_colnorm_launch(e->dense_v, Bc, e->P, e->C, 1e-5f, s));
                    if (hp_targets->target_layer_norm_last)
                        CHECK(uvit_target_accum_launch(e->dense_v, e->idrows, e->idcount, e->dense_acc, n_t == 0, Bc * e->P, e->C, 1e-5f, s));
                    else CHECK(uvit_axpy_rows_launch(e->dense_acc, e->dense_v, n_t == 0, (size_t)Bc * e->P * e->C, s));
                }
                ++n_t;
            }
        }
    }
    if (teacher && hp_targets) {
        if (n_t != hp_targets->n_target_layers) return UVIT_ERR_ARG;     // an index outside [0, depth) reached the C ABI
        const bool dense = hp_targets->target_batch_norm || hp_targets->target_instance_norm ||
                           hp_targets->post_target_instance_norm || !hp_targets->target_layer_norm_last;
        for (int st = dense ? 1 : 0; st < e->S; ++st)
            CHECK(uvit_target_finalize_launch(e->targets[st], e->count, n_t, hp_targets->post_target_layer_norm, Bc * e->P, e->C, 1e-5f, s));
        if (dense) {
            const bool pin = hp_targets->post_target_instance_norm != 0;
            CHECK(uvit_target_finalize_launch(e->dense_acc, e->idcount, n_t, hp_targets->post_target_layer_norm && !pin, Bc * e->P, e->C, 1e-5f, s));
            if (pin) {
                CHECK(uvit_colnorm_launch(e->dense_acc, Bc, e->P, e->C, 1e-5f, s));
                if (hp_targets->post_target_layer_norm)
                    CHECK(uvit_target_finalize_launch(e->dense_acc, e->idcount, 1, 1, Bc * e->P, e->C, 1e-5f, s));
            }
            CHECK(uvit_gather_masked_rows_launch(e->dense_acc, e->rowidx, e->count, e->targets[0], Bc * e->P, e->P, e->C, s));
        }
    }
    e->cur_B = Bc;
    return UVIT_OK;
}

extern "C" int uvit_engine_forward_features(uvit_engine* e, int which, const float* images, const int64_t* mask, int batch,
                                            int train_dropout, uint32_t seed, uint32_t it, uvit_stream stream) {
    if (!e || !images || (which != 0 && which != 1)) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (mask) CHECK(uvit_mask_compact_launch(mask, e->rowidx, e->count, batch, e->P, s));
    return run_forward(e, which, images, mask, batch, false, train_dropout != 0, seed, it, nullptr, false, s);
}

// final norm (shared) + drop cls + masked-row gather + per-stream head (modeling_cyclical.py:207,215-225)
static int head_forward(uvit_engine* e, const Weights& w, int Bc, int st, bool all_tokens, float* out, hipStream_t s, int rows = 0) {
    const int BP = rows > 0 ? rows : Bc * e->P;        // rows > 0: a host-side bound on the masked rows (training step with n_rows_hint)
    const float* x = e->X[e->cfg.depth] + (size_t)st * e->Mpad * e->C;
    const size_t lmw = st ? e->lo.clmw : e->lo.lmw, lmb = st ? e->lo.clmb : e->lo.lmb;
    if (all_tokens) {
        // normalise all tokens, then run the head on the patch rows of each sample
        CHECK(uvit_ln_fwd_launch(x, w.f + e->lo.normw, w.f + e->lo.normb, e->acts[0].ln1, e->acts[0].mean1, e->acts[0].rstd1,
                                 Bc * e->N, e->C, e->cfg.ln_eps, s));
        for (int b = 0; b < Bc; ++b) {
            GemmEpi h; h.out = out + (size_t)b * e->P * e->C; h.bias = w.f + lmb; h.ldo = e->C;
            CHECK(GEMM_NT(EPI_F32, e->acts[0].ln1 + ((size_t)b * e->N + 1) * e->C, w.b + lmw, e->P, e->C, e->C,
                                      e->C, e->C, &h, s));
        }
        return UVIT_OK;
    }
    CHECK(uvit_ln_fwd_gather_launch(x, e->rowidx, e->count, w.f + e->lo.normw, w.f + e->lo.normb, e->normed[st], e->meanF[st],
                                    e->rstdF[st], BP, e->C, e->cfg.ln_eps, s));
    GemmEpi h; h.out = out; h.bias = w.f + lmb; h.ldo = e->C;
    CHECK(GEMM_NT(EPI_F32, e->normed[st], w.b + lmw, BP, e->C, e->C, e->C, e->C, &h, s));
    return UVIT_OK;
}

extern "C" int uvit_engine_head(uvit_engine* e, int which, int all_tokens, float* out, int32_t* count_dev, uvit_stream stream) {
    if (!e || !out) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool teacher = (which & 1) != 0;
    const int st = (which >> 1) & 1;                     // bit 1 selects the covariance stream of the two-stream model
    if (st >= e->S) return UVIT_ERR_ARG;
    Weights w{teacher ? e->buf.ema : e->buf.params, (const bf16*)(teacher ? e->buf.ema_bf16 : e->buf.params_bf16)};
    CHECK(head_forward(e, w, e->cur_B, st, all_tokens != 0, out, s));
    if (count_dev && !all_tokens) HIPCHECK(hipMemcpyAsync(count_dev, e->count, sizeof(int), hipMemcpyDeviceToDevice, s));
    return UVIT_OK;
}

extern "C" int uvit_engine_compact_rows(uvit_engine* e) { return e ? e->compact_R : 0; }

extern "C" void* uvit_engine_ws_ptr(uvit_engine* e, const char* name, int layer) {
    if (!e || !name) return nullptr;
    const std::string n(name);
    const size_t cov = (size_t)e->Mpad * e->C;           // element offset of the covariance stream in stacked buffers
    if (n == "x" && layer >= 0 && layer <= e->cfg.depth) return e->X[layer];
    if (n == "xm" && layer >= 0 && layer < e->cfg.depth) return e->XM[layer];
    if (n == "x_cov" && e->S == 2 && layer >= 0 && layer <= e->cfg.depth) return e->X[layer] + cov;
    if (n == "xm_cov" && e->S == 2 && layer >= 0 && layer < e->cfg.depth) return e->XM[layer] + cov;
    if (n == "loss") return e->loss;
    if (n == "grad_norm") return e->gnorm;
    if (n == "targets") return e->targets[0];
    if (n == "outputs") return e->outputs[0];
    if (n == "targets_cov" && e->S == 2) return e->targets[1];
    if (n == "outputs_cov" && e->S == 2) return e->outputs[1];
    if (n == "count") return e->count;
    if (n == "dx") return e->dXa;
    return nullptr;
}

// ---- training step ----
extern "C" int uvit_step_begin(uvit_engine* e, const float* images, const int64_t* mask, const uvit_step_params* hp,
                               uvit_stream stream) {
    if (!e || !images || !mask || !hp || hp->n_target_layers < 1 || hp->n_target_layers > UVIT_MAX_DEPTH) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int Bc = e->B, BP = e->BP, C = e->C;
    Layout& lo = e->lo;
    // every gradient is accumulated (split-K wgrad atomics, bias/LN/gamma column sums): zero the arena once per step.
    // With two streams the ~440 MB of memsets ride on the second stream ahead of the teacher forward (which has slack
    // against the student forward); their first consumers run after the ev_teacher join below.
    hipStream_t ts = e->dual ? e->aux : s;
    if (e->dual) { HIPCHECK(hipEventRecord(e->ev_fork, s)); HIPCHECK(hipStreamWaitEvent(ts, e->ev_fork, 0)); }
    CHECK(uvit_zero_launch(e->buf.grads, lo.n_live * sizeof(float), ts));      // frozen tensors never receive a gradient
    CHECK(uvit_zero_launch(e->grep, (size_t)NREP * e->n_nd * sizeof(float), ts));
    CHECK(uvit_zero_launch(e->loss, 64 * sizeof(float), ts));
    CHECK(uvit_zero_launch(e->dXa, e->rows_alloc() * C * sizeof(float), ts));
    e->slab_started = false; e->ls_prefused = -1;
    // the last block's MLP on the masked rows only (see forward_layer): base model, a row bound from the host, and fewer rows than tokens
    e->compact_R = 0;
    if (e->S == 1 && hp->n_rows_hint > 0) {
        const int R = (int)roundup((size_t)(hp->n_rows_hint < BP ? hp->n_rows_hint : BP), 64);
        if (R >= 512 && R < e->M) {
            e->compact_R = R;
            // its LayerNorm backward writes the residual-stream gradient and the proj branch's dY at the listed rows only: the rest is zero
            CHECK(uvit_zero_launch(e->dXb, e->rows_alloc() * C * sizeof(float), ts));
            CHECK(uvit_zero_launch(e->dY2[(e->cfg.depth - 1) & 1], e->rows_alloc() * C * sizeof(bf16), ts));
        }
    }
    HIPCHECK(hipMemcpyAsync(e->mask_copy, mask, (size_t)BP * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
    CHECK(uvit_mask_compact_launch(mask, e->rowidx, e->count, Bc, e->P, s));
    CHECK(uvit_im2col_launch(images, e->cols, Bc, e->cfg.in_chans, e->cfg.img_size, e->cfg.patch_size, s));
    // teacher (EMA weights, eval mode, no grad: engine_for_cyclical.py:68-122) runs on the second stream,
    // beside the student forward (engine_for_cyclical.py:124-128); they share only read-only inputs
    if (e->dual) { HIPCHECK(hipEventRecord(e->ev_fork, s)); HIPCHECK(hipStreamWaitEvent(ts, e->ev_fork, 0)); }   // im2col / mask list
    CHECK(run_forward(e, 1, images, nullptr, Bc, false, false, 0, 0, hp, true, ts));
    if (e->dual) HIPCHECK(hipEventRecord(e->ev_teacher, ts));
    CHECK(run_forward(e, 0, images, mask, Bc, true, hp->train_dropout != 0, hp->seed, hp->it, nullptr, true, s));
    Weights w{e->buf.params, (const bf16*)e->buf.params_bf16};
    const int Rh = e->compact_R > 0 ? e->compact_R : BP;      // rows of the head and of its backward (rows beyond the device-side count are zero)
    for (int st = 0; st < e->S; ++st) CHECK(head_forward(e, w, Bc, st, false, e->outputs[st], s, e->compact_R));
    if (e->dual) HIPCHECK(hipStreamWaitEvent(s, e->ev_teacher, 0));
    // loss + dLoss/dOutputs: engine_for_cyclical.py:130-163 (+ WassersteinLoss for the two-stream model, :152-161)
    const float ls = hp->loss_scale == -1.0f ? 1.0f : hp->loss_scale;
    CHECK(uvit_smooth_l1_launch(e->outputs[0], e->targets[0], e->count, hp->l1_beta, hp->l2_loss, ls, e->loss, e->dout[0], BP, C, s));
    if (hp->var_w0 > 0.f)      // variance term on the mean-stream outputs (engine_for_cyclical.py:130-139); std_loss0 -> loss[4]
        CHECK(uvit_variance_loss_launch(e->outputs[0], e->count, hp->var_w0, hp->var_margin0, ls, e->var_scratch, e->loss, e->loss + 4,
                                        e->dout[0], BP, C, s));
    if (e->compact_R > 0) CHECK(uvit_rows_guard_launch(e->count, e->compact_R, e->loss, s));     // more masked rows than the host promised
    if (e->dpl_on) CHECK(uvit_droppath_lists_guard_launch(e->dpl_cnt, 2 * e->S * e->cfg.depth, e->loss, s));   // a sample list that is not the host's
    if (e->S == 2)
        CHECK(uvit_wasserstein_loss_launch(e->outputs[0], e->outputs[1], e->targets[0], e->targets[1], e->count, hp->lambda_pretraining,
                                           ls, e->wl_scratch, e->loss, e->dout[0], e->dout[1], BP, C, s));
    // head backward
    float* g = e->buf.grads;
    const bf16* wt = (const bf16*)e->buf.params_bf16_t;
    for (int st = 0; st < e->S; ++st) {
        const size_t lmw = st ? lo.clmw : lo.lmw, lmb = st ? lo.clmb : lo.lmb;
        // lm_head weight gradient + bias column sum: one launch of the grouped wgrad kernel when the shapes qualify (was a column-sum launch
        // and the plain TN kernel, 50 us, on the critical chain between the loss and the first dgrad)
        TnProb hw; hw.Y = e->dout[st]; hw.X = e->normed[st]; hw.C = g + lmw; hw.M = e->compact_R > 0 ? e->compact_R : e->BPpad; hw.Nn = C; hw.Kk = C;
        hw.ldy = C; hw.ldx = C; hw.ldc = C; hw.bias = RP(lmb); hw.bias_end = C;
        if (uvit_gemm_tn_group_ok(&hw, 1, &e->tune)) CHECK(GEMM_TN_GROUP(&hw, 1, s));
        else {
            CHECK(uvit_colsum_launch(e->dout[st], C, 0, C, Rh, RP(lmb), NREP, e->n_nd, s));
            CHECK(GEMM_TN(e->dout[st], e->normed[st], e->compact_R > 0 ? e->compact_R : e->BPpad, C, C, C, C, g + lmw, C, 1, s));
        }
        GemmEpi d; d.out = e->dnormed[st]; d.ldo = C;
        CHECK(GEMM_NT(EPI_BF16, e->dout[st], wt + lmw, Rh, C, C, C, C, &d, s));
        // final LayerNorm backward scattered into the (zeroed) residual-stream gradient
        CHECK(uvit_ln_bwd_scatter_launch(e->dnormed[st], e->X[e->cfg.depth] + (size_t)st * e->Mpad * C, e->rowidx, e->count,
                                         e->meanF[st], e->rstdF[st], e->buf.params + lo.normw, e->dXa + (size_t)st * e->Mpad * C,
                                         RP(lo.normw), RP(lo.normb), Rh, C, NREP, e->n_nd, s));
    }
    return UVIT_OK;
}

extern "C" int uvit_step_backward_layer(uvit_engine* e, int l, const uvit_step_params* hp, uvit_stream stream) {
    if (!e || l < 0 || l >= e->cfg.depth) return UVIT_ERR_ARG;
    (void)hp;
    hipStream_t s = (hipStream_t)stream;
    const LayerOff& o = e->lo.L[l];
    LayerActs& a = e->acts[l];
    const int M = e->M, C = e->C, Hd = e->Hd, S = e->S;
    const size_t Mp = e->Mpad;
    const int Mall = (int)e->rows_all();                      // stacked rows of a row-wise op
    const int Mred = (int)roundup(e->rows_all(), 64);          // stacked reduction length of a wgrad (pad rows are zero in dY)
    const int Mred1 = (int)roundup(M, 64);                     // one stream
    float* g = e->buf.grads;
    const float* pf = e->buf.params;
    const bf16* wt = (const bf16*)e->buf.params_bf16_t;
    const bool dp_on = e->last_dropout && e->cfg.drop_path_rate > 0.f;
    const float pdrop = e->last_dropout ? e->cfg.attn_drop_rate : 0.f;
    // wgrad GEMMs + bias column sums go to the second stream; the dgrad chain stays on `s`.  The
    // gradient buffers they read are double-buffered by layer parity; before reusing a parity the
    // main stream waits for the wgrad work of layer l+2.
    const int par = l & 1;
    hipStream_t ws = e->dual ? e->aux : s;
    auto handoff = [&](int k) -> int {          // work enqueued on `s` so far is visible to the wgrad stream
        if (!e->dual) return UVIT_OK;
        HIPCHECK(hipEventRecord(e->ev_x[k], s));
        HIPCHECK(hipStreamWaitEvent(ws, e->ev_x[k], 0));
        return UVIT_OK;
    };
    if (e->dual && l + 2 < e->cfg.depth) HIPCHECK(hipStreamWaitEvent(s, e->ev_wdone[l + 2], 0));
    bf16 *dY1 = e->dY1[par], *dY2 = e->dY2[par], *dH = e->dH[par], *dqkv = e->dqkv[par];
    // last block with a masked-row list (see forward_layer): its MLP branch -- LayerScale backward, both dgrads, both wgrads, LN2 backward --
    // runs on the R compact rows; dY1 / dH / dLN are compact, the LN2 backward scatters into the (zeroed) dense dXb / dY2
    const int R = (l == e->cfg.depth - 1) ? e->compact_R : 0;
    // drop-path sample lists (see forward_layer): the attention branch on Ma = Ka tokens rows, the MLP branch on Km tokens rows; the wgrad
    // reductions run to the next multiple of 64 (pad rows of the dY operands are zero)
    DpList da{}, dm{}, dm1{};
    const bool la = S == 1 && dp_list(e, l, 0, da);
    const bool lm = R == 0 && S == 1 && dp_list(e, l, 1, dm);
    const bool lm1 = l > 0 && S == 1 && dp_list(e, l - 1, 1, dm1);          // the MLP branch of the layer below (its LayerScale backward rides here)
    const int Ma = la ? da.K * e->N : M, Mac = la ? (int)roundup(Ma, 64) : Mred1;
    // two-stream model: the MLP branch on the kept rows of both streams, stacked without a gap (see forward_layer)
    DpList dl2[2] = {}, dl21[2] = {};
    const bool lm2 = dp_list2(e, l, dl2), lm21 = l > 0 && dp_list2(e, l - 1, dl21);
    const size_t off2[2] = {0, lm2 ? (size_t)dl2[0].K * e->N : Mp}, off21[2] = {0, lm21 ? (size_t)dl21[0].K * e->N : Mp};
    const int Mmlp = R > 0 ? R : (lm ? (int)roundup((size_t)dm.K * e->N, 64) : (lm2 ? (int)roundup((size_t)(dl2[0].K + dl2[1].K) * e->N, 64) : Mall));
    const int Mmlp_red = (R > 0 || lm || lm2) ? Mmlp : Mred;
    // weight gradients: one grouped launch per layer (bias column sums of fc1 / q / v fused) when every Linear has
    // 256-multiple dimensions; otherwise one launch per Linear as the operands become available
    TnProb wg[UVIT_TN_GROUP_MAX];
    int nwg = 0;
    {
        TnProb& f2 = wg[nwg++]; f2.Y = dY1; f2.X = a.a; f2.C = g + o.fc2w; f2.M = Mmlp_red; f2.Nn = C; f2.Kk = Hd; f2.ldy = C; f2.ldx = Hd; f2.ldc = Hd;
        TnProb& f1 = wg[nwg++]; f1.Y = dH; f1.X = a.ln2; f1.C = g + o.fc1w; f1.M = Mmlp_red; f1.Nn = Hd; f1.Kk = C; f1.ldy = Hd; f1.ldx = C; f1.ldc = C;
        f1.bias = RP(o.fc1b); f1.bias_end = Hd;
        for (int st = 0; st < S; ++st) {
            TnProb& pj = wg[nwg++]; pj.Y = dY2 + st * Mp * C; pj.X = a.attn + st * Mp * C; pj.C = g + off_projw(o, st);
            pj.M = Mac; pj.Nn = C; pj.Kk = C; pj.ldy = C; pj.ldx = C; pj.ldc = C;
        }
        TnProb& qk = wg[nwg++]; qk.Y = dqkv; qk.X = a.ln1; qk.C = g + o.qkvw; qk.M = la ? Mac : Mred; qk.Nn = 3 * C; qk.Kk = C; qk.ldy = 3 * C; qk.ldx = C; qk.ldc = C;
        qk.bias = RP(off_qb(o, 0)); qk.bias_end = C; qk.bias2 = RP(off_vb(o, 0)); qk.bias2_begin = 2 * C;
        // two-stream: the q / v biases differ per stream while the stacked wgrad reduces over both: the token chunks of stream 1
        // (rows from Mpad on) sum into the covariance stream's biases (round 4; was 4 colsum launches per layer)
        if (S == 2) { qk.bias_s1 = RP(off_qb(o, 1)); qk.bias2_s1 = RP(off_vb(o, 1)); qk.s1_row = (int)Mp; }
    }
    const bool grouped = uvit_gemm_tn_group_ok(wg, nwg, &e->tune);
    // --- MLP branch: x_out = x_mid + dp * gamma2 * fc2(gelu(fc1(ln2(x_mid))))   (weights shared by the streams)
    // (the LayerScale backward of a branch rides in the LayerNorm backward that produces its input; in the two-stream
    //  model that kernel is launched once per stream, because drop-path scales and the proj bias differ per stream)
    const bool fuse_ls = true;
    if (R > 0)       // compact dY1 from the residual-stream gradient and the saved branch output at the listed rows (S == 1)
        CHECK(uvit_ls_bwd_launch(e->dXa, a.mlpout, pf + o.g2, dp_ptr(e, dp_on, l, 0, 1, e->B), dY1, RP(o.g2), RP(o.fc2b), R, C, e->N, NREP,
                                 e->n_nd, s, e->rowidx, e->count));
    else if (e->ls_prefused != l && lm2)   // two-stream: compact dY1 of each stream's kept samples, stacked; the second launch zero-fills the pad rows
        for (int st = 0; st < 2; ++st)
            CHECK(uvit_ls_bwd_launch(e->dXa + st * Mp * C, a.mlpout + st * Mp * C, pf + o.g2, dp_ptr(e, dp_on, l, st, 1, e->B), dY1 + off2[st] * C,
                                     RP(o.g2), RP(o.fc2b), st == 0 ? dl2[0].K * e->N : Mmlp - dl2[0].K * e->N, C, e->N, NREP, e->n_nd, s,
                                     dl2[st].rows, dl2[st].cnt));
    else if (e->ls_prefused != l && lm)    // compact dY1 of the kept samples (pad rows zero)
        CHECK(uvit_ls_bwd_launch(e->dXa, a.mlpout, pf + o.g2, dp_ptr(e, dp_on, l, 0, 1, e->B), dY1, RP(o.g2), RP(o.fc2b), Mmlp, C, e->N, NREP,
                                 e->n_nd, s, dm.rows, dm.cnt));
    else if (e->ls_prefused != l)
        for (int st = 0; st < S; ++st)
            CHECK(uvit_ls_bwd_launch(e->dXa + st * Mp * C, a.mlpout + st * Mp * C, pf + o.g2, dp_ptr(e, dp_on, l, st, 1, e->B), dY1 + st * Mp * C,
                                     RP(o.g2), RP(o.fc2b), M, C, e->N, NREP, e->n_nd, s));
    e->ls_prefused = -1;
    if (!grouped) {
        CHECK(handoff(0));
        CHECK(GEMM_TN(dY1, a.a, Mmlp_red, C, Hd, C, Hd, g + o.fc2w, Hd, 1, ws));
    }
    GemmEpi d1; d1.out = dH; d1.aux = a.h; d1.ldo = Hd;
    CHECK(GEMM_NT(EPI_MULAUX, dY1, wt + o.fc2w, Mmlp, Hd, C, C, C, &d1, s));      // dH = (dY.W2) * gelu'(h)
    if (!grouped) {
        CHECK(handoff(1));
        CHECK(uvit_colsum_launch(dH, Hd, 0, Hd, Mmlp, RP(o.fc1b), NREP, e->n_nd, ws));
        CHECK(GEMM_TN(dH, a.ln2, Mmlp_red, Hd, C, Hd, C, g + o.fc1w, C, 1, ws));
    }
    GemmEpi d2; d2.out = e->dLN; d2.ldo = C;
    CHECK(GEMM_NT(EPI_BF16, dH, wt + o.fc1w, Mmlp, C, Hd, Hd, Hd, &d2, s));
    // --- attention branch: x_mid = x_in + dp * gamma1 * proj(attn(ln1(x_in)))   (proj differs per stream)
    if (lm2) {
        // two-stream: LayerNorm 2 backward through each stream's MLP list; the attention branch's LayerScale backward stays dense
        for (int st = 0; st < 2; ++st) {
            const size_t ro = st * Mp, eo = ro * C;
            CHECK(uvit_ln_bwd_keep_launch(e->dLN + off2[st] * C, e->XM[l] + eo, dl2[st].pos, a.mean2 + off2[st], a.rstd2 + off2[st], pf + o.n2w,
                                          e->dXa + eo, e->dXb + eo, RP(o.n2w), RP(o.n2b), a.projout + eo, pf + o.g1, dp_ptr(e, dp_on, l, st, 0, e->B),
                                          dY2 + eo, RP(o.g1), RP(off_projb(o, st)), nullptr, nullptr, e->N, M, C, NREP, e->n_nd, s));
        }
    } else if (R == 0 && (la || lm)) {
        // LayerNorm 2 backward (MLP list) + the attention branch's LayerScale backward (attention list) over the dense rows
        CHECK(uvit_ln_bwd_keep_launch(e->dLN, e->XM[l], lm ? dm.pos : nullptr, a.mean2, a.rstd2, pf + o.n2w, e->dXa, e->dXb, RP(o.n2w), RP(o.n2b),
                                      a.projout, pf + o.g1, dp_ptr(e, dp_on, l, 0, 0, e->B), dY2, RP(o.g1), RP(off_projb(o, 0)),
                                      la ? da.pos : nullptr, la ? da.cnt : nullptr, e->N, M, C, NREP, e->n_nd, s, 0,
                                      la ? dqkv : nullptr, 3 * C));      // (also zero-fills the pad rows of dqkv: the attention backward writes Ka samples)
    } else if (fuse_ls) {
        for (int st = 0; st < S; ++st) {
            const size_t ro = st * Mp, eo = ro * C;
            CHECK(uvit_ln_bwd_ls_launch(e->dLN + eo, e->XM[l] + eo, a.mean2 + ro, a.rstd2 + ro, pf + o.n2w, e->dXa + eo, e->dXb + eo,
                                        RP(o.n2w), RP(o.n2b), a.projout + eo, pf + o.g1, dp_ptr(e, dp_on, l, st, 0, e->B), dY2 + eo,
                                        RP(o.g1), RP(off_projb(o, st)), e->N, R > 0 ? R : M, C, NREP, e->n_nd, s,
                                        R > 0 ? e->rowidx : nullptr, R > 0 ? e->count : nullptr,
                                        (R > 0 && la) ? da.pos : nullptr));      // masked-row last block: dY2 compact by the attention list (rest pre-zeroed)
        }
    } else {
        CHECK(uvit_ln_bwd_launch(e->dLN, e->XM[l], a.mean2, a.rstd2, pf + o.n2w, e->dXa, e->dXb, RP(o.n2w), RP(o.n2b), Mall, C, NREP, e->n_nd, s));
        for (int st = 0; st < S; ++st)
            CHECK(uvit_ls_bwd_launch(e->dXb + st * Mp * C, a.projout + st * Mp * C, pf + o.g1, dp_ptr(e, dp_on, l, st, 0, e->B), dY2 + st * Mp * C,
                                     RP(o.g1), RP(off_projb(o, st)), M, C, e->N, NREP, e->n_nd, s));
    }
    if (!grouped) CHECK(handoff(2));
    for (int st = 0; st < S; ++st) {
        if (!grouped) CHECK(GEMM_TN(dY2 + st * Mp * C, a.attn + st * Mp * C, Mac, C, C, C, C, g + off_projw(o, st), C, 1, ws));
        GemmEpi d3; d3.out = e->dAttn + st * Mp * C; d3.ldo = C;
        CHECK(GEMM_NT(EPI_BF16, dY2 + st * Mp * C, wt + off_projw(o, st), Ma, C, C, C, C, &d3, s));
    }
    const float* biasP = e->biasP_s;
    float* slabs = e->cfg.use_shared_rel_pos_bias ? e->slabs : nullptr;
    {
        // fused backward (both model families): every gradient from one recomputation of P; dS (bf16) goes to the parity buffer and
        // its batch reduction into the ONE bias-gradient slab runs on the second stream, beside the dgrad chain.  The parity buffer
        // of layer l was last read by the reduction of layer l + 2, which precedes ev_wdone[l + 2] on that stream (waited for above).
        void* dsw = slabs ? e->ds_ws[par] : nullptr;
        if (S == 1) {
            // (a compact launch writes Ka samples: the rows up to the wgrad's reduction length are zero-filled by the LayerNorm 2 backward above,
            //  or here in the masked-row last block, whose LayerNorm 2 backward is the row-list kernel)
            if (la && R > 0 && Mac > Ma) CHECK(uvit_zero_launch(dqkv + (size_t)Ma * 3 * C, (size_t)(Mac - Ma) * 3 * C * sizeof(bf16), s));
            CHECK(uvit_attn_bwd_fused_launch(a.qkv, a.attn, e->dAttn, biasP, a.lse, e->delta, dqkv, dsw, dsw != nullptr, la ? da.K : e->B, e->H,
                                             e->N, e->NP, 0.125f, pdrop, e->last_seed, (uint32_t)l, s, la ? da.bmap : nullptr));
        } else {
            CHECK(uvit_attn2_bwd_launch(a.qkv, a.qkv + Mp * 3 * C, a.attn, a.attn + Mp * C, e->dAttn, e->dAttn + Mp * C, biasP, a.lse,
                                        e->delta, dqkv, dqkv + Mp * 3 * C, dsw, dsw != nullptr, e->B, e->H, e->N, e->NP, 0.125f, pdrop,
                                        e->last_seed, (uint32_t)l, s));
        }
        if (dsw) {
            if (e->dual) { HIPCHECK(hipEventRecord(e->ev_ds, s)); HIPCHECK(hipStreamWaitEvent(ws, e->ev_ds, 0)); }
            if (S == 1) CHECK(uvit_attn_dbias_reduce_launch(dsw, slabs, e->slab_started ? 1 : 0, la ? da.K : e->B, e->H, e->N, e->NP, ws));
            else CHECK(uvit_attn2_dbias_reduce_launch(dsw, slabs, e->slab_started ? 1 : 0, e->B, e->H, e->N, e->NP, ws));
        }
    }
    e->slab_started = true;
    CHECK(handoff(3));
    if (!grouped)
        for (int st = 0; st < S; ++st) {
            CHECK(uvit_colsum_launch(dqkv + st * Mp * 3 * C, 3 * C, 0, C, Ma, RP(off_qb(o, st)), NREP, e->n_nd, ws));
            CHECK(uvit_colsum_launch(dqkv + st * Mp * 3 * C, 3 * C, 2 * C, C, Ma, RP(off_vb(o, st)), NREP, e->n_nd, ws));
        }
    if (grouped) CHECK(GEMM_TN_GROUP(wg, nwg, ws));
    else CHECK(GEMM_TN(dqkv, a.ln1, la ? Mac : Mred, 3 * C, C, 3 * C, C, g + o.qkvw, C, 1, ws));
    if (e->dual) HIPCHECK(hipEventRecord(e->ev_wdone[l], ws));
    GemmEpi d4; d4.out = e->dLN; d4.ldo = C;
    CHECK(GEMM_NT(EPI_BF16, dqkv, wt + o.qkvw, la ? Ma : Mall, C, 3 * C, 3 * C, 3 * C, &d4, s));
    if (lm21) {
        // two-stream: LayerNorm 1 backward (dense) + the LayerScale backward of layer l-1's MLP branch into its stacked compact dY1
        if (e->dual && l + 1 < e->cfg.depth) HIPCHECK(hipStreamWaitEvent(s, e->ev_wdone[l + 1], 0));
        const LayerOff& on = e->lo.L[l - 1];
        for (int st = 0; st < 2; ++st) {
            const size_t ro = st * Mp, eo = ro * C;
            CHECK(uvit_ln_bwd_keep_launch(e->dLN + eo, e->X[l] + eo, nullptr, a.mean1 + ro, a.rstd1 + ro, pf + o.n1w, e->dXb + eo, e->dXa + eo,
                                          RP(o.n1w), RP(o.n1b), e->acts[l - 1].mlpout + eo, pf + on.g2, dp_ptr(e, dp_on, l - 1, st, 1, e->B),
                                          e->dY1[(l - 1) & 1] + off21[st] * C, RP(on.g2), RP(on.fc2b), dl21[st].pos, st == 1 ? dl21[1].cnt : nullptr,
                                          e->N, M, C, NREP, e->n_nd, s, st == 1 ? dl21[0].K * e->N : 0));
        }
        e->ls_prefused = l - 1;
    } else if (la || lm1) {
        // LayerNorm 1 backward (attention list) + the LayerScale backward of layer l-1's MLP branch (its list), dense rows
        if (l > 0 && e->dual && l + 1 < e->cfg.depth) HIPCHECK(hipStreamWaitEvent(s, e->ev_wdone[l + 1], 0));
        const LayerOff& on = e->lo.L[l > 0 ? l - 1 : 0];
        CHECK(uvit_ln_bwd_keep_launch(e->dLN, e->X[l], la ? da.pos : nullptr, a.mean1, a.rstd1, pf + o.n1w, e->dXb, e->dXa, RP(o.n1w), RP(o.n1b),
                                      l > 0 ? e->acts[l - 1].mlpout : nullptr, l > 0 ? pf + on.g2 : nullptr,
                                      l > 0 ? dp_ptr(e, dp_on, l - 1, 0, 1, e->B) : nullptr, l > 0 ? e->dY1[(l - 1) & 1] : nullptr,
                                      l > 0 ? RP(on.g2) : nullptr, l > 0 ? RP(on.fc2b) : nullptr, lm1 ? dm1.pos : nullptr, lm1 ? dm1.cnt : nullptr,
                                      e->N, M, C, NREP, e->n_nd, s));
        if (l > 0) e->ls_prefused = l - 1;
    } else if (fuse_ls && l > 0) {
        // the MLP-branch LayerScale backward of layer l-1 writes dY1 of parity (l-1) & 1, last read by the wgrad of layer l+1
        if (e->dual && l + 1 < e->cfg.depth) HIPCHECK(hipStreamWaitEvent(s, e->ev_wdone[l + 1], 0));
        const LayerOff& on = e->lo.L[l - 1];
        for (int st = 0; st < S; ++st) {
            const size_t ro = st * Mp, eo = ro * C;
            CHECK(uvit_ln_bwd_ls_launch(e->dLN + eo, e->X[l] + eo, a.mean1 + ro, a.rstd1 + ro, pf + o.n1w, e->dXb + eo, e->dXa + eo,
                                        RP(o.n1w), RP(o.n1b), e->acts[l - 1].mlpout + eo, pf + on.g2, dp_ptr(e, dp_on, l - 1, st, 1, e->B),
                                        e->dY1[(l - 1) & 1] + eo, RP(on.g2), RP(on.fc2b), e->N, M, C, NREP, e->n_nd, s));
        }
        e->ls_prefused = l - 1;
    } else {
        CHECK(uvit_ln_bwd_launch(e->dLN, e->X[l], a.mean1, a.rstd1, pf + o.n1w, e->dXb, e->dXa, RP(o.n1w), RP(o.n1b), Mall, C, NREP, e->n_nd, s));
    }
    return UVIT_OK;
}

extern "C" int uvit_step_backward_embed(uvit_engine* e, uvit_stream stream) {
    if (!e) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Layout& lo = e->lo;
    float* g = e->buf.grads;
    const int C = e->C, BP = e->BP;
    // token assembly backward: d cls_token, d mask_token, gradient of the patch-embedding output; then the patch-embedding weight gradient and
    // its bias column sum as ONE grouped launch (end of round 4; was a column-sum launch and the plain TN kernel per stream, 137 us each at the
    // very end of backward) when the shapes qualify
    TnProb pe[2];
    for (int st = 0; st < e->S; ++st) {
        CHECK(uvit_token_bwd_launch(e->dXa + (size_t)st * e->Mpad * C, e->mask_copy, e->dpatch[st], g + (st ? lo.ccls : lo.cls),
                                    g + (st ? lo.cmask_tok : lo.mask_tok), e->B, e->P, C, s));
        TnProb& q = pe[st]; q.Y = e->dpatch[st]; q.X = e->cols; q.C = g + (st ? lo.cpew : lo.pew); q.M = (int)roundup(BP, 64); q.Nn = C; q.Kk = e->Kpe;
        q.ldy = C; q.ldx = e->Kpe; q.ldc = e->Kpe; q.bias = RP(st ? lo.cpeb : lo.peb); q.bias_end = C;
    }
    if (uvit_gemm_tn_group_ok(pe, e->S, &e->tune)) CHECK(GEMM_TN_GROUP(pe, e->S, s));
    else
        for (int st = 0; st < e->S; ++st) {
            CHECK(uvit_colsum_launch(e->dpatch[st], C, 0, C, BP, RP(st ? lo.cpeb : lo.peb), NREP, e->n_nd, s));
            CHECK(GEMM_TN(e->dpatch[st], e->cols, (int)roundup(BP, 64), C, e->Kpe, C, e->Kpe, g + (st ? lo.cpew : lo.pew), e->Kpe, 1, s));
        }
    if (e->cfg.use_abs_pos_emb) CHECK(uvit_pos_bwd_launch(e->dXa, g + lo.pos, e->B, e->N, C, s));     // d pos_embed = sum_b dX[b]
    if (e->dual) HIPCHECK(hipStreamWaitEvent(s, e->ev_wdone[0], 0));   // every wgrad / bias sum / bias-gradient reduction has landed
    if (e->cfg.use_shared_rel_pos_bias && e->slab_started)
        CHECK(uvit_relpos_scatter_launch(e->slabs, 1, e->buf.rel_index, g + lo.relt, e->H, e->N, e->NP, s));
    // fold the replicated column-sum accumulators into the no-decay gradients
    CHECK(uvit_reduce_replicas_launch(e->grep, g + lo.n_decay, e->n_nd, NREP, e->n_nd, s));
    CHECK(uvit_poison_if_nonfinite_launch(e->loss, g + lo.n_decay, e->poisoned, s));     // non-finite loss -> every rank's norm is NaN
    return UVIT_OK;
}


extern "C" int uvit_step_wait_layer_grads(uvit_engine* e, int layer, uvit_stream stream) {
    if (!e || layer < 0 || layer >= e->cfg.depth) return UVIT_ERR_ARG;
    if (e->dual) HIPCHECK(hipStreamWaitEvent((hipStream_t)stream, e->ev_wdone[layer], 0));
    return UVIT_OK;
}

extern "C" int uvit_step_update(uvit_engine* e, const uvit_step_params* hp, uvit_stream stream) {
    if (!e || !hp) return UVIT_ERR_ARG;
    if (hp->sched_dev && (hp->sched_len < 1 || hp->sched_index < 0 || hp->sched_index >= hp->sched_len)) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Layout& lo = e->lo;
    // (end of round 4, measured and removed: summing each block's share of the clip norm behind its wgrad on the second stream, which leaves
    //  5 MB instead of 345 MB for this pass -- the 12 streaming launches take CUs from the dgrad chain: 23.76 vs 23.54 ms per step, 4 rotated pairs)
    HIPCHECK(hipMemsetAsync(e->sumsq, 0, sizeof(double), s));
    CHECK(uvit_sumsq_launch(e->buf.grads, lo.n_live, e->sumsq, s));
    const float gs = hp->grad_scale > 0.f ? hp->grad_scale : 1.0f;
    CHECK(uvit_adamw_launch(e->buf.params, e->buf.grads, e->buf.adam_m, e->buf.adam_v, e->buf.params_bf16, lo.n_live, lo.n_decay,
                            hp->lr, hp->weight_decay, hp->beta1, hp->beta2, hp->eps, hp->opt_step, e->sumsq, hp->clip_grad, gs,
                            e->gnorm, s, e->loss, (hp->do_ema || hp->sched_dev) ? e->buf.ema : nullptr, e->buf.ema_bf16, hp->ema_decay, e->poisoned,
                            hp->sched_dev, hp->sched_len, hp->sched_index));
    // (round 4 tried these transposed copies on the second stream, beside the next forward: 24.9-25.2 vs 24.7-25.2 ms per step, no gain; again at the
    //  end of the round with the order-rotated A/B, five pairs: 23.47 vs 23.43 ms)
    CHECK(uvit_transpose_batch_launch(e->tdesc, e->n_tdesc, e->n_ttiles, s));
    // frozen tensors (two-stream cov_qkv.weight) sit behind n_live: AdamW never touches them, the reference's EMA does
    // average them (ModelEmaV2 walks every state-dict value) -- a no-op on values that never change, skipped
    return UVIT_OK;
}

extern "C" int uvit_train_step(uvit_engine* e, const float* images, const int64_t* mask, const uvit_step_params* hp,
                               uvit_stream stream) {
    CHECK(uvit_step_begin(e, images, mask, hp, stream));
    for (int l = e->cfg.depth - 1; l >= 0; --l) CHECK(uvit_step_backward_layer(e, l, hp, stream));
    CHECK(uvit_step_backward_embed(e, stream));
    return uvit_step_update(e, hp, stream);
}

extern "C" int uvit_engine_read_stats_async(uvit_engine* e, float* host_out2, uvit_stream stream) {
    if (!e || !host_out2) return UVIT_ERR_ARG;
    HIPCHECK(hipMemcpyAsync(host_out2, e->loss, 8 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return UVIT_OK;
}

extern "C" int uvit_engine_read_stats(uvit_engine* e, float* host_out2, uvit_stream stream) {
    if (!e || !host_out2) return UVIT_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    HIPCHECK(hipMemcpyAsync(host_out2, e->loss, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHECK(hipStreamSynchronize(s));
    return UVIT_OK;
}

// ------------------------------------------------------------------------------------------
// operator-level C ABI (thin wrappers over the internal launchers)
// ------------------------------------------------------------------------------------------
#define S(x) ((hipStream_t)(x))
static GemmEpi epi_from_abi(const uvit_gemm_epilogue* ep) {
    GemmEpi g; g.out = ep->out; g.out2 = ep->out2; g.bias = ep->bias; g.bias2 = ep->bias2; g.gamma = ep->gamma;
    g.resid = ep->resid; g.rowscale = ep->rowscale; g.aux = ep->aux; g.mask = ep->mask; g.mask_token = ep->mask_token;
    g.ldo = ep->ldo; g.tokens = ep->tokens > 0 ? ep->tokens : 1; g.patches = ep->patches > 0 ? ep->patches : 1;
    g.row0 = ep->row0;
    return g;
}
extern "C" int uvit_op_gemm_nt_tuned(int mode, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                                     const uvit_gemm_epilogue* ep, const uvit_tuning* tune, int* tail_rows, uvit_stream st) {
    if (!A || !W || !ep || !ep->out) return UVIT_ERR_ARG;
    GemmTune tu;
    const int trc = tune_from_abi(tune, tu);
    if (trc) return trc;
    GemmEpi g = epi_from_abi(ep);
    return uvit_gemm_nt_launch(mode, A, W, M, N, K, lda, ldw, &g, S(st), &tu, tail_rows);
}
extern "C" int uvit_op_gemm_nt(int mode, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                               const uvit_gemm_epilogue* ep, uvit_stream st) {
    return uvit_op_gemm_nt_tuned(mode, A, W, M, N, K, lda, ldw, ep, nullptr, nullptr, st);
}
extern "C" int uvit_op_gemm_nt_sched(int mode, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                                     const uvit_gemm_epilogue* ep, const uvit_tuning* tune, uint32_t* tile_counters, uvit_stream st) {
    if (!ep || !A || !W || !ep->out) return UVIT_ERR_ARG;
    GemmTune tu;
    const int trc = tune_from_abi(tune, tu);
    if (trc) return trc;
    GemmEpi g = epi_from_abi(ep);
    g.tile_counter = tile_counters;
    return uvit_gemm_nt_launch(mode, A, W, M, N, K, lda, ldw, &g, S(st), &tu, nullptr);
}
extern "C" int uvit_op_wgrad_group(const uvit_wgrad_problem* problems, int count, const uvit_tuning* tune, uvit_stream st) {
    if (!problems || count < 1 || count > UVIT_TN_GROUP_MAX) return UVIT_ERR_ARG;
    GemmTune tu;
    const int trc = tune_from_abi(tune, tu);
    if (trc) return trc;
    TnProb pr[UVIT_TN_GROUP_MAX];
    for (int i = 0; i < count; ++i) {
        const uvit_wgrad_problem& q = problems[i];
        if (!q.Y_bf16 || !q.X_bf16 || !q.C) return UVIT_ERR_ARG;
        pr[i].Y = q.Y_bf16; pr[i].X = q.X_bf16; pr[i].C = q.C; pr[i].bias = q.bias; pr[i].bias2 = q.bias2;
        pr[i].bias_end = q.bias_end; pr[i].bias2_begin = q.bias2_begin;
        pr[i].M = q.M; pr[i].Nn = q.N; pr[i].Kk = q.K; pr[i].ldy = q.ldy; pr[i].ldx = q.ldx; pr[i].ldc = q.ldc;
    }
    return uvit_gemm_tn_group_launch(pr, count, S(st), &tu);
}

extern "C" int uvit_op_gemm_tn(const void* Y, const void* X, int M, int N, int K, int ldy, int ldx, float* C, int ldc,
                               const uvit_tuning* tune, uvit_stream st) {
    if (!Y || !X || !C) return UVIT_ERR_ARG;
    GemmTune tu;
    const int trc = tune_from_abi(tune, tu);
    if (trc) return trc;
    if (hipMemsetAsync(C, 0, (size_t)N * ldc * sizeof(float), S(st)) != hipSuccess) return UVIT_ERR_LAUNCH;
    return uvit_gemm_tn_launch(Y, X, M, N, K, ldy, ldx, C, ldc, 1, S(st), &tu);
}
extern "C" int uvit_op_attn_fwd(const void* qkv, const float* biasP, void* out, float* lse, int B, int H, int N, int NP,
                                float scale, float p_drop, uint32_t seed, uint32_t layer, uvit_stream st) {
    if (!qkv || !out || !lse) return UVIT_ERR_ARG;
    return uvit_attn_fwd_launch(qkv, biasP, out, lse, B, H, N, NP, scale, p_drop, seed, layer, S(st));
}
extern "C" int64_t uvit_op_attn_bwd_ws_bytes(int B, int H, int N) {
    if (B < 1 || H < 1 || N < 1 || N > 208) return UVIT_ERR_SHAPE;
    return (int64_t)uvit_attn_bwd_fused_ws_bytes(B, H, N);
}
extern "C" int uvit_op_attn_bwd(const void* qkv, const void* o_fwd, const void* d_o, const float* biasP, const float* lse,
                                      float* delta, void* dqkv, float* slab, int acc, void* ds_ws, int B, int H, int N, int NP,
                                      float scale, float p_drop, uint32_t seed, uint32_t layer, uvit_stream st) {
    if (!qkv || !o_fwd || !d_o || !lse || !delta || !dqkv || (slab && !ds_ws)) return UVIT_ERR_ARG;
    CHECK(uvit_attn_bwd_fused_launch(qkv, o_fwd, d_o, biasP, lse, delta, dqkv, ds_ws, slab != nullptr, B, H, N, NP, scale, p_drop, seed, layer, S(st)));
    if (slab) CHECK(uvit_attn_dbias_reduce_launch(ds_ws, slab, acc, B, H, N, NP, S(st)));
    return UVIT_OK;
}
extern "C" int uvit_op_attn2_fwd(const void* qkv_m, const void* qkv_c, const float* biasP, void* out_m, void* out_c, float* lse, int B,
                                 int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, uvit_stream st) {
    if (!qkv_m || !qkv_c || !out_m || !out_c || !lse) return UVIT_ERR_ARG;
    return uvit_attn2_fwd_launch(qkv_m, qkv_c, biasP, out_m, out_c, lse, B, H, N, NP, scale, p_drop, seed, layer, S(st));
}
extern "C" int64_t uvit_op_attn2_bwd_ws_bytes(int B, int H, int N) {
    if (B <= 0 || H <= 0 || N <= 0) return UVIT_ERR_SHAPE;
    return (int64_t)uvit_attn2_bwd_ws_bytes(B, H, N);
}
extern "C" int uvit_op_attn2_bwd(const void* qkv_m, const void* qkv_c, const void* o_m, const void* o_c, const void* d_m, const void* d_c,
                                 const float* biasP, const float* lse, float* delta, void* dqkv_m, void* dqkv_c, float* slab, int acc,
                                 void* ds_ws, int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer,
                                 uvit_stream st) {
    if (!qkv_m || !qkv_c || !o_m || !o_c || !d_m || !d_c || !lse || !delta || !dqkv_m || !dqkv_c || (slab && !ds_ws)) return UVIT_ERR_ARG;
    CHECK(uvit_attn2_bwd_launch(qkv_m, qkv_c, o_m, o_c, d_m, d_c, biasP, lse, delta, dqkv_m, dqkv_c, ds_ws, slab != nullptr, B, H, N, NP,
                                scale, p_drop, seed, layer, S(st)));
    if (slab) CHECK(uvit_attn2_dbias_reduce_launch(ds_ws, slab, acc, B, H, N, NP, S(st)));
    return UVIT_OK;
}
extern "C" int uvit_op_relpos_gather(const float* t, const int32_t* idx, float* biasP, int H, int N, int NP, uvit_stream st) {
    return uvit_relpos_gather_launch(t, idx, biasP, H, N, NP, S(st));
}
extern "C" int uvit_op_relpos_scatter(const float* slab, int nslab, const int32_t* idx, float* dt, int H, int N, int NP, uvit_stream st) {
    return uvit_relpos_scatter_launch(slab, nslab, idx, dt, H, N, NP, S(st));
}
extern "C" int uvit_op_ln_fwd(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd, int M, int C,
                              float eps, uvit_stream st) { return uvit_ln_fwd_launch(x, w, b, y, mean, rstd, M, C, eps, S(st)); }
extern "C" int uvit_op_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* w,
                              const float* dres, float* dx, float* dw, float* db, int M, int C, uvit_stream st) {
    return uvit_ln_bwd_launch(dy, x, mean, rstd, w, dres, dx, dw, db, M, C, 1, 0, S(st));
}
extern "C" int uvit_op_ema(float* ema, const float* p, void* eb, int64_t n, float d, uvit_stream st) {
    return uvit_ema_launch(ema, p, eb, (size_t)n, d, S(st));
}
extern "C" int uvit_op_sumsq(const float* g, int64_t n, double* out, uvit_stream st) { return uvit_sumsq_launch(g, (size_t)n, out, S(st)); }
extern "C" int uvit_op_adamw(float* p, const float* g, float* m, float* v, void* pb, int64_t n, int64_t nd, float lr, float wd,
                             float b1, float b2, float eps, int step, const double* sumsq, float max_norm, float gs,
                             float* gn, uvit_stream st) {
    return uvit_adamw_launch(p, g, m, v, pb, (size_t)n, (size_t)nd, lr, wd, b1, b2, eps, step, sumsq, max_norm, gs, gn, S(st));
}
extern "C" int uvit_op_smooth_l1(const float* out, const float* target, const int32_t* count, float beta, int l2, float ls,
                                 float* loss, void* dout, int Mmax, int C, uvit_stream st) {
    return uvit_smooth_l1_launch(out, target, count, beta, l2, ls, loss, dout, Mmax, C, S(st));
}
extern "C" int uvit_op_wasserstein_loss(const float* om, const float* oc, const float* tm, const float* tc, const int32_t* count, float lam,
                                        float ls, float* scratch, float* loss, void* dm, void* dc, int Mmax, int C, uvit_stream st) {
    if (!om || !oc || !tm || !tc || !count || !scratch || !loss || !dm || !dc || Mmax < 1 || C < 1) return UVIT_ERR_ARG;
    return uvit_wasserstein_loss_launch(om, oc, tm, tc, count, lam, ls, scratch, loss, dm, dc, Mmax, C, S(st));
}
extern "C" int uvit_op_variance_loss(const float* out, const int32_t* count, float w, float margin, float ls, float* scratch, float* loss,
                                     float* std_out, void* dout, int Mmax, int C, uvit_stream st) {
    if (!out || !count || !scratch || !loss || !dout) return UVIT_ERR_ARG;
    return uvit_variance_loss_launch(out, count, w, margin, ls, scratch, loss, std_out, dout, Mmax, C, S(st));
}
extern "C" int uvit_op_target_accum(const float* x, const int32_t* rowidx, const int32_t* count, float* acc, int first, int Mmax,
                                    int C, float eps, uvit_stream st) {
    return uvit_target_accum_launch(x, rowidx, count, acc, first, Mmax, C, eps, S(st));
}
extern "C" int uvit_op_target_finalize(float* acc, const int32_t* count, int nl, int post_ln, int Mmax, int C, float eps, uvit_stream st) {
    return uvit_target_finalize_launch(acc, count, nl, post_ln, Mmax, C, eps, S(st));
}
extern "C" int uvit_op_mask_compact(const int64_t* mask, int32_t* rowidx, int32_t* count, int B, int P, uvit_stream st) {
    return uvit_mask_compact_launch(mask, rowidx, count, B, P, S(st));
}
extern "C" int uvit_op_synth_batch(float* images, int64_t* mask, int B, int chans, int img_size, int patches, int n_mask, uint32_t seed,
                                   uint32_t it, uvit_stream st) {
    if (!images && !mask) return UVIT_ERR_ARG;
    return uvit_synth_batch_launch(images, mask, B, chans, img_size, patches, n_mask, seed, it, S(st));
}
extern "C" int uvit_op_im2col(const float* img, void* cols, int B, int Cin, int S_, int p, uvit_stream st) {
    return uvit_im2col_launch(img, cols, B, Cin, S_, p, S(st));
}
extern "C" int uvit_op_droppath(float* sc, const float* rates, int depth, int B, uint32_t seed, uint32_t step, uvit_stream st) {
    return uvit_droppath_launch(sc, rates, depth, 2, B, seed, step, S(st));
}
extern "C" int uvit_op_cast_bf16(const float* src, void* dst, int64_t n, uvit_stream st) { return uvit_cast_bf16_launch(src, dst, (size_t)n, S(st)); }
