// LayerNorm forward/backward and the data2vec target builder, gfx950.  HBM-bound: one wave per
// row, float4 loads of the fp32 residual stream, row statistics by wave shuffles, bf16x4 stores.
//
// Reference: nn.LayerNorm(eps=1e-6) inside Block (modeling_finetune.py:290-299); target builder
// = affine-free F.layer_norm(eps=1e-5) per layer, mean over layers, optional post layer_norm,
// masked-row gather (engine_for_cyclical.py:92-122).
#include "common.h"
#include "uvit_internal.h"

#define LN_MAXV 8          // float4 per lane: C <= 2048
#define LN_WAVES 4

// NV = float4 per lane actually needed (ceil(C / 256)): a template parameter so register use and
// hence occupancy follow the real row length (NV = 3 for C = 768) instead of the maximum.
template <int NV> struct RowVec { float4 v[NV]; };

template <int NV>
__device__ __forceinline__ void load_row(RowVec<NV>& r, const float* x, int C, int lane) {
    const int nv = C >> 2;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        r.v[k] = i < nv ? ((const float4*)x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NV>
__device__ __forceinline__ void row_stats(const RowVec<NV>& r, int C, int lane, float eps, float& mean, float& rstd) {
    const int nv = C >> 2;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += r.v[k].x + r.v[k].y + r.v[k].z + r.v[k].w;
    mean = wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (lane + 64 * k < nv) {
            const float a = r.v[k].x - mean, b = r.v[k].y - mean, c = r.v[k].z - mean, d = r.v[k].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    rstd = rsqrtf(wave_sum(q) / C + eps);
}

// y = (x - mean) * rstd * w + b  ->  bf16
template <int NV>
__global__ __launch_bounds__(LN_WAVES * 64)
void ln_fwd_kernel(const float* __restrict__ x, const int* __restrict__ rowidx, const int* __restrict__ count,
                   const float* __restrict__ w, const float* __restrict__ b, bf16* __restrict__ y,
                   float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;
    const int n_valid = count ? *count : M;
    if (row >= n_valid) {     // padded compact rows stay zero
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nv) ((bf16x4*)(y + (size_t)row * C))[i] = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        }
        return;
    }
    const int src = rowidx ? rowidx[row] : row;
    RowVec<NV> r;
    load_row(r, x + (size_t)src * C, C, lane);
    float mean, rstd;
    row_stats(r, C, lane, eps, mean, rstd);
    if (lane == 0 && mean_o) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            const float4 ww = ((const float4*)w)[i], bb = ((const float4*)b)[i];
            bf16x4 o = {f2bf((r.v[k].x - mean) * rstd * ww.x + bb.x), f2bf((r.v[k].y - mean) * rstd * ww.y + bb.y),
                        f2bf((r.v[k].z - mean) * rstd * ww.z + bb.z), f2bf((r.v[k].w - mean) * rstd * ww.w + bb.w)};
            ((bf16x4*)(y + (size_t)row * C))[i] = o;
        }
    }
}

// dx = dres + rstd * (dy*w - mean(dy*w) - xhat * mean(dy*w*xhat));  dw += dy*xhat;  db += dy
// LS = true fuses the LayerScale + DropPath backward of the branch that consumes dx next (modeling_finetune.py:295-298):
// with e = dx * dp[row / tokens]:  dy_next = bf16(e * gamma),  dgamma += e * y_next,  dbias += dy_next  -- the fp32
// residual gradient is then not read a second time by a separate pass.
//
// Shape of the launch (round 2): ONE 8-wave workgroup per CU walks a contiguous slab of rows.
//   * every operand row of a wave's NEXT row (x, dy, dres, y_next: 9 KB) is requested before the current row's two wave
//     reductions, so 2 rows x 8 waves = 144 KB are in flight per CU and the reductions / stores of one row hide under the
//     loads of the next (the old kernel asked for dres / y_next only after the reductions);
//   * the column sums (dw, db, dgamma, dbias) are reduced across the 8 waves in LDS and leave as ONE atomic per column
//     per workgroup: ~250 x 4 x C atomics per launch instead of ~900 x 4 x C (the old 4-wave blocks), 8 adders per
//     replica address instead of 28 -- same-address float atomics run at a fraction of the streaming rate
//     (MI355X_MICROARCH.md, Global float atomics), and they were a large part of this kernel's 108 us.
struct LsNext {
    const bf16* y; const float* gamma; const float* rowscale; bf16* dy; float* dgamma; float* dbias; int tokens;
    const int* pos = nullptr;      // row-list kernel only: dy is COMPACT by this sample map (drop-path sample list of the branch), -1 = dropped
};
#define LNB_WAVES 8

template <int NV, bool LS>
struct LnbRow {                      // operands of one row, as loaded
    float4 x[NV], dres[NV];
    bf16x4 dy[NV], y[LS ? NV : 1];
    float mean, rstd, dp;
    int xr, yr;                      // residual-stream row; row of dy_next (-1: its branch dropped the sample)
};

template <int NV, bool LS>
__device__ __forceinline__ void lnb_load(LnbRow<NV, LS>& r, int row, const bf16* dy, const float* x, const int* rowidx,
                                         const float* mean_i, const float* rstd_i, const float* dres, const LsNext& ls,
                                         int C, int nv, int lane) {
    const int xr = rowidx ? rowidx[row] : row;
    r.xr = xr;
    r.mean = mean_i[row]; r.rstd = rstd_i[row];
    r.dp = 1.0f; r.yr = xr;
    if constexpr (LS) {
        if (ls.rowscale) r.dp = ls.rowscale[xr / ls.tokens];
        if (ls.pos) { const int smp = xr / ls.tokens, sl = ls.pos[smp]; r.yr = sl < 0 ? -1 : sl * ls.tokens + (xr - smp * ls.tokens); }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            r.x[k] = ((const float4*)(x + (size_t)xr * C))[i];
            r.dy[k] = ((const bf16x4*)(dy + (size_t)row * C))[i];
            r.dres[k] = dres ? ((const float4*)(dres + (size_t)xr * C))[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (LS) r.y[k] = ((const bf16x4*)(ls.y + (size_t)xr * C))[i];
        }
    }
}

template <int NV, bool LS>
__global__ __launch_bounds__(LNB_WAVES * 64)
void ln_bwd_kernel(const bf16* __restrict__ dy, const float* __restrict__ x, const int* __restrict__ rowidx,
                   const int* __restrict__ count, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                   const float* __restrict__ w, const float* __restrict__ dres, float* __restrict__ dx,
                   float* __restrict__ dw, float* __restrict__ db, int M, int C, int nrep, size_t rep_stride, LsNext ls,
                   int rows_per_block) {
    __shared__ float red[LNB_WAVES][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    const int n_valid = count ? min(*count, M) : M;
    float4 ww[NV], gm[LS ? NV : 1];
    RowVec<NV> aw, ab;
    RowVec<LS ? NV : 1> ag, ay;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        aw.v[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab.v[k] = aw.v[k];
        ww[k] = lane + 64 * k < nv ? ((const float4*)w)[lane + 64 * k] : aw.v[k];
    }
#pragma unroll
    for (int k = 0; k < (LS ? NV : 1); ++k) {
        ag.v[k] = make_float4(0.f, 0.f, 0.f, 0.f); ay.v[k] = ag.v[k]; gm[k] = ag.v[k];
        if constexpr (LS) { if (lane + 64 * k < nv) gm[k] = ((const float4*)ls.gamma)[lane + 64 * k]; }
    }
    const int row_end = min((int)(blockIdx.x + 1) * rows_per_block, n_valid);
    int row = blockIdx.x * rows_per_block + wave;
    LnbRow<NV, LS> cur, nxt;
    if (row < row_end) lnb_load<NV, LS>(cur, row, dy, x, rowidx, mean_i, rstd_i, dres, ls, C, nv, lane);
    for (; row < row_end; row += LNB_WAVES) {
        const bool more = row + LNB_WAVES < row_end;
        if (more) lnb_load<NV, LS>(nxt, row + LNB_WAVES, dy, x, rowidx, mean_i, rstd_i, dres, ls, C, nv, lane);
        const float mean = cur.mean, rstd = cur.rstd;
        float4 g[NV], h[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (lane + 64 * k < nv) {
                const float d0 = bf2f(cur.dy[k][0]), d1 = bf2f(cur.dy[k][1]), d2 = bf2f(cur.dy[k][2]), d3 = bf2f(cur.dy[k][3]);
                h[k] = make_float4((cur.x[k].x - mean) * rstd, (cur.x[k].y - mean) * rstd, (cur.x[k].z - mean) * rstd,
                                   (cur.x[k].w - mean) * rstd);
                aw.v[k].x += d0 * h[k].x; aw.v[k].y += d1 * h[k].y; aw.v[k].z += d2 * h[k].z; aw.v[k].w += d3 * h[k].w;
                ab.v[k].x += d0; ab.v[k].y += d1; ab.v[k].z += d2; ab.v[k].w += d3;
                g[k] = make_float4(d0 * ww[k].x, d1 * ww[k].y, d2 * ww[k].z, d3 * ww[k].w);
                s1 += g[k].x + g[k].y + g[k].z + g[k].w;
                s2 += g[k].x * h[k].x + g[k].y * h[k].y + g[k].z * h[k].z + g[k].w * h[k].w;
            } else {
                g[k] = make_float4(0.f, 0.f, 0.f, 0.f); h[k] = g[k];
            }
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nv) {
                const float4 o = make_float4(cur.dres[k].x + rstd * (g[k].x - s1 - h[k].x * s2), cur.dres[k].y + rstd * (g[k].y - s1 - h[k].y * s2),
                                             cur.dres[k].z + rstd * (g[k].z - s1 - h[k].z * s2), cur.dres[k].w + rstd * (g[k].w - s1 - h[k].w * s2));
                ((float4*)(dx + (size_t)cur.xr * C))[i] = o;
                if constexpr (LS) if (cur.yr >= 0) {
                    const float e0 = o.x * cur.dp, e1 = o.y * cur.dp, e2 = o.z * cur.dp, e3 = o.w * cur.dp;
                    ag.v[k].x += e0 * bf2f(cur.y[k][0]); ag.v[k].y += e1 * bf2f(cur.y[k][1]);
                    ag.v[k].z += e2 * bf2f(cur.y[k][2]); ag.v[k].w += e3 * bf2f(cur.y[k][3]);
                    const bf16x4 ob = {f2bf(e0 * gm[k].x), f2bf(e1 * gm[k].y), f2bf(e2 * gm[k].z), f2bf(e3 * gm[k].w)};
                    ((bf16x4*)(ls.dy + (size_t)cur.yr * C))[i] = ob;
                    ay.v[k].x += bf2f(ob[0]); ay.v[k].y += bf2f(ob[1]); ay.v[k].z += bf2f(ob[2]); ay.v[k].w += bf2f(ob[3]);
                }
            }
        }
        if (more) cur = nxt;
    }
    // cross-wave reduction of the column partials in LDS, then one atomic per column per workgroup, into replica
    // (block % nrep) of the accumulators (summed once per step)
    const size_t rep = (size_t)(blockIdx.x % nrep) * rep_stride;
    auto fold = [&](const float4& part, float* dst, int k) {
        __syncthreads();
        ((float4*)red[wave])[lane] = part;
        __syncthreads();
        // 8 waves x 256 floats -> wave q folds floats [32 q, 32 q + 32) of the 256 (lanes 0..31), 8 partial rows each
        if (lane < 32) {
            const int col = wave * 32 + lane;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < LNB_WAVES; ++q) sum += red[q][col];
            const int c = 256 * k + col;                 // column of the row: float4 index (lane' + 64 k) * 4 + component
            if (c < C) atomicAdd(dst + rep + c, sum);
        }
    };
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (64 * k < nv) {
            fold(aw.v[k], dw, k);
            fold(ab.v[k], db, k);
            if constexpr (LS) { fold(ag.v[k], ls.dgamma, k); fold(ay.v[k], ls.dbias, k); }
        }
    }
}

// ---- drop-path sample lists (round 4): the same two kernels over a DENSE walk of the residual stream with COMPACT branch buffers ----
// pos[b] = compact slot of sample b, -1 when the branch dropped it.  Forward: a kept row is normalised into compact row slot * tokens + t;
// a dropped row is copied to the branch's output stream (x + 0 * branch = x), which the residual epilogue of the branch's last GEMM then
// does not touch.
template <int NV>
__global__ __launch_bounds__(LN_WAVES * 64)
void ln_fwd_keep_kernel(const float* __restrict__ x, const int* __restrict__ pos, const float* __restrict__ w,
                        const float* __restrict__ b, bf16* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                        float* __restrict__ xcopy, int M, int C, float eps, int tokens) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = C >> 2;
    const int smp = row / tokens, slot = pos[smp];
    RowVec<NV> r;
    load_row(r, x + (size_t)row * C, C, lane);
    if (slot < 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) if (lane + 64 * k < nv) ((float4*)(xcopy + (size_t)row * C))[lane + 64 * k] = r.v[k];
        return;
    }
    const size_t crow = (size_t)slot * tokens + (row - smp * tokens);
    float mean, rstd;
    row_stats(r, C, lane, eps, mean, rstd);
    if (lane == 0) { mean_o[crow] = mean; rstd_o[crow] = rstd; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            const float4 ww = ((const float4*)w)[i], bb = ((const float4*)b)[i];
            bf16x4 o = {f2bf((r.v[k].x - mean) * rstd * ww.x + bb.x), f2bf((r.v[k].y - mean) * rstd * ww.y + bb.y),
                        f2bf((r.v[k].z - mean) * rstd * ww.z + bb.z), f2bf((r.v[k].w - mean) * rstd * ww.w + bb.w)};
            ((bf16x4*)(y + crow * C))[i] = o;
        }
    }
}

// Backward: rows are walked densely.  posA maps the sample to the compact slot of the LayerNorm's own branch (dy, mean, rstd compact;
// nullptr = every sample kept, dense): a dropped sample's row passes dres through (dx = dres) and adds nothing to dw / db.  posB maps it
// to the slot of the branch whose LayerScale backward rides along (dy_next compact, y_next dense-indexed; nullptr = dense; ls.dy ==
// nullptr: no such branch): a dropped sample writes nothing and adds nothing to dgamma / dbias.  The pad rows of dy_next (cntB .. next
// multiple of 64 of pad_base + cntB: the wgrad's reduction length; pad_base = rows in front of dy_next in a stacked buffer) are zero-filled
// by the last workgroup when cntB is given.
template <int NV>
struct LnkRow {
    float4 x[NV], dres[NV];
    bf16x4 dy[NV], y[NV];
    float mean, rstd, dp;
    int ca, cb;                     // compact rows of the two branches, -1 = dropped
};

template <int NV>
__device__ __forceinline__ void lnk_load(LnkRow<NV>& r, int row, const bf16* dy, const float* x, const int* posA, const int* posB,
                                         const float* mean_i, const float* rstd_i, const float* dres, const LsNext& ls,
                                         int C, int nv, int lane) {
    const int smp = row / ls.tokens, t = row - smp * ls.tokens;
    const int sa = posA ? posA[smp] : smp, sb = ls.dy ? (posB ? posB[smp] : smp) : -1;
    r.ca = sa < 0 ? -1 : sa * ls.tokens + t;
    r.cb = sb < 0 ? -1 : sb * ls.tokens + t;
    r.mean = 0.f; r.rstd = 0.f; r.dp = 1.0f;
    if (r.ca >= 0) { r.mean = mean_i[r.ca]; r.rstd = rstd_i[r.ca]; }
    if (r.cb >= 0 && ls.rowscale) r.dp = ls.rowscale[smp];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            r.dres[k] = ((const float4*)(dres + (size_t)row * C))[i];
            if (r.ca >= 0) {
                r.x[k] = ((const float4*)(x + (size_t)row * C))[i];
                r.dy[k] = ((const bf16x4*)(dy + (size_t)r.ca * C))[i];
            }
            if (r.cb >= 0) r.y[k] = ((const bf16x4*)(ls.y + (size_t)row * C))[i];
        }
    }
}

template <int NV>
__global__ __launch_bounds__(LNB_WAVES * 64)
void ln_bwd_keep_kernel(const bf16* __restrict__ dy, const float* __restrict__ x, const int* __restrict__ posA,
                        const int* __restrict__ posB, const int* __restrict__ cntB, const float* __restrict__ mean_i,
                        const float* __restrict__ rstd_i, const float* __restrict__ w, const float* __restrict__ dres,
                        float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int M, int C, int nrep,
                        size_t rep_stride, LsNext ls, int rows_per_block, int pad_base, bf16* __restrict__ pad2, int pad2_cols) {
    __shared__ float red[LNB_WAVES][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    if (ls.dy && cntB && blockIdx.x == gridDim.x - 1) {
        const int n = *cntB, npad = ((pad_base + n + 63) & ~63) - pad_base;
        for (int r = n + wave; r < npad; r += LNB_WAVES) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (lane + 64 * k < nv) ((bf16x4*)(ls.dy + (size_t)r * C))[lane + 64 * k] = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
            // a second buffer with the same row list (the attention branch's dqkv, written by the attention backward for the kept samples only)
            if (pad2) for (int c = lane * 4; c < pad2_cols; c += 256) *(bf16x4*)(pad2 + (size_t)r * pad2_cols + c) = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        }
    }
    float4 ww[NV], gm[NV];
    RowVec<NV> aw, ab, ag, ay;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        aw.v[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab.v[k] = aw.v[k]; ag.v[k] = aw.v[k]; ay.v[k] = aw.v[k]; gm[k] = aw.v[k];
        ww[k] = lane + 64 * k < nv ? ((const float4*)w)[lane + 64 * k] : aw.v[k];
        if (ls.dy && lane + 64 * k < nv) gm[k] = ((const float4*)ls.gamma)[lane + 64 * k];
    }
    const int row_end = min((int)(blockIdx.x + 1) * rows_per_block, M);
    int row = blockIdx.x * rows_per_block + wave;
    LnkRow<NV> cur, nxt;
    if (row < row_end) lnk_load<NV>(cur, row, dy, x, posA, posB, mean_i, rstd_i, dres, ls, C, nv, lane);
    for (; row < row_end; row += LNB_WAVES) {
        const bool more = row + LNB_WAVES < row_end;
        if (more) lnk_load<NV>(nxt, row + LNB_WAVES, dy, x, posA, posB, mean_i, rstd_i, dres, ls, C, nv, lane);
        float4 o[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) o[k] = cur.dres[k];
        if (cur.ca >= 0) {
            const float mean = cur.mean, rstd = cur.rstd;
            float4 g[NV], h[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (lane + 64 * k < nv) {
                    const float d0 = bf2f(cur.dy[k][0]), d1 = bf2f(cur.dy[k][1]), d2 = bf2f(cur.dy[k][2]), d3 = bf2f(cur.dy[k][3]);
                    h[k] = make_float4((cur.x[k].x - mean) * rstd, (cur.x[k].y - mean) * rstd, (cur.x[k].z - mean) * rstd,
                                       (cur.x[k].w - mean) * rstd);
                    aw.v[k].x += d0 * h[k].x; aw.v[k].y += d1 * h[k].y; aw.v[k].z += d2 * h[k].z; aw.v[k].w += d3 * h[k].w;
                    ab.v[k].x += d0; ab.v[k].y += d1; ab.v[k].z += d2; ab.v[k].w += d3;
                    g[k] = make_float4(d0 * ww[k].x, d1 * ww[k].y, d2 * ww[k].z, d3 * ww[k].w);
                    s1 += g[k].x + g[k].y + g[k].z + g[k].w;
                    s2 += g[k].x * h[k].x + g[k].y * h[k].y + g[k].z * h[k].z + g[k].w * h[k].w;
                } else {
                    g[k] = make_float4(0.f, 0.f, 0.f, 0.f); h[k] = g[k];
                }
            }
            s1 = wave_sum(s1) / C;
            s2 = wave_sum(s2) / C;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                o[k] = make_float4(cur.dres[k].x + rstd * (g[k].x - s1 - h[k].x * s2), cur.dres[k].y + rstd * (g[k].y - s1 - h[k].y * s2),
                                   cur.dres[k].z + rstd * (g[k].z - s1 - h[k].z * s2), cur.dres[k].w + rstd * (g[k].w - s1 - h[k].w * s2));
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nv) {
                ((float4*)(dx + (size_t)row * C))[i] = o[k];
                if (cur.cb >= 0) {
                    const float e0 = o[k].x * cur.dp, e1 = o[k].y * cur.dp, e2 = o[k].z * cur.dp, e3 = o[k].w * cur.dp;
                    ag.v[k].x += e0 * bf2f(cur.y[k][0]); ag.v[k].y += e1 * bf2f(cur.y[k][1]);
                    ag.v[k].z += e2 * bf2f(cur.y[k][2]); ag.v[k].w += e3 * bf2f(cur.y[k][3]);
                    const bf16x4 ob = {f2bf(e0 * gm[k].x), f2bf(e1 * gm[k].y), f2bf(e2 * gm[k].z), f2bf(e3 * gm[k].w)};
                    ((bf16x4*)(ls.dy + (size_t)cur.cb * C))[i] = ob;
                    ay.v[k].x += bf2f(ob[0]); ay.v[k].y += bf2f(ob[1]); ay.v[k].z += bf2f(ob[2]); ay.v[k].w += bf2f(ob[3]);
                }
            }
        }
        if (more) cur = nxt;
    }
    const size_t rep = (size_t)(blockIdx.x % nrep) * rep_stride;
    auto fold = [&](const float4& part, float* dst, int k) {
        __syncthreads();
        ((float4*)red[wave])[lane] = part;
        __syncthreads();
        if (lane < 32) {
            const int col = wave * 32 + lane;
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < LNB_WAVES; ++q) sum += red[q][col];
            const int c = 256 * k + col;
            if (c < C) atomicAdd(dst + rep + c, sum);
        }
    };
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (64 * k < nv) {
            fold(aw.v[k], dw, k);
            fold(ab.v[k], db, k);
            if (ls.dy) { fold(ag.v[k], ls.dgamma, k); fold(ay.v[k], ls.dbias, k); }
        }
    }
}

// acc[i] (+)= layer_norm(x[rowidx[i]] - sub[rowidx[i]])  (no affine; sub == nullptr: 0).  `sub` = the stream before the
// MLP branch: x - sub is the block's `fc` output, the `--layer_results fc` target (modeling_cyclical.py:199-205).
template <int NV>
__global__ __launch_bounds__(LN_WAVES * 64)
void target_accum_kernel(const float* __restrict__ x, const float* __restrict__ sub, const int* __restrict__ rowidx,
                         const int* __restrict__ count, float* __restrict__ acc, int first, int Mmax, int C, float eps, int ln) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= Mmax) return;
    const int nv = C >> 2;
    float4* dst = (float4*)(acc + (size_t)row * C);
    if (row >= *count) {
        if (first)
#pragma unroll
            for (int k = 0; k < NV; ++k) if (lane + 64 * k < nv) dst[lane + 64 * k] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    RowVec<NV> r;
    load_row(r, x + (size_t)rowidx[row] * C, C, lane);
    if (sub) {
        RowVec<NV> q;
        load_row(q, sub + (size_t)rowidx[row] * C, C, lane);
#pragma unroll
        for (int k = 0; k < NV; ++k) { r.v[k].x -= q.v[k].x; r.v[k].y -= q.v[k].y; r.v[k].z -= q.v[k].z; r.v[k].w -= q.v[k].w; }
    }
    float mean = 0.f, rstd = 1.f;
    if (ln) row_stats(r, C, lane, eps, mean, rstd);          // ln == 0 (--no_target_layer_norm_last): the rows are summed as they are
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv) {
            float4 o = make_float4((r.v[k].x - mean) * rstd, (r.v[k].y - mean) * rstd, (r.v[k].z - mean) * rstd,
                                   (r.v[k].w - mean) * rstd);
            if (!first) { const float4 a = dst[i]; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
            dst[i] = o;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(LN_WAVES * 64)
void target_finalize_kernel(float* __restrict__ acc, const int* __restrict__ count, float inv_layers, int post_ln,
                            int Mmax, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    if (row >= Mmax || row >= *count) return;
    const int nv = C >> 2;
    RowVec<NV> r;
    load_row(r, acc + (size_t)row * C, C, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) { r.v[k].x *= inv_layers; r.v[k].y *= inv_layers; r.v[k].z *= inv_layers; r.v[k].w *= inv_layers; }
    float mean = 0.f, rstd = 1.f;
    if (post_ln) row_stats(r, C, lane, eps, mean, rstd);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        if (i < nv)
            ((float4*)(acc + (size_t)row * C))[i] = make_float4((r.v[k].x - mean) * rstd, (r.v[k].y - mean) * rstd,
                                                                 (r.v[k].z - mean) * rstd, (r.v[k].w - mean) * rstd);
    }
}

#define LN_DISPATCH2(KERNEL, FLAG, C, ...) do { const int _nv = ((C) + 255) / 256; \
    if (_nv <= 1) hipLaunchKernelGGL((KERNEL<1, FLAG>), __VA_ARGS__); else if (_nv == 2) hipLaunchKernelGGL((KERNEL<2, FLAG>), __VA_ARGS__); \
    else if (_nv == 3) hipLaunchKernelGGL((KERNEL<3, FLAG>), __VA_ARGS__); else if (_nv == 4) hipLaunchKernelGGL((KERNEL<4, FLAG>), __VA_ARGS__); \
    else if (_nv == 5) hipLaunchKernelGGL((KERNEL<5, FLAG>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<8, FLAG>), __VA_ARGS__); } while (0)
#define LN_DISPATCH(KERNEL, C, ...) do { const int _nv = ((C) + 255) / 256; \
    if (_nv <= 1) hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); else if (_nv == 2) hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); \
    else if (_nv == 3) hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); else if (_nv == 4) hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); \
    else if (_nv == 5) hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__); else hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); } while (0)

// Slabs per CU.  Round 3 tried 2 (512 workgroups that could rebalance when some CUs are held by RCCL channel workgroups): the step was
// slower with all CUs (25.4 -> 25.7 ms) AND with 240 / 224 CUs masked in (28.2 -> 28.8, 28.7 -> 29.5 ms; tools/cu_mask_bench.sh), so 1 stays.
#define LNB_BLOCKS_PER_CU 1
static int lnb_rows(int M, int resident_blocks_per_cu) {
    // rows per workgroup so that the grid is `resident_blocks_per_cu` balanced workgroups per CU
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const int target = ncu * resident_blocks_per_cu;
    int rows = (M + target - 1) / target;
    rows = ((rows + LNB_WAVES - 1) / LNB_WAVES) * LNB_WAVES;      // every wave of a block walks the same number of rows
    return rows < LNB_WAVES ? LNB_WAVES : rows;
}

static int ln_shape_ok(int M, int C) { return (M > 0 && C > 0 && (C % 4) == 0 && C <= LN_MAXV * 256) ? UVIT_OK : UVIT_ERR_SHAPE; }

int uvit_ln_fwd_launch(const float* x, const float* w, const float* b, void* y, float* mean, float* rstd, int M, int C,
                       float eps, hipStream_t s) {
    if (ln_shape_ok(M, C)) return UVIT_ERR_SHAPE;
    LN_DISPATCH(ln_fwd_kernel, C, dim3((M + LN_WAVES - 1) / LN_WAVES), dim3(LN_WAVES * 64), 0, s, x, (const int*)nullptr,
                       (const int*)nullptr, w, b, (bf16*)y, mean, rstd, M, C, eps);
    return uvit_check_launch();
}
int uvit_ln_fwd_gather_launch(const float* x, const int* rowidx, const int* count, const float* w, const float* b,
                              void* y, float* mean, float* rstd, int Mmax, int C, float eps, hipStream_t s) {
    if (ln_shape_ok(Mmax, C)) return UVIT_ERR_SHAPE;
    LN_DISPATCH(ln_fwd_kernel, C, dim3((Mmax + LN_WAVES - 1) / LN_WAVES), dim3(LN_WAVES * 64), 0, s, x, rowidx, count,
                       w, b, (bf16*)y, mean, rstd, Mmax, C, eps);
    return uvit_check_launch();
}
int uvit_ln_bwd_launch(const void* dy, const float* x, const float* mean, const float* rstd, const float* w,
                       const float* dres, float* dx, float* dw, float* db, int M, int C, int nrep, size_t rep_stride, hipStream_t s) {
    if (ln_shape_ok(M, C)) return UVIT_ERR_SHAPE;
    const int rpb = lnb_rows(M, LNB_BLOCKS_PER_CU);
    LN_DISPATCH2(ln_bwd_kernel, false, C, dim3((M + rpb - 1) / rpb), dim3(LNB_WAVES * 64), 0, s, (const bf16*)dy, x,
                       (const int*)nullptr, (const int*)nullptr, mean, rstd, w, dres, dx, dw, db, M, C, nrep > 0 ? nrep : 1, rep_stride, LsNext{}, rpb);
    return uvit_check_launch();
}
int uvit_ln_bwd_ls_launch(const void* dy, const float* x, const float* mean, const float* rstd, const float* w,
                          const float* dres, float* dx, float* dw, float* db, const void* y_next, const float* gamma_next,
                          const float* rowscale_next, void* dy_next, float* dgamma_next, float* dbias_next, int tokens,
                          int M, int C, int nrep, size_t rep_stride, hipStream_t s, const int* rowidx, const int* count, const int* pos_next) {
    if (ln_shape_ok(M, C) || tokens <= 0 || (rowidx && !count)) return UVIT_ERR_SHAPE;
    const LsNext ls{(const bf16*)y_next, gamma_next, rowscale_next, (bf16*)dy_next, dgamma_next, dbias_next, tokens, pos_next};
    const int rpb = lnb_rows(M, LNB_BLOCKS_PER_CU);
    LN_DISPATCH2(ln_bwd_kernel, true, C, dim3((M + rpb - 1) / rpb), dim3(LNB_WAVES * 64), 0, s, (const bf16*)dy, x,
                       rowidx, count, mean, rstd, w, dres, dx, dw, db, M, C, nrep > 0 ? nrep : 1, rep_stride, ls, rpb);
    return uvit_check_launch();
}
int uvit_ln_fwd_keep_launch(const float* x, const int* pos, const float* w, const float* b, void* y, float* mean, float* rstd,
                            float* xcopy, int M, int C, int tokens, float eps, hipStream_t s) {
    if (ln_shape_ok(M, C) || tokens <= 0 || (M % tokens) || !pos || !xcopy || !mean || !rstd) return UVIT_ERR_SHAPE;
    LN_DISPATCH(ln_fwd_keep_kernel, C, dim3((M + LN_WAVES - 1) / LN_WAVES), dim3(LN_WAVES * 64), 0, s, x, pos, w, b, (bf16*)y, mean, rstd,
                       xcopy, M, C, eps, tokens);
    return uvit_check_launch();
}
int uvit_ln_bwd_keep_launch(const void* dy, const float* x, const int* posA, const float* mean, const float* rstd, const float* w,
                            const float* dres, float* dx, float* dw, float* db, const void* y_next, const float* gamma_next,
                            const float* rowscale_next, void* dy_next, float* dgamma_next, float* dbias_next, const int* posB,
                            const int* cntB, int tokens, int M, int C, int nrep, size_t rep_stride, hipStream_t s, int pad_base,
                            void* pad2, int pad2_cols) {
    if (ln_shape_ok(M, C) || tokens <= 0 || (M % tokens) || !dres || ((posB || cntB) && !dy_next) || pad_base < 0 ||
        (pad2 && (!cntB || pad2_cols <= 0 || (pad2_cols % 4)))) return UVIT_ERR_SHAPE;
    const LsNext ls{(const bf16*)y_next, gamma_next, rowscale_next, (bf16*)dy_next, dgamma_next, dbias_next, tokens};
    const int rpb = lnb_rows(M, LNB_BLOCKS_PER_CU);
    LN_DISPATCH(ln_bwd_keep_kernel, C, dim3((M + rpb - 1) / rpb), dim3(LNB_WAVES * 64), 0, s, (const bf16*)dy, x, posA, posB, cntB, mean, rstd,
                       w, dres, dx, dw, db, M, C, nrep > 0 ? nrep : 1, rep_stride, ls, rpb, pad_base, (bf16*)pad2, pad2_cols);
    return uvit_check_launch();
}
int uvit_ln_bwd_scatter_launch(const void* dy, const float* x, const int* rowidx, const int* count, const float* mean,
                               const float* rstd, const float* w, float* dx, float* dw, float* db, int Mmax, int C,
                               int nrep, size_t rep_stride, hipStream_t s) {
    if (ln_shape_ok(Mmax, C)) return UVIT_ERR_SHAPE;
    const int rpb = lnb_rows(Mmax, 1);
    LN_DISPATCH2(ln_bwd_kernel, false, C, dim3((Mmax + rpb - 1) / rpb), dim3(LNB_WAVES * 64), 0, s, (const bf16*)dy, x,
                       rowidx, count, mean, rstd, w, (const float*)nullptr, dx, dw, db, Mmax, C, nrep > 0 ? nrep : 1, rep_stride, LsNext{}, rpb);
    return uvit_check_launch();
}
int uvit_target_accum_launch(const float* x, const int* rowidx, const int* count, float* acc, int first, int Mmax,
                             int C, float eps, hipStream_t s, const float* sub, int ln) {
    if (ln_shape_ok(Mmax, C)) return UVIT_ERR_SHAPE;
    LN_DISPATCH(target_accum_kernel, C, dim3((Mmax + LN_WAVES - 1) / LN_WAVES), dim3(LN_WAVES * 64), 0, s, x, sub, rowidx,
                       count, acc, first, Mmax, C, eps, ln);
    return uvit_check_launch();
}
int uvit_target_finalize_launch(float* acc, const int* count, int n_layers, int post_ln, int Mmax, int C, float eps,
                                hipStream_t s) {
    if (ln_shape_ok(Mmax, C) || n_layers <= 0) return UVIT_ERR_SHAPE;
    LN_DISPATCH(target_finalize_kernel, C, dim3((Mmax + LN_WAVES - 1) / LN_WAVES), dim3(LN_WAVES * 64), 0, s, acc,
                       count, 1.0f / n_layers, post_ln, Mmax, C, eps);
    return uvit_check_launch();
}

// grads[i] += sum_r rep[r][i]  (replicated column-sum accumulators -> gradient arena), once per step
__global__ __launch_bounds__(256)
void reduce_replicas_kernel(const float* __restrict__ rep, float* __restrict__ out, size_t n4, int nrep, size_t stride4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ((const float4*)out)[i];
        for (int r = 0; r < nrep; ++r) {
            const float4 b = ((const float4*)rep)[(size_t)r * stride4 + i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        ((float4*)out)[i] = a;
    }
}
int uvit_reduce_replicas_launch(const float* rep, float* out, size_t n, int nrep, size_t stride, hipStream_t s) {
    if (n % 4 || stride % 4) return UVIT_ERR_SHAPE;
    size_t g = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(reduce_replicas_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, s, rep, out, n / 4, nrep, stride / 4);
    return uvit_check_launch();
}

// ------------------------------------------------------------------------------------------
// Dense target builder for the batch- / instance-norm target variants (engine_for_cyclical.py:94-118; flags off in every
// BASELINE config).  Those normalise over ALL tokens, so the masked-rows-only builder above cannot serve them: each
// target layer is gathered into a dense [B*P, C] buffer (cls dropped), normalised per channel over (B, T) ("batch") and /
// or per (sample, channel) over T ("instance"; both affine-free, biased variance, eps 1e-5), then accumulated.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void gather_patch_rows_kernel(const float* __restrict__ x, const float* __restrict__ sub, float* __restrict__ v, int B, int P, int C4) {
    const size_t total = (size_t)B * P * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / C4; const int c = (int)(i - row * C4);
        const int b = (int)(row / P), p_ = (int)(row - (size_t)b * P);
        const size_t src = ((size_t)b * (P + 1) + 1 + p_) * C4 + c;
        float4 a = ((const float4*)x)[src];
        if (sub) { const float4 q = ((const float4*)sub)[src]; a.x -= q.x; a.y -= q.y; a.z -= q.z; a.w -= q.w; }
        ((float4*)v)[i] = a;
    }
}

// one workgroup per (group of `rows` consecutive rows, 64 channels): v <- (v - mean_c) * rsqrt(var_c + eps), statistics over
// the group's rows (biased variance, accumulated in fp64)
__global__ __launch_bounds__(256)
void colnorm_kernel(float* __restrict__ v, int rows, int C, float eps) {
    __shared__ double red[2][4][64];
    __shared__ float stat[2][64];
    const int ch = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float* base = v + (size_t)blockIdx.y * rows * C;
    double s1 = 0.0, s2 = 0.0;
    if (ch < C)
        for (int r = ph; r < rows; r += 4) { const double a = base[(size_t)r * C + ch]; s1 += a; s2 += a * a; }
    red[0][ph][threadIdx.x & 63] = s1; red[1][ph][threadIdx.x & 63] = s2;
    __syncthreads();
    if (ph == 0) {
        const int l = threadIdx.x;
        const double a = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        const double q = red[1][0][l] + red[1][1][l] + red[1][2][l] + red[1][3][l];
        const double mean = a / rows, var = q / rows - mean * mean;
        stat[0][l] = (float)mean; stat[1][l] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
    __syncthreads();
    if (ch < C) {
        const float mean = stat[0][threadIdx.x & 63], rstd = stat[1][threadIdx.x & 63];
        for (int r = ph; r < rows; r += 4) { float* q = base + (size_t)r * C + ch; *q = (*q - mean) * rstd; }
    }
}

__global__ __launch_bounds__(256)
void axpy_rows_kernel(float* __restrict__ acc, const float* __restrict__ v, int first, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ((const float4*)v)[i];
        if (!first) { const float4 b = ((const float4*)acc)[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        ((float4*)acc)[i] = a;
    }
}

// out[i] = dense[patch row of token row rowidx[i]]  (rowidx holds b * (P + 1) + 1 + p in mask order)
__global__ __launch_bounds__(256)
void gather_masked_rows_kernel(const float* __restrict__ dense, const int* __restrict__ rowidx, const int* __restrict__ count,
                               float* __restrict__ out, int Mmax, int P, int C4) {
    const int n = min(*count, Mmax);
    const size_t total = (size_t)n * C4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C4; const int c = (int)(i - r * C4);
        const int tok = rowidx[r], b = tok / (P + 1), p_ = tok - b * (P + 1) - 1;
        ((float4*)out)[i] = ((const float4*)dense)[((size_t)b * P + p_) * C4 + c];
    }
}

static unsigned dense_grid(size_t n) { size_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

int uvit_gather_patch_rows_launch(const float* x, const float* sub, float* v, int B, int P, int C, hipStream_t s) {
    if (C % 4) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(gather_patch_rows_kernel, dim3(dense_grid((size_t)B * P * (C / 4))), dim3(256), 0, s, x, sub, v, B, P, C / 4);
    return uvit_check_launch();
}
int uvit_colnorm_launch(float* v, int groups, int rows, int C, float eps, hipStream_t s) {
    if (groups < 1 || rows < 1) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(colnorm_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, s, v, rows, C, eps);
    return uvit_check_launch();
}
int uvit_axpy_rows_launch(float* acc, const float* v, int first, size_t n, hipStream_t s) {
    if (n % 4) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(axpy_rows_kernel, dim3(dense_grid(n / 4)), dim3(256), 0, s, acc, v, first, n / 4);
    return uvit_check_launch();
}
int uvit_gather_masked_rows_launch(const float* dense, const int* rowidx, const int* count, float* out, int Mmax, int P, int C,
                                   hipStream_t s) {
    if (C % 4) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(gather_masked_rows_kernel, dim3(dense_grid((size_t)Mmax * (C / 4))), dim3(256), 0, s, dense, rowidx, count, out,
                       Mmax, P, C / 4);
    return uvit_check_launch();
}
