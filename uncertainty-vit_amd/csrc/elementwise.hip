// HBM-bound helper kernels of the ViT step (gfx950): patch re-indexing, token assembly and its
// backward, mask compaction, relative-position bias gather / gradient scatter, LayerScale
// backward, column sums (bias gradients), the fused SmoothL1 loss forward+backward, batched
// weight transposes and the drop-path multipliers.
#include "common.h"
#include "uvit_internal.h"

// ------------------------------------------------------------------------------------------
// images (B,Cin,S,S) f32 -> cols (B*P, Cin*p*p) bf16, k = c*p*p + i*p + j   (conv == GEMM,
// modeling_finetune.py:317-325).  One thread moves 8 pixels of one patch row.
// ------------------------------------------------------------------------------------------
__global__ void im2col_kernel(const float* __restrict__ img, bf16* __restrict__ cols, int B, int Cin, int S, int p) {
    const int g = S / p, P = g * g, K = Cin * p * p, kc_per_row = K / 8;
    const size_t total = (size_t)B * P * kc_per_row;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int kc = idx % kc_per_row;
        const size_t row = idx / kc_per_row;
        const int b = row / P, pp = row % P, py = pp / g, px = pp % g;
        const int k = kc * 8, c = k / (p * p), i = (k % (p * p)) / p, j = k % p;
        const float* src = img + (((size_t)b * Cin + c) * S + py * p + i) * S + px * p + j;
        const float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
        bf16x8 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w), f2bf(d.x), f2bf(d.y), f2bf(d.z), f2bf(d.w)};
        *(bf16x8*)(cols + row * K + k) = o;
    }
}

// ------------------------------------------------------------------------------------------
// mask (B*P int64 0/1) -> ordered list of masked token rows b*(P+1)+1+p and their count.
// Single workgroup; 25k elements.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024)
void mask_compact_kernel(const int64_t* __restrict__ mask, int* __restrict__ rowidx, int* __restrict__ count, int B, int P) {
    // (round 4: was one strided scan per thread + a 10-step LDS scan, 95 us at the head of every step)  The flags of up to 32 chunks of 1024
    // consecutive elements are fetched first, coalesced and all in flight (bit k of a thread = element 1024 k + tid); each chunk is then placed by a
    // wave ballot plus the 16 wave totals through LDS -- the list stays in element order.
    __shared__ int s_w[16];
    const int n = B * P, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int sc = 0; sc < n; sc += 32 * 1024) {
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int i = sc + k * 1024 + tid;
            if (i < n && mask[i] != 0) bits |= 1u << k;
        }
        for (int k = 0; k < 32 && sc + k * 1024 < n; ++k) {
            const bool keep = (bits >> k) & 1u;
            const unsigned long long bal = __ballot(keep);
            if (lane == 0) s_w[wave] = __popcll(bal);
            __syncthreads();
            int before = base, tot = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int v = s_w[q]; tot += v; if (q < wave) before += v; }
            if (keep) {
                const int i = sc + k * 1024 + tid, b = i / P;
                rowidx[before + __popcll(bal & ((1ull << lane) - 1ull))] = b * (P + 1) + 1 + (i - b * P);
            }
            base += tot;
            __syncthreads();
        }
    }
    if (tid == 0) *count = base;
    // unused tail entries point at a valid row so gathers stay in bounds
    for (int i = base + tid; i < n; i += 1024) rowidx[i] = 0;
}

__global__ void set_cls_kernel(float* __restrict__ x, const float* __restrict__ cls, int B, int N, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * C) { const int b = i / C, c = i - b * C; x[(size_t)b * N * C + c] = cls[c]; }
}

// biasP[h][q][k] = table[index[q*N+k]][h] * log2(e)  (modeling_finetune.py:359-364) in the padded NP x NP layout
// the attention kernels read: padded key columns hold -1e30 (vanish in the softmax), padded query rows 0.
// table == nullptr (no relative position bias) gives the pure padding mask.
__global__ void relpos_gather_kernel(const float* __restrict__ table, const int* __restrict__ index,
                                     float* __restrict__ biasP, int H, int N, int NP) {
    const size_t total = (size_t)H * NP * NP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = i % NP, q = (i / NP) % NP, h = i / ((size_t)NP * NP);
        float v = 0.f;
        if (k >= N) v = -1e30f;
        else if (q < N && table) v = table[(size_t)index[q * N + k] * H + h] * 1.4426950408889634f;
        biasP[i] = v;
    }
}

// dtable[index[q*N+k]][h] += sum_slabs slab[s][h][k][q]   (slabs hold dS^T summed over batch chunks and layers)
__global__ void relpos_scatter_kernel(const float* __restrict__ slab, int nslab, const int* __restrict__ index,
                                      float* __restrict__ dtable, int H, int N, int NP, int ntable) {
    extern __shared__ float tsum[];
    const int h = blockIdx.x;
    for (int i = threadIdx.x; i < ntable; i += blockDim.x) tsum[i] = 0.f;
    __syncthreads();
    const int total = N * N;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < total; i += gridDim.y * blockDim.x) {
        const int k = i / N, q = i - k * N;
        float s = 0.f;
        for (int sl = 0; sl < nslab; ++sl) s += slab[(((size_t)sl * H + h) * NP + k) * NP + q];
        atomicAdd(&tsum[index[q * N + k]], s);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ntable; i += blockDim.x)
        if (tsum[i] != 0.f) atomicAdd(dtable + (size_t)i * H + h, tsum[i]);
}

// ------------------------------------------------------------------------------------------
// column-partial machinery: 4 waves x 64 lanes, each lane owns float4 column groups
// ------------------------------------------------------------------------------------------
#define CP_MAXV 8
#define CP_ROWS 32

template <int NV, int WAVES = 4>
__device__ __forceinline__ void block_col_atomic(float4 (&acc)[NV], float* out, int nv, float (*red)[64 * 4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (64 * k < nv) {
            __syncthreads();
            ((float4*)red[wave])[lane] = acc[k];
            __syncthreads();
            if (wave == 0) {
                float4 s = ((float4*)red[0])[lane];
#pragma unroll
                for (int q = 1; q < WAVES; ++q) { const float4 a = ((float4*)red[q])[lane]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
                const int i = lane + 64 * k;
                if (i < nv) { atomicAdd(out + 4 * i, s.x); atomicAdd(out + 4 * i + 1, s.y); atomicAdd(out + 4 * i + 2, s.z); atomicAdd(out + 4 * i + 3, s.w); }
            }
        }
    }
}

// LayerScale + DropPath backward (modeling_finetune.py:295-298):
//   dy = dx * gamma * dp[b]   (bf16, gradient of the Linear output y)
//   dgamma += sum_m dx * dp * y ;  dbias += sum_m dy
template <int NV>
__global__ __launch_bounds__(256)
void ls_bwd_kernel(const float* __restrict__ dx, const bf16* __restrict__ y, const float* __restrict__ gamma,
                   const float* __restrict__ rowscale, bf16* __restrict__ dy, float* __restrict__ dgamma,
                   float* __restrict__ dbias, int M, int C, int tokens, int nrep, size_t rep_stride,
                   const int* __restrict__ rowidx, const int* __restrict__ count) {
    // rowidx != nullptr: COMPACT output -- row r of dy belongs to residual-stream row rowidx[r] (dx, y and the drop-path sample are taken
    // there); rows r >= *count are padding and get zeros
    __shared__ float red[4][64 * 4];
    dgamma += (size_t)(blockIdx.x % nrep) * rep_stride; dbias += (size_t)(blockIdx.x % nrep) * rep_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = C >> 2;
    float4 ag[NV], ab[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { ag[k] = make_float4(0.f, 0.f, 0.f, 0.f); ab[k] = ag[k]; }
    const int row_end = min((int)(blockIdx.x + 1) * CP_ROWS, M);
    const int n_valid = rowidx ? min(*count, M) : M;
    for (int row = blockIdx.x * CP_ROWS + wave; row < row_end; row += 4) {
        if (row >= n_valid) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (lane + 64 * k < nv) ((bf16x4*)(dy + (size_t)row * C))[lane + 64 * k] = bf16x4{f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
            continue;
        }
        const int xr = rowidx ? rowidx[row] : row;
        const float dp = rowscale ? rowscale[xr / tokens] : 1.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nv) {
                const float4 d = ((const float4*)(dx + (size_t)xr * C))[i];
                const float4 g = ((const float4*)gamma)[i];
                const bf16x4 yy = ((const bf16x4*)(y + (size_t)xr * C))[i];
                const float e0 = d.x * dp, e1 = d.y * dp, e2 = d.z * dp, e3 = d.w * dp;
                ag[k].x += e0 * bf2f(yy[0]); ag[k].y += e1 * bf2f(yy[1]); ag[k].z += e2 * bf2f(yy[2]); ag[k].w += e3 * bf2f(yy[3]);
                bf16x4 o = {f2bf(e0 * g.x), f2bf(e1 * g.y), f2bf(e2 * g.z), f2bf(e3 * g.w)};
                ((bf16x4*)(dy + (size_t)row * C))[i] = o;
                ab[k].x += bf2f(o[0]); ab[k].y += bf2f(o[1]); ab[k].z += bf2f(o[2]); ab[k].w += bf2f(o[3]);
            }
        }
    }
    block_col_atomic(ag, dgamma, nv, red);
    block_col_atomic(ab, dbias, nv, red);
}

// out[c] += sum_m y[m][col0 + c]: lane = 8 columns (16 B), wave = 512 columns, 4 waves walk rows 4-way unrolled
#define CS_ROWS 256
__global__ __launch_bounds__(256)
void colsum_kernel(const bf16* __restrict__ y, int ld, int col0, int ncols, int M, float* __restrict__ out, int nrep, size_t rep_stride) {
    __shared__ float red[4][64 * 8];
    out += (size_t)(blockIdx.y % nrep) * rep_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 512 + lane * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int row_end = min((int)(blockIdx.y + 1) * CS_ROWS, M);
    if (c < ncols) {
        const bf16* base = y + col0 + c;
        int row = blockIdx.y * CS_ROWS + wave;
        for (; row + 12 < row_end; row += 16) {
            const bf16x8 v0 = *(const bf16x8*)(base + (size_t)row * ld), v1 = *(const bf16x8*)(base + (size_t)(row + 4) * ld);
            const bf16x8 v2 = *(const bf16x8*)(base + (size_t)(row + 8) * ld), v3 = *(const bf16x8*)(base + (size_t)(row + 12) * ld);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (bf2f(v0[j]) + bf2f(v1[j])) + (bf2f(v2[j]) + bf2f(v3[j]));
        }
        for (; row < row_end; row += 4) {
            const bf16x8 v = *(const bf16x8*)(base + (size_t)row * ld);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int cc = blockIdx.x * 512 + i;
        if (cc < ncols) atomicAdd(out + cc, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
    }
}

// ------------------------------------------------------------------------------------------
// loss = mean smooth_l1(out, target; beta) over count*C elements (engine_for_cyclical.py:147-163),
// fused with its gradient (bf16), rows >= count produce zero gradient.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void smooth_l1_kernel(const float* __restrict__ out, const float* __restrict__ target, const int* __restrict__ count,
                      float beta, int l2, float loss_scale, float* __restrict__ loss, bf16* __restrict__ dout,
                      int Mmax, int C) {
    __shared__ float red[4];
    const int n_valid = min(*count, Mmax);
    const size_t nv = (size_t)Mmax * C / 4, valid = (size_t)n_valid * C / 4;
    const float inv = loss_scale / ((float)n_valid * (float)C);
    float part = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        bf16x4 g = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
        if (i < valid) {
            const float4 o = ((const float4*)out)[i], t = ((const float4*)target)[i];
            const float d[4] = {o.x - t.x, o.y - t.y, o.z - t.z, o.w - t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = fabsf(d[j]);
                float l, gr;
                if (l2) { l = d[j] * d[j]; gr = 2.0f * d[j]; }
                else if (a < beta) { l = 0.5f * d[j] * d[j] / beta; gr = d[j] / beta; }
                else { l = a - 0.5f * beta; gr = d[j] > 0.f ? 1.0f : (d[j] < 0.f ? -1.0f : 0.f); }
                part += l;
                g[j] = f2bf(gr * inv);
            }
        }
        ((bf16x4*)dout)[i] = g;
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// ------------------------------------------------------------------------------------------
// token-assembly backward (modeling_cyclical.py:179-192):  dpatch = (1-w) * dx[:,1:],
// dcls += sum_b dx[b,0], dmask_token += sum_masked dx
// ------------------------------------------------------------------------------------------
template <int NV, int TKB_WAVES>
__global__ __launch_bounds__(TKB_WAVES * 64)
void token_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ mask, bf16* __restrict__ dpatch,
                      float* __restrict__ dcls, float* __restrict__ dmask, int B, int P, int C, int rows_per_block) {
    __shared__ float red[TKB_WAVES][64 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = C >> 2, N = P + 1;
    float4 ac[NV], am[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) { ac[k] = make_float4(0.f, 0.f, 0.f, 0.f); am[k] = ac[k]; }
    const int M = B * N;
    const int row_end = min((int)(blockIdx.x + 1) * rows_per_block, M);
    for (int row = blockIdx.x * rows_per_block + wave; row < row_end; row += TKB_WAVES) {
        const int b = row / N, t = row - b * N;
        const bool is_cls = t == 0;
        const bool masked = !is_cls && mask[b * P + t - 1] != 0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < nv) {
                const float4 d = ((const float4*)(dx + (size_t)row * C))[i];
                if (is_cls) { ac[k].x += d.x; ac[k].y += d.y; ac[k].z += d.z; ac[k].w += d.w; }
                else {
                    if (masked) { am[k].x += d.x; am[k].y += d.y; am[k].z += d.z; am[k].w += d.w; }
                    const float m = masked ? 0.f : 1.f;
                    bf16x4 o = {f2bf(d.x * m), f2bf(d.y * m), f2bf(d.z * m), f2bf(d.w * m)};
                    ((bf16x4*)(dpatch + ((size_t)b * P + t - 1) * C))[i] = o;
                }
            }
        }
    }
    // a workgroup whose rows hold no cls token has nothing to add to d cls_token (block-uniform test)
    const int row0 = blockIdx.x * rows_per_block;
    if ((row0 + N - 1) / N * N < row_end) block_col_atomic<NV, TKB_WAVES>(ac, dcls, nv, red);
    block_col_atomic<NV, TKB_WAVES>(am, dmask, nv, red);
}

// ------------------------------------------------------------------------------------------
// batched bf16 transposes (weights -> W^T copies for the dgrad GEMMs), 64x64 tiles through LDS
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void transpose_batch_kernel(const TransposeDesc* __restrict__ descs, int ndesc) {
    __shared__ bf16 tile[64][66];
    int lo = 0, hi = ndesc - 1;
    const int t = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (descs[mid].tile0 <= t) lo = mid; else hi = mid - 1; }
    const TransposeDesc d = descs[lo];
    const int local = t - d.tile0, tc = (d.cols + 63) / 64;
    const int tr0 = (local / tc) * 64, tc0 = (local % tc) * 64;
    if (tr0 >= d.rows) return;
    const bf16* src = (const bf16*)d.src; bf16* dst = (bf16*)d.dst;
    if (tr0 + 64 <= d.rows && tc0 + 64 <= d.cols && !(d.rows & 7) && !(d.cols & 7)) {
        // whole tile (every Linear weight of the models: dimensions are multiples of 64): 16 B per lane on both sides (end of round 4; the
        // element-wise form below moved 2 B per lane and ran at 2.5 TB/s: 139 us at the very end of every step)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = threadIdx.x + 256 * k, r = idx >> 3, c8 = (idx & 7) * 8;
            const bf16x8 v = *(const bf16x8*)(src + (size_t)(tr0 + r) * d.cols + tc0 + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) tile[r][c8 + j] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = threadIdx.x + 256 * k, oc = idx >> 3, r8 = (idx & 7) * 8;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = tile[r8 + j][oc];
            *(bf16x8*)(dst + (size_t)(tc0 + oc) * d.rows + tr0 + r8) = o;
        }
        return;
    }
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (tr0 + r < d.rows && tc0 + c < d.cols) ? src[(size_t)(tr0 + r) * d.cols + tc0 + c] : f2bf(0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (tr0 + r < d.rows && tc0 + c < d.cols) dst[(size_t)(tc0 + c) * d.rows + tr0 + r] = tile[r][c];
    }
}

// scales[(layer*nbr + branch)*B + b] (nbr = 2 draws per block, 4 for the two-stream model) : 1/keep or 0 (timm drop_path, modeling_finetune.py:51-62); rate 0 -> 1
__global__ void droppath_kernel(float* __restrict__ scales, const float* __restrict__ rates, int depth, int nbr, int B,
                                uint32_t seed, uint32_t step) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= depth * nbr * B) return;
    const int b = i % B, lb = i / B, layer = lb / nbr, br = lb - layer * nbr;
    const float r = rates[layer];
    float v = 1.0f;
    if (r > 0.f) {
        const uint32_t key = uvit_hash32(seed ^ ((step * (uint32_t)nbr * depth + (uint32_t)nbr * layer + br + 1u) * 0x9E3779B9u));
        v = uvit_hash32((uint32_t)b ^ key) >= uvit_drop_threshold(r) ? 1.0f / (1.0f - r) : 0.f;
    }
    scales[i] = v;
}

// Drop-path sample lists (round 4).  A branch whose sample drew scale 0 contributes exactly nothing to the forward (x + 0 * branch) and receives
// exactly no gradient (modeling_finetune.py:51-62, 295-298), so the step runs every branch on the KEPT samples only, in compact buffers.  One
// workgroup per (layer, branch) list turns the step's multipliers into:  pos[b] = compact slot of sample b or -1;  bmap[slot] = sample;
// rows[r] = residual-stream row of compact row r (slot * tokens + t -> bmap[slot] * tokens + t; -1 on the pad rows up to the next multiple of
// 64);  cnt = kept rows.  The host sized the launches from ITS evaluation of the same hash (engine.hip): a list whose count differs from the
// host's raises a flag behind the counts, which poisons the step's loss (droppath_lists_guard_kernel).
#define DP_MAX_LISTS 128                     // 2 x UVIT_MAX_DEPTH (include/uvit.h)
struct DpHostCounts { int k[DP_MAX_LISTS]; };
#define DPL_CHUNKS 8
__global__ __launch_bounds__(256)
void droppath_lists_kernel(const float* __restrict__ scales, int* __restrict__ pos, int* __restrict__ bmap, int* __restrict__ rows,
                           int* __restrict__ cnt, int B, int tokens, int rows_stride, DpHostCounts host) {
    // grid (lists, DPL_CHUNKS): every workgroup of a list derives the slot / sample maps (a ballot pass per 256 samples; the sample map stays in
    // LDS), the first one publishes them, and each fills its share of the row list
    extern __shared__ int s_m[];                      // [B] slot -> sample
    const int lb = blockIdx.x;
    const float* sc = scales + (size_t)lb * B;
    int* p = pos + (size_t)lb * B; int* m = bmap + (size_t)lb * B; int* r = rows + (size_t)lb * rows_stride;
    const bool pub = blockIdx.y == 0;
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const bool keep = b < B && sc[b] != 0.f;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int before = base;
        for (int q = 0; q < wave; ++q) before += s_w[q];
        const int slot = before + __popcll(bal & ((1ull << lane) - 1ull));
        if (pub && b < B) p[b] = keep ? slot : -1;
        if (keep) { s_m[slot] = b; if (pub) m[slot] = b; }
        base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (pub) {
        for (int b = base + threadIdx.x; b < B; b += 256) m[b] = 0;
        if (threadIdx.x == 0) {
            cnt[lb] = base * tokens;
            cnt[gridDim.x + lb] = base != host.k[lb];         // checked by droppath_lists_guard_kernel once the loss has been zeroed
        }
    }
    const int n = base * tokens, npad = (n + 63) & ~63;
    const int per = (npad + DPL_CHUNKS - 1) / DPL_CHUNKS, lo = blockIdx.y * per, hi = min(lo + per, npad);
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int slot = i / tokens;
        r[i] = i < n ? s_m[slot] * tokens + (i - slot * tokens) : -1;
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define CP_DISPATCH(KERNEL, C, ...) do { const int _nv = ((C) + 255) / 256; \
    if (_nv <= 1) hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); else if (_nv == 2) hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__); \
    else if (_nv == 3) hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); else if (_nv == 4) hipLaunchKernelGGL(KERNEL<4>, __VA_ARGS__); \
    else if (_nv == 5) hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__); else hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); } while (0)

#define LN_DISPATCH2_TKB(W) do { const int _nv = (C + 255) / 256; const dim3 g_((M + rpb - 1) / rpb), b_((W) * 64); \
    if (_nv <= 1) hipLaunchKernelGGL((token_bwd_kernel<1, W>), g_, b_, 0, s, dx, mask, (bf16*)dpatch, dcls, dmask_token, B, P, C, rpb); \
    else if (_nv == 2) hipLaunchKernelGGL((token_bwd_kernel<2, W>), g_, b_, 0, s, dx, mask, (bf16*)dpatch, dcls, dmask_token, B, P, C, rpb); \
    else if (_nv == 3) hipLaunchKernelGGL((token_bwd_kernel<3, W>), g_, b_, 0, s, dx, mask, (bf16*)dpatch, dcls, dmask_token, B, P, C, rpb); \
    else if (_nv == 4) hipLaunchKernelGGL((token_bwd_kernel<4, W>), g_, b_, 0, s, dx, mask, (bf16*)dpatch, dcls, dmask_token, B, P, C, rpb); \
    else if (_nv == 5) hipLaunchKernelGGL((token_bwd_kernel<5, W>), g_, b_, 0, s, dx, mask, (bf16*)dpatch, dcls, dmask_token, B, P, C, rpb); \
    else hipLaunchKernelGGL((token_bwd_kernel<8, W>), g_, b_, 0, s, dx, mask, (bf16*)dpatch, dcls, dmask_token, B, P, C, rpb); } while (0)

static inline int grid_for(size_t n, int block, int cap = 256 * 8) {
    size_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > (size_t)cap ? cap : g));
}

int uvit_im2col_launch(const float* img, void* cols, int B, int Cin, int S, int p, hipStream_t s) {
    if (B <= 0 || p % 8 || S % p || S % 4) return UVIT_ERR_SHAPE;
    const size_t total = (size_t)B * (S / p) * (S / p) * (Cin * p * p / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, img, (bf16*)cols, B, Cin, S, p);
    return uvit_check_launch();
}
int uvit_mask_compact_launch(const int64_t* mask, int* rowidx, int* count, int B, int P, hipStream_t s) {
    hipLaunchKernelGGL(mask_compact_kernel, dim3(1), dim3(1024), 0, s, mask, rowidx, count, B, P);
    return uvit_check_launch();
}
int uvit_set_cls_launch(float* x, const float* cls, const float* pos, int B, int N, int C, hipStream_t s) {
    if (pos) return UVIT_ERR_ARG;   // absolute position embedding: not on the configured path (use_abs_pos_emb=False)
    hipLaunchKernelGGL(set_cls_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, x, cls, B, N, C);
    return uvit_check_launch();
}
int uvit_relpos_gather_launch(const float* table, const int* index, float* biasP, int H, int N, int NP, hipStream_t s) {
    hipLaunchKernelGGL(relpos_gather_kernel, dim3(grid_for((size_t)H * NP * NP, 256)), dim3(256), 0, s, table, index, biasP, H, N, NP);
    return uvit_check_launch();
}
int uvit_relpos_scatter_launch(const float* slab, int nslab, const int* index, float* dtable, int H, int N, int NP,
                               hipStream_t s) {
    const int g = (int)(sqrtf((float)(N - 1)) + 0.5f);
    const int ntable = (2 * g - 1) * (2 * g - 1) + 3;
    hipLaunchKernelGGL(relpos_scatter_kernel, dim3(H, 24), dim3(256), ntable * sizeof(float), s, slab, nslab, index, dtable, H, N, NP, ntable);
    return uvit_check_launch();
}
int uvit_ls_bwd_launch(const float* dx, const void* y, const float* gamma, const float* rowscale, void* dy,
                       float* dgamma, float* dbias, int M, int C, int tokens, int nrep, size_t rep_stride, hipStream_t s,
                       const int* rowidx, const int* count) {
    if (C % 4 || C > CP_MAXV * 256 || (rowidx && !count)) return UVIT_ERR_SHAPE;
    CP_DISPATCH(ls_bwd_kernel, C, dim3((M + CP_ROWS - 1) / CP_ROWS), dim3(256), 0, s, dx, (const bf16*)y, gamma, rowscale,
                       (bf16*)dy, dgamma, dbias, M, C, tokens, nrep > 0 ? nrep : 1, rep_stride, rowidx, count);
    return uvit_check_launch();
}
__global__ void rows_guard_kernel(const int* __restrict__ count, int limit, float* __restrict__ loss) {
    if (*count > limit) loss[0] = __builtin_nanf("");
}
int uvit_rows_guard_launch(const int* count, int limit, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(rows_guard_kernel, dim3(1), dim3(1), 0, s, count, limit, loss);
    return uvit_check_launch();
}
int uvit_colsum_launch(const void* y, int ld, int col0, int ncols, int M, float* out, int nrep, size_t rep_stride, hipStream_t s) {
    if (ncols % 8 || col0 % 8 || ld % 8) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(colsum_kernel, dim3((ncols + 511) / 512, (M + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, s, (const bf16*)y, ld, col0, ncols, M, out, nrep > 0 ? nrep : 1, rep_stride);
    return uvit_check_launch();
}
int uvit_smooth_l1_launch(const float* out, const float* target, const int* count, float beta, int l2, float loss_scale,
                          float* loss, void* dout, int Mmax, int C, hipStream_t s) {
    if (C % 4) return UVIT_ERR_SHAPE;
    // (<= 512 workgroups: every workgroup ends with one atomic on the same loss word)
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(grid_for((size_t)Mmax * C / 4, 256, 512)), dim3(256), 0, s, out, target, count, beta, l2,
                       loss_scale, loss, (bf16*)dout, Mmax, C);
    return uvit_check_launch();
}
int uvit_token_bwd_launch(const float* dx, const int64_t* mask, void* dpatch, float* dcls, float* dmask_token, int B,
                          int P, int C, hipStream_t s) {
    if (C % 4 || C > CP_MAXV * 256) return UVIT_ERR_SHAPE;
    // every workgroup ends with 2 x C same-address atomics (d cls_token, d mask_token have ONE accumulator each): with
    // 32-row workgroups (788 of them at bs = 128) those contended atomics were most of the kernel's 169 us.  One
    // workgroup per CU-slot instead: <= 512 workgroups.
    const int M = B * (P + 1);
    // (end of round 4: 16-wave workgroups, <= 128 of them -- the same rows in flight as 512 4-wave ones with a quarter of the same-address adders,
    //  and a workgroup without a cls row skips that column sum: 111 -> 43 us alone; inside the two-stream step, where it shares the CUs with the
    //  last wgrad, the step time does not change)
    int rpb = (M + 127) / 128;
    rpb = ((rpb + 15) / 16) * 16;
    if (rpb < 32) rpb = 32;
    LN_DISPATCH2_TKB(16);
    return uvit_check_launch();
}
int uvit_transpose_batch_launch(const void* descs_dev, int ndesc, int total_tiles, hipStream_t s) {
    if (ndesc <= 0 || total_tiles <= 0) return UVIT_OK;
    hipLaunchKernelGGL(transpose_batch_kernel, dim3(total_tiles), dim3(256), 0, s, (const TransposeDesc*)descs_dev, ndesc);
    return uvit_check_launch();
}
int uvit_droppath_launch(float* scales, const float* rates_dev, int depth, int nbr, int B, uint32_t seed, uint32_t step, hipStream_t s) {
    hipLaunchKernelGGL(droppath_kernel, dim3((depth * nbr * B + 255) / 256), dim3(256), 0, s, scales, rates_dev, depth, nbr, B, seed, step);
    return uvit_check_launch();
}

__global__ void droppath_lists_guard_kernel(const int* __restrict__ flags, int n, float* __restrict__ loss) {
    int bad = 0;
    for (int i = 0; i < n; ++i) bad |= flags[i];
    if (bad) loss[0] = __builtin_nanf("");
}
int uvit_droppath_lists_guard_launch(const int* cnt, int nlists, float* loss, hipStream_t s) {
    hipLaunchKernelGGL(droppath_lists_guard_kernel, dim3(1), dim3(1), 0, s, cnt + nlists, nlists, loss);
    return uvit_check_launch();
}
int uvit_droppath_lists_launch(const float* scales, int* pos, int* bmap, int* rows, int* cnt, int nlists, int B, int tokens,
                               int rows_stride, const int* host_counts, hipStream_t s) {
    if (nlists < 1 || nlists > DP_MAX_LISTS || rows_stride < ((B * tokens + 63) & ~63)) return UVIT_ERR_ARG;
    DpHostCounts h;
    for (int i = 0; i < DP_MAX_LISTS; ++i) h.k[i] = i < nlists ? host_counts[i] : 0;
    hipLaunchKernelGGL(droppath_lists_kernel, dim3(nlists, DPL_CHUNKS), dim3(256), (size_t)B * sizeof(int), s, scales, pos, bmap, rows, cnt, B, tokens, rows_stride, h);
    return uvit_check_launch();
}

// A rank whose loss is not finite puts a NaN into its gradient arena BEFORE the all-reduce: the SUM carries it to every
// rank, so all ranks see a non-finite gradient norm, skip AdamW / EMA (optim.hip) and stop together (the reference's
// rank-local exit at engine_for_cyclical.py:166-168 would leave the peers blocked in the next collective).
__global__ void poison_kernel(const float* __restrict__ loss, float* __restrict__ dst, int* __restrict__ sticky) {
    if ((__float_as_uint(*loss) & 0x7F800000u) == 0x7F800000u) { *dst = __int_as_float(0x7FC00000); if (sticky) *sticky = 1; }
}
int uvit_poison_if_nonfinite_launch(const float* loss, float* dst, int* sticky, hipStream_t s) {
    hipLaunchKernelGGL(poison_kernel, dim3(1), dim3(1), 0, s, loss, dst, sticky);
    return uvit_check_launch();
}

// ------------------------------------------------------------------------------------------
// WassersteinLoss forward + backward (distloss.py:13-30, 73-79):
//   pos_r = sum (sig(o)-sig(t))^2 + sum (sqrt(sig(co)) - sqrt(sig(ct)))^2 ; u = pos / max(pos)
//   loss  = lambda * sum_r softplus(u_r) / max_r softplus(u_r)
// max_r u_r is exactly 1, so the second normaliser is the constant softplus(1) and carries no gradient;
// the first one routes -sum_s g'(u_s) u_s / m to the arg-max row (autograd of `x / x.abs().max()`).
// scratch: [0] max pos (float bits) [1] sum softplus [2] sum g' u [3] argmax row (int) ; pos[] from scratch+16
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sig_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256)
void wl_pos_kernel(const float* __restrict__ om, const float* __restrict__ oc, const float* __restrict__ tm,
                   const float* __restrict__ tc, const int* __restrict__ count, float* __restrict__ scratch, int Mmax, int C) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= min(*count, Mmax)) return;
    float acc = 0.f;
    for (int i = lane; i < C; i += 64) {
        const size_t o = (size_t)row * C + i;
        const float d = sig_(om[o]) - sig_(tm[o]);
        const float e = sqrtf(fmaxf(sig_(oc[o]), 1e-24f)) - sqrtf(fmaxf(sig_(tc[o]), 1e-24f));
        acc += d * d + e * e;
    }
    acc = wave_sum(acc);
    if (lane == 0) { scratch[16 + row] = acc; atomicMax((int*)scratch, __float_as_int(acc)); }
}

__global__ __launch_bounds__(256)
void wl_sum_kernel(const int* __restrict__ count, float* __restrict__ scratch, int Mmax) {
    const int n = min(*count, Mmax);
    const float m = scratch[0];
    float s1 = 0.f, s2 = 0.f;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const float pos = scratch[16 + r], u = pos / m;
        s1 += log1pf(__expf(u));                       // softplus(u) = -log(sigmoid(-u))
        s2 += sig_(u) * u;
        if (pos == m) atomicMin((int*)scratch + 3, r);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(scratch + 1, s1); atomicAdd(scratch + 2, s2); }
}

__global__ __launch_bounds__(256)
void wl_grad_kernel(const float* __restrict__ om, const float* __restrict__ oc, const float* __restrict__ tm,
                    const float* __restrict__ tc, const int* __restrict__ count, const float* __restrict__ scratch,
                    float lam, float* __restrict__ loss, bf16* __restrict__ dout_m, bf16* __restrict__ dout_c, int Mmax, int C) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(*count, Mmax);
    const float F = 1.3132616875182228f;                // softplus(1)
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(loss, lam * scratch[1] / F);
    if (row >= Mmax) return;
    if (row >= n) {
        for (int i = lane; i < C; i += 64) dout_c[(size_t)row * C + i] = f2bf(0.f);
        return;
    }
    const float m = scratch[0], u = scratch[16 + row] / m;
    float coef = sig_(u);
    if (row == ((const int*)scratch)[3]) coef -= scratch[2];
    coef *= lam / (F * m);
    for (int i = lane; i < C; i += 64) {
        const size_t o = (size_t)row * C + i;
        const float a = sig_(om[o]), c1 = sig_(oc[o]), c2 = sig_(tc[o]);
        const float gm = coef * 2.0f * (a - sig_(tm[o])) * a * (1.0f - a);
        const float s1 = sqrtf(fmaxf(c1, 1e-24f)), s2 = sqrtf(fmaxf(c2, 1e-24f));
        const float gc = coef * (s1 - s2) / s1 * c1 * (1.0f - c1);
        dout_m[o] = f2bf(bf2f(dout_m[o]) + gm);         // on top of the SmoothL1 gradient already there
        dout_c[o] = f2bf(gc);
    }
}

int uvit_wasserstein_loss_launch(const float* out_m, const float* out_c, const float* tgt_m, const float* tgt_c, const int* count,
                                 float lam, float loss_scale, float* scratch, float* loss, void* dout_m, void* dout_c,
                                 int Mmax, int C, hipStream_t s) {
    if (hipMemsetAsync(scratch, 0, 16 * sizeof(float), s) != hipSuccess) return UVIT_ERR_LAUNCH;
    const int big = 0x7FFFFFFF;
    (void)big;
    hipLaunchKernelGGL(wl_pos_kernel, dim3((Mmax + 3) / 4), dim3(256), 0, s, out_m, out_c, tgt_m, tgt_c, count, scratch, Mmax, C);
    // argmax slot starts at +inf (as int): set after the memset via a 4-byte memset pattern is not possible -> small fill
    if (hipMemsetD32Async((hipDeviceptr_t)((int*)scratch + 3), 0x7FFFFFFF, 1, s) != hipSuccess) return UVIT_ERR_LAUNCH;
    hipLaunchKernelGGL(wl_sum_kernel, dim3(64), dim3(256), 0, s, count, scratch, Mmax);
    hipLaunchKernelGGL(wl_grad_kernel, dim3((Mmax + 3) / 4), dim3(256), 0, s, out_m, out_c, tgt_m, tgt_c, count, scratch,
                       lam * loss_scale, loss, (bf16*)dout_m, (bf16*)dout_c, Mmax, C);
    return uvit_check_launch();
}

// ------------------------------------------------------------------------------------------
// On-device synthetic batch: the loader contract of datasets.py:110-118 produced where it is
// consumed -- images (B, Cin, S, S) f32 ~ N(0, 1) (post-Normalize statistics) and bool_masked_pos (B, P) int64 with
// EXACTLY n_mask ones per image (a uniformly random subset) -- so a benchmark / soak run needs no host loader and no
// PCIe traffic.  Counter-based: element i of step `it` depends only on (seed, it, i).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void synth_images_kernel(float* __restrict__ img, size_t n2, uint32_t key) {
    // two normals per pair of uniforms (Box-Muller); n2 = number of float2 pairs
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t a = uvit_hash32((uint32_t)(2 * i) ^ key), b = uvit_hash32((uint32_t)(2 * i + 1) ^ key ^ (uint32_t)(i >> 31));
        const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1)
        const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = sqrtf(-2.0f * __logf(u1));
        float sn, cs;
        __sincosf(6.283185307179586f * u2, &sn, &cs);
        ((float2*)img)[i] = make_float2(r * cs, r * sn);
    }
}

// one workgroup per image: patch i is masked when its key ranks among the n_mask smallest of the image's P keys
__global__ __launch_bounds__(256)
void synth_mask_kernel(int64_t* __restrict__ mask, int P, int n_mask, uint32_t key) {
    extern __shared__ uint32_t keys[];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = uvit_hash32((uint32_t)(b * P + i) ^ key);
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const uint32_t k = keys[i];
        int rank = 0;
        for (int j = 0; j < P; ++j) rank += (keys[j] < k) || (keys[j] == k && j < i);
        mask[(size_t)b * P + i] = rank < n_mask ? 1 : 0;
    }
}

int uvit_synth_batch_launch(float* images, int64_t* mask, int B, int chans, int img_size, int patches, int n_mask, uint32_t seed,
                            uint32_t it, hipStream_t s) {
    if (B < 1 || chans < 1 || img_size < 1 || patches < 1 || n_mask < 0 || n_mask > patches) return UVIT_ERR_SHAPE;
    const size_t n = (size_t)B * chans * img_size * img_size;
    if (n % 2) return UVIT_ERR_SHAPE;
    const uint32_t key = uvit_hash32(seed ^ (it * 0x9E3779B9u + 0x51ED270Bu));
    if (images) {
        size_t g = (n / 2 + 255) / 256;
        hipLaunchKernelGGL(synth_images_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, images, n / 2, key);
    }
    if (mask) hipLaunchKernelGGL(synth_mask_kernel, dim3(B), dim3(256), patches * sizeof(uint32_t), s, mask, patches, n_mask, key ^ 0xA5A5A5A5u);
    return uvit_check_launch();
}

// ------------------------------------------------------------------------------------------
// Variance term of the loss (engine_for_cyclical.py:130-139, 161; off in every BASELINE config):
//   z0_c = sqrt(var_r(out[r, c]) + 1e-6)  (unbiased, over the masked rows);  std_loss0 = sum_c relu(margin - z0_c) / C
//   d std_loss0 / d out[r, c] = -(out[r, c] - mean_c) / (C (M - 1) z0_c)   where z0_c < margin, else 0
// Two passes for the column statistics (mean, then centred squares: no cancellation), one for loss + gradient.
// scratch: [0, C) column sums -> means, [C, 2C) centred square sums, [2C] std_loss0.
// ------------------------------------------------------------------------------------------
#define VL_ROWS 64
template <int PASS>
__global__ __launch_bounds__(256)
void varloss_colstat_kernel(const float* __restrict__ out, const int* __restrict__ count, float* __restrict__ scratch, int Mmax, int C) {
    const int n = min(*count, Mmax);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * VL_ROWS, r1 = min(r0 + VL_ROWS, n);
    const float mean = PASS == 1 ? scratch[c] / (float)max(n, 1) : 0.f;
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float v = out[(size_t)r * C + c];
        acc += PASS == 0 ? v : (v - mean) * (v - mean);
    }
    if (r0 < r1) atomicAdd(scratch + PASS * C + c, acc);
}

__global__ __launch_bounds__(256)
void varloss_sum_kernel(const int* __restrict__ count, float* __restrict__ scratch, float w, float margin, float* __restrict__ loss,
                        float* __restrict__ std_out, int Mmax, int C) {
    __shared__ float red[4];
    const int n = min(*count, Mmax);
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float z0 = sqrtf(scratch[C + c] / (float)max(n - 1, 1) + 1e-6f);
        acc += fmaxf(margin - z0, 0.f);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (red[0] + red[1] + red[2] + red[3]) / (float)C;
        scratch[2 * C] = s;
        if (std_out) *std_out = s;
        atomicAdd(loss, w * s);
    }
}

__global__ __launch_bounds__(256)
void varloss_grad_kernel(const float* __restrict__ out, const int* __restrict__ count, const float* __restrict__ scratch, float w,
                         float margin, bf16* __restrict__ dout, int Mmax, int C) {
    const int n = min(*count, Mmax);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C || n < 2) return;
    const float mean = scratch[c] / (float)n;
    const float z0 = sqrtf(scratch[C + c] / (float)(n - 1) + 1e-6f);
    if (!(z0 < margin)) return;
    const float k = -w / ((float)C * (float)(n - 1) * z0);
    const int r0 = blockIdx.y * VL_ROWS, r1 = min(r0 + VL_ROWS, n);
    for (int r = r0; r < r1; ++r) {
        const size_t o = (size_t)r * C + c;
        dout[o] = f2bf(bf2f(dout[o]) + k * (out[o] - mean));
    }
}

int uvit_variance_loss_launch(const float* out, const int* count, float w, float margin, float loss_scale, float* scratch, float* loss,
                              float* std_loss0_out, void* dout, int Mmax, int C, hipStream_t s) {
    if (Mmax < 1 || C < 1) return UVIT_ERR_SHAPE;
    if (hipMemsetAsync(scratch, 0, (2 * (size_t)C + 16) * sizeof(float), s) != hipSuccess) return UVIT_ERR_LAUNCH;
    const dim3 grid((C + 255) / 256, (Mmax + VL_ROWS - 1) / VL_ROWS);
    hipLaunchKernelGGL(varloss_colstat_kernel<0>, grid, dim3(256), 0, s, out, count, scratch, Mmax, C);
    hipLaunchKernelGGL(varloss_colstat_kernel<1>, grid, dim3(256), 0, s, out, count, scratch, Mmax, C);
    hipLaunchKernelGGL(varloss_sum_kernel, dim3(1), dim3(256), 0, s, count, scratch, w * loss_scale, margin, loss, std_loss0_out, Mmax, C);
    hipLaunchKernelGGL(varloss_grad_kernel, grid, dim3(256), 0, s, out, count, scratch, w * loss_scale, margin, (bf16*)dout, Mmax, C);
    return uvit_check_launch();
}

// ------------------------------------------------------------------------------------------
// Absolute position embedding (--abs_pos_emb; modeling_cyclical.py:80-84,193-194): x[b, n, :] += pos[n, :] after the cls
// concat, and its gradient d pos[n, :] = sum_b dX[b, n, :].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void add_pos_kernel(float* __restrict__ x, const float* __restrict__ pos, size_t n4, size_t per_sample4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ((float4*)x)[i];
        const float4 p = ((const float4*)pos)[i % per_sample4];
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
        ((float4*)x)[i] = a;
    }
}
__global__ __launch_bounds__(256)
void pos_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dpos, int B, size_t per_sample4) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= per_sample4) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
        const float4 d = ((const float4*)dx)[(size_t)b * per_sample4 + i];
        acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
    }
    ((float4*)dpos)[i] = acc;
}
int uvit_add_pos_launch(float* x, const float* pos, int B, int N, int C, hipStream_t s) {
    if (C % 4) return UVIT_ERR_SHAPE;
    const size_t per = (size_t)N * C / 4, n4 = per * B;
    hipLaunchKernelGGL(add_pos_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, x, pos, n4, per);
    return uvit_check_launch();
}
int uvit_pos_bwd_launch(const float* dx, float* dpos, int B, int N, int C, hipStream_t s) {
    if (C % 4) return UVIT_ERR_SHAPE;
    const size_t per = (size_t)N * C / 4;
    hipLaunchKernelGGL(pos_bwd_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, dx, dpos, B, per);
    return uvit_check_launch();
}
