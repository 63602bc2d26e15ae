// Two-stream ("stochastic") Wasserstein attention, fused forward and backward, gfx950.
//
// Reference: modeling_finetune_dist.py:111-179 with wasserstein_distance_matmul
// (uncertainty_evaluations.py:276-294).  Per (batch, head), tokens i (query) and j (key):
//     m1 = sigmoid(q * scale)   c1 = sigmoid(cov_q)     m2 = sigmoid(k)    c2 = sigmoid(cov_k)
//     W_ij = |m1_i|^2 + sum(c1_i) + |m2_j|^2 + sum(c2_j) - 2 (m1_i . m2_j + sqrt(c1_i) . sqrt(c2_j))
//     P    = softmax_j( sigmoid(-W_ij) + rel_pos_bias_ij );   PD = dropout(P)
//     mean_i = sum_j PD_ij v_j          cov_i = sum_j PD_ij^2 cov_v_j
// The two dot products are ONE K=128 MFMA contraction over A_i = [m1_i | sqrt(c1_i)] and
// B_j = [m2_j | sqrt(c2_j)]; row terms r_i, c_j are fp32 side vectors.  cov_q/k/v arrive as
// ELU(.)+1 (the QKV epilogue); the backward folds ELU'(x) = min(ELU(x)+1, 1) into its outputs.
//
// Same machinery as attention.hip: [224][64] bf16 LDS images (swizzled 128-B rows) read by rows
// (ds_read_b128) and by columns (ds_read_b64_tr_b16), accumulator tiles reused as MFMA operands,
// log2-unit scores (biasP = bias*log2e, -1e30 in padded key columns), pair-hash dropout.
//   fwd     : images Bm, Bc (keys), V, CV;  the wave's queries in registers
//   bwd q   : + dMean, dCov rows in registers -> dq, dcov_q, rel-pos-bias gradient slabs, delta
//   bwd kv  : images Am, Ac (queries), dMean, dCov; the wave's keys in registers -> dk, dcov_k, dv, dcov_v
#include <mutex>
#include "common.h"
#include "uvit_internal.h"

#define HD 64
#define NT_MAX 13
#define ROWS_PAD 224
#define IMG_BYTES (ROWS_PAD * 128)
#define W2_WAVES 7
#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1e30f)

__device__ __forceinline__ int img_off2(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

enum { TR_NONE = 0, TR_SIG = 1, TR_SQRT_SIG = 2 };

// stage rows of a (tokens x 64) bf16 slice into a swizzled image with an elementwise transform; optionally
// accumulate per-row side sums (sum of sig^2 for TR_SIG, sum of sig for TR_SQRT_SIG) into rowsum[] (LDS floats)
template <int TR>
__device__ __forceinline__ void load_image_tr(char* img, const bf16* src, size_t stride, int n_valid, float pre_scale,
                                              float* rowsum, int tid, int nthreads) {
    for (int idx = tid; idx < ROWS_PAD * 8; idx += nthreads) {
        const int row = idx >> 3, chunk = idx & 7;
        bf16x8 o;
        float part = 0.f;
        if (row < n_valid) {
            const bf16x8 v = *(const bf16x8*)(src + (size_t)row * stride + chunk * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = bf2f(v[j]);
                if constexpr (TR == TR_SIG) { x = sigm(x * pre_scale); part += x * x; }
                else if constexpr (TR == TR_SQRT_SIG) { x = sigm(x); part += x; x = sqrtf(x); }
                o[j] = f2bf(x);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(0.f);
        }
        *(bf16x8*)(img + img_off2(row, chunk)) = o;
        if constexpr (TR != TR_NONE) {
            if (rowsum) {          // 8 consecutive lanes hold one row's chunks
                part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
                if (chunk == 0) atomicAdd(&rowsum[row], part);
            }
        }
    }
}

__device__ __forceinline__ bf16x8 rowf(const char* img, int row, int chunk) { return *(const bf16x8*)(img + img_off2(row, chunk)); }

__device__ __forceinline__ bf16x8 colf(const char* img, int r_lo, int r_hi, int col0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int chunk = (col0 >> 3) + (p >> 1), within = (p & 1) << 3;
    const int ra = r_lo + 4 * g + q, rb = r_hi + 4 * g + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off2(ra, chunk) + within));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off2(rb, chunk) + within));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pk8(const float* a, const float* b) {
    bf16x8 v = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return v;
}
__device__ __forceinline__ uint32_t pair_hash2(uint32_t key32, uint32_t pidx) {
    uint32_t x = (pidx ^ key32) * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x;
}
__device__ __forceinline__ void keep4b(uint32_t key32, uint32_t rowpair, int kbase, uint32_t thr16, bool (&k)[4]) {
    const uint32_t h0 = pair_hash2(key32, rowpair + (kbase >> 1)), h1 = pair_hash2(key32, rowpair + (kbase >> 1) + 1);
    k[0] = (h0 & 0xFFFFu) >= thr16; k[1] = (h0 >> 16) >= thr16; k[2] = (h1 & 0xFFFFu) >= thr16; k[3] = (h1 >> 16) >= thr16;
}
__device__ __forceinline__ bool keep1b(uint32_t key32, uint32_t rowpair, int key, uint32_t thr16) {
    const uint32_t h = pair_hash2(key32, rowpair + (key >> 1));
    return ((key & 1) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
}
__device__ __forceinline__ float gsum4(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float gmax4(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

// the wave's 16 tokens as MFMA B-operand fragments: lane (g, li) holds token li, features 8g..8g+7 (+32)
struct TokFrags { bf16x8 m[2], c[2]; float side; };     // side = sum sig^2 + sum sig over this lane's 16+16 features

__device__ __forceinline__ TokFrags load_tok(const bf16* mean_row, const bf16* cov_row, int g, float pre_scale) {
    TokFrags t;
    t.side = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 a = *(const bf16x8*)(mean_row + kk * 32 + g * 8), b = *(const bf16x8*)(cov_row + kk * 32 + g * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float m1 = sigm(bf2f(a[j]) * pre_scale), c1 = sigm(bf2f(b[j]));
            t.side += m1 * m1 + c1;
            t.m[kk][j] = f2bf(m1);
            t.c[kk][j] = f2bf(sqrtf(c1));
        }
    }
    return t;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <bool HAS_BIAS>
__global__ __launch_bounds__(W2_WAVES * 64)
void attn2_fwd_kernel(const bf16* __restrict__ qkv_m, const bf16* __restrict__ qkv_c, const float* __restrict__ biasP,
                      bf16* __restrict__ out_m, bf16* __restrict__ out_c, float* __restrict__ lse, int H, int N, int NP,
                      float scale, uint32_t drop_thr, float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *bm = smem, *bc = smem + IMG_BYTES, *vimg = smem + 2 * IMG_BYTES, *cvimg = smem + 3 * IMG_BYTES;
    float* cj = (float*)(smem + 4 * IMG_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const bf16* base_m = qkv_m + (size_t)b * N * ld + h * HD;
    const bf16* base_c = qkv_c + (size_t)b * N * ld + h * HD;
    for (int i = tid; i < ROWS_PAD; i += W2_WAVES * 64) cj[i] = 0.f;
    __syncthreads();
    load_image_tr<TR_SIG>(bm, base_m + C, ld, N, 1.0f, cj, tid, W2_WAVES * 64);
    load_image_tr<TR_SQRT_SIG>(bc, base_c + C, ld, N, 1.0f, cj, tid, W2_WAVES * 64);
    load_image_tr<TR_NONE>(vimg, base_m + 2 * C, ld, N, 1.0f, nullptr, tid, W2_WAVES * 64);
    load_image_tr<TR_NONE>(cvimg, base_c + 2 * C, ld, N, 1.0f, nullptr, tid, W2_WAVES * 64);
    __syncthreads();
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;

    for (int qt = wave; qt < nt; qt += W2_WAVES) {
        const int q = qt * 16 + li;
        const int qr = q < N ? q : N - 1;
        const TokFrags A = load_tok(base_m + (size_t)qr * ld, base_c + (size_t)qr * ld, g, scale);
        const float ri = gsum4(A.side);
        float s[NT_MAX][4];
        // all bias rows of the tile are requested up front, into the registers that will hold the scores
        if constexpr (HAS_BIAS) {
#pragma unroll
            for (int t = 0; t < NT_MAX; ++t)
                if (t < nt) {
                    const float4 bv = *(const float4*)(biasP + ((size_t)h * NP + q) * NP + t * 16 + 4 * g);
                    s[t][0] = bv.x; s[t][1] = bv.y; s[t][2] = bv.z; s[t][3] = bv.w;
                }
        }
        float mx = NEG_BIG;
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t) {
            if (t < nt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(bm, t * 16 + li, kk * 4 + g), A.m[kk], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(bc, t * 16 + li, kk * 4 + g), A.c[kk], a, 0, 0, 0);
                }
                const float4 cv4 = *(const float4*)(cj + t * 16 + 4 * g);
                const float cc[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
                float bb[4];
                if constexpr (HAS_BIAS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bb[r] = s[t][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bb[r] = (t * 16 + 4 * g + r) < N ? 0.f : NEG_BIG;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = sigm(2.0f * a[r] - ri - cc[r]) * LOG2E + bb[r];     // sigmoid(-W) + bias, log2 units
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = 0.f;
            }
        }
        mx = gmax4(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t)
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float p = __builtin_amdgcn_exp2f(s[t][r] - mx); s[t][r] = p; sum += p; }
            }
        sum = gsum4(sum);
        if (g == 0 && q < N) lse[(size_t)bh * N + q] = mx + __builtin_amdgcn_logf(sum);
        const float f = inv_keep / sum;        // PD = p * f (kept) ; PD^2 = p^2 * f^2
        f32x4 om[4], oc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { om[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; oc[dt] = om[dt]; }
        const uint32_t rowpair = ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1);
#pragma unroll
        for (int ks = 0; ks < (NT_MAX + 1) / 2; ++ks) {
            if (ks < nt2) {
                const int t0 = 2 * ks, t1 = 2 * ks + 1;
                float pa[4], pb[4], qa[4], qb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pa[r] = s[t0][r] * f; pb[r] = t1 < NT_MAX ? s[t1 < NT_MAX ? t1 : 0][r] * f : 0.f; }
                if (drop_thr) {
                    bool k4[4];
                    keep4b(drop_key, rowpair, t0 * 16 + 4 * g, drop_thr, k4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pa[r] = k4[r] ? pa[r] : 0.f;
                    keep4b(drop_key, rowpair, t1 * 16 + 4 * g, drop_thr, k4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[r] = k4[r] ? pb[r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { qa[r] = pa[r] * pa[r]; qb[r] = pb[r] * pb[r]; }
                const bf16x8 pf = pk8(pa, pb), pf2 = pk8(qa, qb);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    om[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(vimg, t0 * 16, t1 * 16, dt * 16, lane), pf, om[dt], 0, 0, 0);
                    oc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(cvimg, t0 * 16, t1 * 16, dt * 16, lane), pf2, oc[dt], 0, 0, 0);
                }
            }
        }
        if (q < N) {
            bf16* dm = out_m + ((size_t)b * N + q) * C + h * HD + 4 * g;
            bf16* dc = out_c + ((size_t)b * N + q) * C + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(bf16x4*)(dm + dt * 16) = bf16x4{f2bf(om[dt][0]), f2bf(om[dt][1]), f2bf(om[dt][2]), f2bf(om[dt][3])};
                *(bf16x4*)(dc + dt * 16) = bf16x4{f2bf(oc[dt][0]), f2bf(oc[dt][1]), f2bf(oc[dt][2]), f2bf(oc[dt][3])};
            }
        }
    }
}

// chain rule back to the pre-sigmoid inputs.  dA = gradient wrt [m | sqrt(c)], dside = gradient wrt the row term.
// mean: d/dx sigmoid(x*s) -> (dA + 2 m dside) * m (1-m) * s ;  cov: c = sigmoid(y), ELU' folded via min(y_elu1, 1)
__device__ __forceinline__ float back_mean(float dA, float dside, float x, float pre_scale) {
    const float m = sigm(x * pre_scale);
    return (dA + 2.0f * m * dside) * m * (1.0f - m) * pre_scale;
}
__device__ __forceinline__ float back_cov(float dA, float dside, float y) {   // y = ELU(pre)+1 (as stored)
    const float c = sigm(y);
    const float dc = dA * 0.5f * __builtin_amdgcn_rsqf(fmaxf(c, 1e-24f)) + dside;
    return dc * c * (1.0f - c) * fminf(y, 1.0f);
}

// ------------------------------------------------------------------------------------------
// backward, query-owned: dq, dcov_q (pre-ELU), delta, rel-pos-bias gradient over a batch chunk
// ------------------------------------------------------------------------------------------
template <bool HAS_BIAS>
__global__ __launch_bounds__(W2_WAVES * 64)
void attn2_bwd_q_kernel(const bf16* __restrict__ qkv_m, const bf16* __restrict__ qkv_c, const bf16* __restrict__ o_m,
                        const bf16* __restrict__ o_c, const bf16* __restrict__ d_m, const bf16* __restrict__ d_c,
                        const float* __restrict__ biasP, const float* __restrict__ lse, float* __restrict__ delta,
                        bf16* __restrict__ dqkv_m, bf16* __restrict__ dqkv_c, float* __restrict__ dbias_slab,
                        int accumulate_slab, int B, int H, int N, int NP, int chunk, int nhalf, float scale,
                        uint32_t drop_thr, float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *bm = smem, *bc = smem + IMG_BYTES, *vimg = smem + 2 * IMG_BYTES, *cvimg = smem + 3 * IMG_BYTES;
    float* cj = (float*)(smem + 4 * IMG_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int half = blockIdx.x % nhalf, hc = blockIdx.x / nhalf, h = hc % H, c = hc / H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;
    const int qt = half * W2_WAVES + wave;
    const bool active = qt < nt;
    const int q = qt * 16 + li;
    const int qr = q < N ? q : N - 1;
    float dbacc[NT_MAX][4];
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbacc[t][r] = 0.f;

    for (int bi = 0; bi < chunk; ++bi) {
        const int b = c * chunk + bi;
        if (b >= B) break;
        const int bh = b * H + h;
        const bf16* base_m = qkv_m + (size_t)b * N * ld + h * HD;
        const bf16* base_c = qkv_c + (size_t)b * N * ld + h * HD;
        __syncthreads();
        for (int i = tid; i < ROWS_PAD; i += W2_WAVES * 64) cj[i] = 0.f;
        __syncthreads();
        load_image_tr<TR_SIG>(bm, base_m + C, ld, N, 1.0f, cj, tid, W2_WAVES * 64);
        load_image_tr<TR_SQRT_SIG>(bc, base_c + C, ld, N, 1.0f, cj, tid, W2_WAVES * 64);
        load_image_tr<TR_NONE>(vimg, base_m + 2 * C, ld, N, 1.0f, nullptr, tid, W2_WAVES * 64);
        load_image_tr<TR_NONE>(cvimg, base_c + 2 * C, ld, N, 1.0f, nullptr, tid, W2_WAVES * 64);
        __syncthreads();
        if (!active) continue;
        const TokFrags A = load_tok(base_m + (size_t)qr * ld, base_c + (size_t)qr * ld, g, scale);
        const float ri = gsum4(A.side);
        const size_t orow = ((size_t)b * N + qr) * C + h * HD;
        bf16x8 dmf[2], dcf[2];
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            dmf[kk] = *(const bf16x8*)(d_m + orow + kk * 32 + g * 8);
            dcf[kk] = *(const bf16x8*)(d_c + orow + kk * 32 + g * 8);
            const bf16x8 om = *(const bf16x8*)(o_m + orow + kk * 32 + g * 8), oc = *(const bf16x8*)(o_c + orow + kk * 32 + g * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(dmf[kk][j]) * bf2f(om[j]) + 2.0f * bf2f(dcf[kk][j]) * bf2f(oc[j]);
        }
        dl = gsum4(dl);                                   // delta_i = dM.mean + 2 dC.cov
        const float lse_q = lse[(size_t)bh * N + qr];
        if (g == 0 && q < N) delta[(size_t)bh * N + q] = dl;
        const uint32_t rowpair = ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1);

        f32x4 dam[4], dac[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dam[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dac[dt] = dam[dt]; }
        float dside = 0.f;
#pragma unroll
        for (int ks = 0; ks < (NT_MAX + 1) / 2; ++ks) {
            if (ks < nt2) {
                float gw[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * ks + tt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) gw[tt][r] = 0.f;
                    if (t < nt && t < NT_MAX) {
                        f32x4 a = {0.f, 0.f, 0.f, 0.f}, pm = {0.f, 0.f, 0.f, 0.f}, pc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(bm, t * 16 + li, kk * 4 + g), A.m[kk], a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(bc, t * 16 + li, kk * 4 + g), A.c[kk], a, 0, 0, 0);
                            pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(vimg, t * 16 + li, kk * 4 + g), dmf[kk], pm, 0, 0, 0);
                            pc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(cvimg, t * 16 + li, kk * 4 + g), dcf[kk], pc, 0, 0, 0);
                        }
                        const float4 cv4 = *(const float4*)(cj + t * 16 + 4 * g);
                        const float cc[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
                        float bb[4];
                        if constexpr (HAS_BIAS) {
                            const float4 bv = *(const float4*)(biasP + ((size_t)h * NP + q) * NP + t * 16 + 4 * g);
                            bb[0] = bv.x; bb[1] = bv.y; bb[2] = bv.z; bb[3] = bv.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) bb[r] = (t * 16 + 4 * g + r) < N ? 0.f : NEG_BIG;
                        }
                        bool k4[4] = {true, true, true, true};
                        if (drop_thr) keep4b(drop_key, rowpair, t * 16 + 4 * g, drop_thr, k4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sg = sigm(2.0f * a[r] - ri - cc[r]);
                            const float p = __builtin_amdgcn_exp2f(sg * LOG2E + bb[r] - lse_q);
                            const float pd = k4[r] ? p * inv_keep : 0.f;
                            // dPD = dM.v + 2 PD (dC.cv);  dP = D dPD;  ds = P (dP - delta)
                            const float dpd = pm[r] + 2.0f * pd * pc[r];
                            const float ds = p * ((k4[r] ? dpd * inv_keep : 0.f) - dl);
                            dbacc[t < NT_MAX ? t : 0][r] += ds;
                            const float gv = -ds * sg * (1.0f - sg);                  // dL/dW
                            gw[tt][r] = gv;
                            dside += gv;
                        }
                    }
                }
                const bf16x8 gf = pk8(gw[0], gw[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dam[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(bm, 2 * ks * 16, (2 * ks + 1) * 16, dt * 16, lane), gf, dam[dt], 0, 0, 0);
                    dac[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(bc, 2 * ks * 16, (2 * ks + 1) * 16, dt * 16, lane), gf, dac[dt], 0, 0, 0);
                }
            }
        }
        dside = gsum4(dside);                              // d r_i = sum_j dL/dW_ij
        if (q < N) {
            const bf16* xm = base_m + (size_t)q * ld + 4 * g;
            const bf16* xc = base_c + (size_t)q * ld + 4 * g;
            bf16* om = dqkv_m + ((size_t)b * N + q) * ld + h * HD + 4 * g;
            bf16* oc = dqkv_c + ((size_t)b * N + q) * ld + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x4 vm = *(const bf16x4*)(xm + dt * 16), vc = *(const bf16x4*)(xc + dt * 16);
                bf16x4 rm, rc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {       // dA = -2 sum_j g B_j
                    rm[r] = f2bf(back_mean(-2.0f * dam[dt][r], dside, bf2f(vm[r]), scale));
                    rc[r] = f2bf(back_cov(-2.0f * dac[dt][r], dside, bf2f(vc[r])));
                }
                *(bf16x4*)(om + dt * 16) = rm;
                *(bf16x4*)(oc + dt * 16) = rc;
            }
        }
    }
    if (dbias_slab && active) {
        float* slab = dbias_slab + ((size_t)(c * H + h) * NP) * NP;
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t)
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* p = slab + (size_t)(t * 16 + 4 * g + r) * NP + q;
                    *p = accumulate_slab ? *p + dbacc[t][r] : dbacc[t][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// backward, key-owned: dk, dcov_k, dv, dcov_v (cov parts pre-ELU)
// ------------------------------------------------------------------------------------------
template <bool HAS_BIAS>
__global__ __launch_bounds__(W2_WAVES * 64)
void attn2_bwd_kv_kernel(const bf16* __restrict__ qkv_m, const bf16* __restrict__ qkv_c, const bf16* __restrict__ d_m,
                         const bf16* __restrict__ d_c, const float* __restrict__ biasP, const float* __restrict__ lse,
                         const float* __restrict__ delta, bf16* __restrict__ dqkv_m, bf16* __restrict__ dqkv_c, int H,
                         int N, int NP, int nhalf, float scale, uint32_t drop_thr, float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *am = smem, *ac = smem + IMG_BYTES, *dmimg = smem + 2 * IMG_BYTES, *dcimg = smem + 3 * IMG_BYTES;
    float* ri_s = (float*)(smem + 4 * IMG_BYTES);
    float* lse_s = ri_s + ROWS_PAD;
    float* dl_s = lse_s + ROWS_PAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int half = blockIdx.x % nhalf, bh = blockIdx.x / nhalf, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const bf16* base_m = qkv_m + (size_t)b * N * ld + h * HD;
    const bf16* base_c = qkv_c + (size_t)b * N * ld + h * HD;
    for (int i = tid; i < ROWS_PAD; i += W2_WAVES * 64) {
        ri_s[i] = 0.f;
        lse_s[i] = i < N ? lse[(size_t)bh * N + i] : 0.f;
        dl_s[i] = i < N ? delta[(size_t)bh * N + i] : 0.f;
    }
    __syncthreads();
    load_image_tr<TR_SIG>(am, base_m, ld, N, scale, ri_s, tid, W2_WAVES * 64);
    load_image_tr<TR_SQRT_SIG>(ac, base_c, ld, N, 1.0f, ri_s, tid, W2_WAVES * 64);
    load_image_tr<TR_NONE>(dmimg, d_m + (size_t)b * N * C + h * HD, C, N, 1.0f, nullptr, tid, W2_WAVES * 64);
    load_image_tr<TR_NONE>(dcimg, d_c + (size_t)b * N * C + h * HD, C, N, 1.0f, nullptr, tid, W2_WAVES * 64);
    __syncthreads();
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;
    const int kt = half * W2_WAVES + wave;
    if (kt >= nt) return;
    const int key = kt * 16 + li;
    const int kr = key < N ? key : N - 1;
    const TokFrags Bf = load_tok(base_m + C + (size_t)kr * ld, base_c + C + (size_t)kr * ld, g, 1.0f);
    const float cjv = key < N ? gsum4(Bf.side) : gsum4(Bf.side);
    bf16x8 vf[2], cvf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        vf[kk] = *(const bf16x8*)(base_m + 2 * C + (size_t)kr * ld + kk * 32 + g * 8);
        cvf[kk] = *(const bf16x8*)(base_c + 2 * C + (size_t)kr * ld + kk * 32 + g * 8);
    }
    f32x4 dbm[4], dbc[4], dv[4], dcv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dbm[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dbc[dt] = dbm[dt]; dv[dt] = dbm[dt]; dcv[dt] = dbm[dt]; }
    float dside = 0.f;

#pragma unroll 1
    for (int qs = 0; qs < nt2; ++qs) {
        float pdv[2][4], pd2[2][4], gw[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int qt = 2 * qs + tt;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pdv[tt][r] = 0.f; pd2[tt][r] = 0.f; gw[tt][r] = 0.f; }
            if (qt < nt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, pm = {0.f, 0.f, 0.f, 0.f}, pc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(am, qt * 16 + li, kk * 4 + g), Bf.m[kk], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(ac, qt * 16 + li, kk * 4 + g), Bf.c[kk], a, 0, 0, 0);
                    pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(dmimg, qt * 16 + li, kk * 4 + g), vf[kk], pm, 0, 0, 0);
                    pc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowf(dcimg, qt * 16 + li, kk * 4 + g), cvf[kk], pc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded query rows: zero A / dM / dC / delta / lse rows and a zero bias row -> finite p, zero gradients
                    const int q = qt * 16 + 4 * g + r;
                    float bv;
                    if constexpr (HAS_BIAS) bv = biasP[((size_t)h * NP + q) * NP + key]; else bv = key < N ? 0.f : NEG_BIG;
                    const float sg = sigm(2.0f * a[r] - ri_s[q] - cjv);
                    const float p = __builtin_amdgcn_exp2f(sg * LOG2E + bv - lse_s[q]);
                    bool kp = true;
                    if (drop_thr) kp = keep1b(drop_key, ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1), key, drop_thr);
                    const float pd = kp ? p * inv_keep : 0.f;
                    const float dpd = pm[r] + 2.0f * pd * pc[r];
                    const float ds = p * ((kp ? dpd * inv_keep : 0.f) - dl_s[q]);
                    const float gv = q < N ? -ds * sg * (1.0f - sg) : 0.f;
                    pdv[tt][r] = pd; pd2[tt][r] = pd * pd; gw[tt][r] = gv;
                    dside += gv;
                }
            }
        }
        const bf16x8 pf = pk8(pdv[0], pdv[1]), pf2 = pk8(pd2[0], pd2[1]), gf = pk8(gw[0], gw[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(dmimg, 2 * qs * 16, (2 * qs + 1) * 16, dt * 16, lane), pf, dv[dt], 0, 0, 0);
            dcv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(dcimg, 2 * qs * 16, (2 * qs + 1) * 16, dt * 16, lane), pf2, dcv[dt], 0, 0, 0);
            dbm[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(am, 2 * qs * 16, (2 * qs + 1) * 16, dt * 16, lane), gf, dbm[dt], 0, 0, 0);
            dbc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(ac, 2 * qs * 16, (2 * qs + 1) * 16, dt * 16, lane), gf, dbc[dt], 0, 0, 0);
        }
    }
    dside = gsum4(dside);                                  // d c_j = sum_i dL/dW_ij
    if (key < N) {
        const bf16* xk = base_m + C + (size_t)key * ld + 4 * g;
        const bf16* xck = base_c + C + (size_t)key * ld + 4 * g;
        const bf16* xcv = base_c + 2 * C + (size_t)key * ld + 4 * g;
        bf16* om = dqkv_m + ((size_t)b * N + key) * ld + h * HD + 4 * g;
        bf16* oc = dqkv_c + ((size_t)b * N + key) * ld + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x4 vk = *(const bf16x4*)(xk + dt * 16), vck = *(const bf16x4*)(xck + dt * 16), vcv = *(const bf16x4*)(xcv + dt * 16);
            bf16x4 rk, rck, rv, rcv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rk[r] = f2bf(back_mean(-2.0f * dbm[dt][r], dside, bf2f(vk[r]), 1.0f));
                rck[r] = f2bf(back_cov(-2.0f * dbc[dt][r], dside, bf2f(vck[r])));
                rv[r] = f2bf(dv[dt][r]);
                rcv[r] = f2bf(dcv[dt][r] * fminf(bf2f(vcv[r]), 1.0f));           // ELU' of the cov_v pre-activation
            }
            *(bf16x4*)(om + C + dt * 16) = rk;
            *(bf16x4*)(om + 2 * C + dt * 16) = rv;
            *(bf16x4*)(oc + C + dt * 16) = rck;
            *(bf16x4*)(oc + 2 * C + dt * 16) = rcv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define FWD2_LDS (4 * IMG_BYTES + ROWS_PAD * 4)
#define BKV2_LDS (4 * IMG_BYTES + 3 * ROWS_PAD * 4)
static std::once_flag g_attr2_once;
static void init2_impl() {
#define SETA(K, B) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, B)
    SETA(attn2_fwd_kernel<true>, FWD2_LDS); SETA(attn2_fwd_kernel<false>, FWD2_LDS);
    SETA(attn2_bwd_q_kernel<true>, FWD2_LDS); SETA(attn2_bwd_q_kernel<false>, FWD2_LDS);
    SETA(attn2_bwd_kv_kernel<true>, BKV2_LDS); SETA(attn2_bwd_kv_kernel<false>, BKV2_LDS);
#undef SETA
}
static void init2() { std::call_once(g_attr2_once, init2_impl); }

int uvit_attn2_fwd_launch(const void* qkv_m, const void* qkv_c, const float* biasP, void* out_m, void* out_c, float* lse,
                          int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s) {
    if (B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16) return UVIT_ERR_SHAPE;
    init2();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
#define FA dim3(B * H), dim3(W2_WAVES * 64), FWD2_LDS, s, (const bf16*)qkv_m, (const bf16*)qkv_c, biasP, (bf16*)out_m, (bf16*)out_c, lse, \
        H, N, NP, scale, thr, inv_keep, uvit_layer_key(seed, layer)
    if (biasP) hipLaunchKernelGGL(attn2_fwd_kernel<true>, FA); else hipLaunchKernelGGL(attn2_fwd_kernel<false>, FA);
#undef FA
    return uvit_check_launch();
}

int uvit_attn2_bwd_launch(const void* qkv_m, const void* qkv_c, const void* o_m, const void* o_c, const void* d_m, const void* d_c,
                          const float* biasP, const float* lse, float* delta, void* dqkv_m, void* dqkv_c, float* dbias_slab,
                          int accumulate_slab, int chunk, int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed,
                          uint32_t layer, hipStream_t s) {
    if (B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16 || chunk <= 0) return UVIT_ERR_SHAPE;
    init2();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t key = uvit_layer_key(seed, layer);
    const int nt = (N + 15) / 16, nhalf = nt > W2_WAVES ? 2 : 1;
    const int nchunk = (B + chunk - 1) / chunk;
#define QA dim3(H * nchunk * nhalf), dim3(W2_WAVES * 64), FWD2_LDS, s, (const bf16*)qkv_m, (const bf16*)qkv_c, (const bf16*)o_m, \
        (const bf16*)o_c, (const bf16*)d_m, (const bf16*)d_c, biasP, lse, delta, (bf16*)dqkv_m, (bf16*)dqkv_c, dbias_slab, accumulate_slab, \
        B, H, N, NP, chunk, nhalf, scale, thr, inv_keep, key
    if (biasP) hipLaunchKernelGGL(attn2_bwd_q_kernel<true>, QA); else hipLaunchKernelGGL(attn2_bwd_q_kernel<false>, QA);
#undef QA
    int rc = uvit_check_launch(); if (rc) return rc;
#define KA dim3(B * H * nhalf), dim3(W2_WAVES * 64), BKV2_LDS, s, (const bf16*)qkv_m, (const bf16*)qkv_c, (const bf16*)d_m, \
        (const bf16*)d_c, biasP, lse, delta, (bf16*)dqkv_m, (bf16*)dqkv_c, H, N, NP, nhalf, scale, thr, inv_keep, key
    if (biasP) hipLaunchKernelGGL(attn2_bwd_kv_kernel<true>, KA); else hipLaunchKernelGGL(attn2_bwd_kv_kernel<false>, KA);
#undef KA
    return uvit_check_launch();
}
