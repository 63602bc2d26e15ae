// (units) biasP is the relative-position bias times log2(e) with -1e30 in padded key columns; LSE is kept in
// log2 units -- both are private to these kernels (uvit_relpos_gather_launch builds biasP).
// Fused multi-head attention with shared relative-position bias and attention dropout for
// ViT token counts (N <= 208, head_dim 64), gfx950.  Scores never touch HBM.
//
// Reference semantics: modeling_finetune.py:145-188 (Attention.forward) -- softmax(q*scale @ k^T
// + rel_pos_bias) -> dropout -> @ v.
//
// Layout: one workgroup owns one (batch, head); the head's K/V (fwd, bwd-dQ) or Q/dO (bwd-dKdV)
// live in LDS as [224 rows][64] bf16 images (128-B rows, 16-B chunks XOR-swizzled by row&7), read
// by rows with ds_read_b128 and by columns with ds_read_b64_tr_b16 -- both conflict-free on the
// same image.  Each wave owns 16-row tiles; products use v_mfma_f32_16x16x32_bf16 and are
// oriented so that (a) the softmax reduction axis lies in registers + 2 wave shuffles and (b) an
// accumulator tile is directly the next MFMA's operand (no LDS round trip for P / dS).
//
//   fwd     : S^T = K.Q^T  -> softmax over keys -> O^T = V^T.P^T            (+ LSE saved)
//   bwd dQ  : S^T, dP^T = V.dO^T, dS^T -> dQ^T = K^T.dS^T ; dBias^T accumulated over a batch chunk
//   bwd dKV : S = Q.K^T, dP = dO.V^T  -> dV^T = dO^T.(P.D), dK^T = Q^T.dS
#include "common.h"
#include "uvit_internal.h"

#define HD 64
#define NT_MAX 13            // 13 * 16 = 208 >= 197 tokens
#define ROWS_PAD 224         // 14 * 16: k-steps pair two 16-row tiles
#define IMG_BYTES (ROWS_PAD * 128)
#define BWD_WAVES 7
#ifndef DQ_OCC
#define DQ_OCC 4
#endif

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// global [rows][stride] bf16 (64 used columns) -> swizzled LDS image, rows >= n_valid zero-filled
__device__ __forceinline__ void load_image(char* img, const bf16* src, size_t stride, int n_valid, int tid, int nthreads) {
    for (int idx = tid; idx < ROWS_PAD * 8; idx += nthreads) {
        const int row = idx >> 3, chunk = idx & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < n_valid) v = *(const uint4*)(src + (size_t)row * stride + chunk * 8);
        *(uint4*)(img + img_off(row, chunk)) = v;
    }
}

__device__ __forceinline__ bf16x8 row_frag(const char* img, int row, int chunk) {
    return *(const bf16x8*)(img + img_off(row, chunk));
}

// operand element j of lane (g, i):  img[row = (j<4 ? r_lo : r_hi) + 4g + (j&3)][col0 + i]
__device__ __forceinline__ bf16x8 col_frag(const char* img, int r_lo, int r_hi, int col0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int chunk = (col0 >> 3) + (p >> 1), within = (p & 1) << 3;
    const int ra = r_lo + 4 * g + q, rb = r_hi + 4 * g + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off(ra, chunk) + within));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off(rb, chunk) + within));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const float* a, const float* b) {
    bf16x8 v = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return v;
}

// Dropout: one 32-bit hash per (query row, key pair); each key takes a 16-bit half and is kept when
// half >= round(p * 65536).  pair index = (bh*N + q) * (NP/2) + (key >> 1).  Mirrored by
// oracle/vit_oracle.py::attn_keep_mask.
__device__ __forceinline__ uint32_t pair_hash(uint32_t key32, uint32_t pidx) {
    uint32_t x = (pidx ^ key32) * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13;
    return x;
}
// keep flags of the 4 consecutive keys kbase..kbase+3 (kbase % 4 == 0) of one query row
__device__ __forceinline__ void keep4(uint32_t key32, uint32_t rowpair, int kbase, uint32_t thr16, bool (&k)[4]) {
    const uint32_t h0 = pair_hash(key32, rowpair + (kbase >> 1)), h1 = pair_hash(key32, rowpair + (kbase >> 1) + 1);
    k[0] = (h0 & 0xFFFFu) >= thr16; k[1] = (h0 >> 16) >= thr16;
    k[2] = (h1 & 0xFFFFu) >= thr16; k[3] = (h1 >> 16) >= thr16;
}
__device__ __forceinline__ bool keep1(uint32_t key32, uint32_t rowpair, int key, uint32_t thr16) {
    const uint32_t h = pair_hash(key32, rowpair + (key >> 1));
    return ((key & 1) ? (h >> 16) : (h & 0xFFFFu)) >= thr16;
}
#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1e30f)

__device__ __forceinline__ float group_sum4(float v) {   // sum over the 4 lane groups (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_max4(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NW, bool HAS_BIAS>
__global__ __launch_bounds__(NW * 64, 4)      // two 7-wave workgroups per CU need <= 128 VGPRs
void attn_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ biasP, bf16* __restrict__ out,
                     float* __restrict__ lse, int H, int N, int NP, float scale, uint32_t drop_thr,
                     float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const bf16* base = qkv + (size_t)b * N * ld + h * HD;
    load_image(kimg, base + C, ld, N, tid, NW * 64);
    load_image(vimg, base + 2 * C, ld, N, tid, NW * 64);
    __syncthreads();
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;

    // the Q fragments of a wave's next tile are fetched while it works on the current one
    bf16x8 qn[2];
    {
        const int q0 = wave * 16 + li, qr0 = q0 < N ? q0 : N - 1;
        qn[0] = *(const bf16x8*)(base + (size_t)qr0 * ld + g * 8);
        qn[1] = *(const bf16x8*)(base + (size_t)qr0 * ld + 32 + g * 8);
    }
    for (int qt = wave; qt < nt; qt += NW) {
        const int q = qt * 16 + li;
        bf16x8 qf[2] = {qn[0], qn[1]};
        // scores in log2 units: s' = (q.k) * scale*log2(e) + biasP   (biasP is pre-multiplied by log2(e) and holds
        // -1e30 in padded key columns, so padded keys vanish in the softmax without per-element selects).
        // All bias rows of the tile are requested up front, into the registers that will hold the scores.
        float s[NT_MAX][4];
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t) {
            if constexpr (HAS_BIAS) {
                if (t < nt) {
                    const float4 bv = *(const float4*)(biasP + ((size_t)h * NP + q) * NP + t * 16 + 4 * g);
                    s[t][0] = bv.x; s[t][1] = bv.y; s[t][2] = bv.z; s[t][3] = bv.w;
                }
            }
        }
        if (qt + NW < nt) {
            const int q1 = (qt + NW) * 16 + li, qr1 = q1 < N ? q1 : N - 1;
            qn[0] = *(const bf16x8*)(base + (size_t)qr1 * ld + g * 8);
            qn[1] = *(const bf16x8*)(base + (size_t)qr1 * ld + 32 + g * 8);
        }
        float mx = NEG_BIG;
        const float c = scale * LOG2E;
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t) {
            if (t < nt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(kimg, t * 16 + li, g), qf[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(kimg, t * 16 + li, 4 + g), qf[1], a, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float bb;
                    if constexpr (HAS_BIAS) bb = s[t][r]; else bb = (t * 16 + 4 * g + r) < N ? 0.f : NEG_BIG;
                    const float v = a[r] * c + bb;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) s[t][r] = 0.f;
            }
        }
        mx = group_max4(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[t][r] - mx);
                    s[t][r] = p;
                    sum += p;
                }
            }
        }
        sum = group_sum4(sum);
        if (g == 0 && q < N) lse[(size_t)bh * N + q] = mx + __builtin_amdgcn_logf(sum);      // log2 units
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t rowpair = ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1);
#pragma unroll
        for (int ks = 0; ks < (NT_MAX + 1) / 2; ++ks) {
            if (ks < nt2) {
                const int t0 = 2 * ks, t1 = 2 * ks + 1;
                float pa[4], pb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pa[r] = s[t0][r]; pb[r] = t1 < NT_MAX ? s[t1 < NT_MAX ? t1 : 0][r] : 0.f; }
                if (drop_thr) {        // dropout applied while packing P: short live ranges for the hash values
                    bool k4[4];
                    keep4(drop_key, rowpair, t0 * 16 + 4 * g, drop_thr, k4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pa[r] = k4[r] ? pa[r] : 0.f;
                    keep4(drop_key, rowpair, t1 * 16 + 4 * g, drop_thr, k4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pb[r] = k4[r] ? pb[r] : 0.f;
                }
                const bf16x8 pf = pack8(pa, pb);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vf = col_frag(vimg, t0 * 16, t1 * 16, dt * 16, lane);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        }
        if (q < N) {
            const float f = inv_keep / sum;
            bf16* dst = out + ((size_t)b * N + q) * C + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 v = {f2bf(o[dt][0] * f), f2bf(o[dt][1] * f), f2bf(o[dt][2] * f), f2bf(o[dt][3] * f)};
                *(bf16x4*)(dst + dt * 16) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, query-owned: dQ, delta, and the rel-pos-bias gradient summed over a batch chunk
// ------------------------------------------------------------------------------------------
template <bool HAS_BIAS>
__global__ __launch_bounds__(BWD_WAVES * 64, DQ_OCC)
void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o_fwd, const bf16* __restrict__ d_o,
                        const float* __restrict__ biasP, const float* __restrict__ lse, float* __restrict__ delta,
                        bf16* __restrict__ dqkv, float* __restrict__ dbias_slab, int accumulate_slab,
                        int B, int H, int N, int NP, int chunk, int nhalf, float scale, uint32_t drop_thr,
                        float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int half = blockIdx.x % nhalf;
    const int hc = blockIdx.x / nhalf;
    const int h = hc % H, c = hc / H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;
    const int qt = half * BWD_WAVES + wave;
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
        const bf16* base = qkv + (size_t)b * N * ld + h * HD;
        __syncthreads();
        load_image(kimg, base + C, ld, N, tid, BWD_WAVES * 64);
        load_image(vimg, base + 2 * C, ld, N, tid, BWD_WAVES * 64);
        __syncthreads();
        if (!active) continue;
        bf16x8 qf[2], dof[2];
        const size_t orow = ((size_t)b * N + qr) * C + h * HD;
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = *(const bf16x8*)(base + (size_t)qr * ld + kk * 32 + g * 8);
            dof[kk] = *(const bf16x8*)(d_o + orow + kk * 32 + g * 8);
            const bf16x8 of = *(const bf16x8*)(o_fwd + orow + kk * 32 + g * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(dof[kk][j]) * bf2f(of[j]);
        }
        dl = group_sum4(dl);
        const float lse_q = lse[(size_t)bh * N + qr];
        if (g == 0 && q < N) delta[(size_t)bh * N + q] = dl;
        const uint32_t rowpair = ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1);
        const float c = scale * LOG2E;

        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < (NT_MAX + 1) / 2; ++ks) {
            if (ks < nt2) {
                float dsv[2][4];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * ks + tt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dsv[tt][r] = 0.f;
                    if (t < nt && t < NT_MAX) {
                        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(kimg, t * 16 + li, kk * 4 + g), qf[kk], s, 0, 0, 0);
                            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(vimg, t * 16 + li, kk * 4 + g), dof[kk], dp, 0, 0, 0);
                        }
                        float bb[4];
                        if constexpr (HAS_BIAS) {
                            const float4 bv = *(const float4*)(biasP + ((size_t)h * NP + q) * NP + t * 16 + 4 * g);
                            bb[0] = bv.x; bb[1] = bv.y; bb[2] = bv.z; bb[3] = bv.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) bb[r] = (t * 16 + 4 * g + r) < N ? 0.f : NEG_BIG;
                        }
                        bool k4[4] = {true, true, true, true};
                        if (drop_thr) keep4(drop_key, rowpair, t * 16 + 4 * g, drop_thr, k4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // padded keys: bias = -1e30 -> p = 0; padded query lanes are never stored / read
                            const float p = __builtin_amdgcn_exp2f(s[r] * c + bb[r] - lse_q);
                            const float dpv = k4[r] ? dp[r] * inv_keep : 0.f;
                            const float ds = p * (dpv - dl);
                            dsv[tt][r] = ds;
                            dbacc[t < NT_MAX ? t : 0][r] += ds;
                        }
                    }
                }
                const bf16x8 dsf = pack8(dsv[0], dsv[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 kf = col_frag(kimg, 2 * ks * 16, (2 * ks + 1) * 16, dt * 16, lane);
                    dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, dq[dt], 0, 0, 0);
                }
            }
        }
        if (q < N) {
            bf16* dst = dqkv + ((size_t)b * N + q) * ld + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 v = {f2bf(dq[dt][0] * scale), f2bf(dq[dt][1] * scale), f2bf(dq[dt][2] * scale), f2bf(dq[dt][3] * scale)};
                *(bf16x4*)(dst + dt * 16) = v;
            }
        }
    }
    if (dbias_slab && active) {
        // slab[c][h][key][q]  (transposed: q is the contiguous index, 16 lanes -> 64 B)
        float* slab = dbias_slab + ((size_t)(c * H + h) * NP) * NP;
#pragma unroll
        for (int t = 0; t < NT_MAX; ++t) {
            if (t < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    float* p = slab + (size_t)key * NP + q;
                    *p = accumulate_slab ? *p + dbacc[t][r] : dbacc[t][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, key-owned: dK, dV
// ------------------------------------------------------------------------------------------
template <bool HAS_BIAS>
__global__ __launch_bounds__(BWD_WAVES * 64, 4)      // two workgroups per CU
void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o, const float* __restrict__ biasP,
                         const float* __restrict__ lse, const float* __restrict__ delta, bf16* __restrict__ dqkv,
                         int H, int N, int NP, int nhalf, float scale, uint32_t drop_thr, float inv_keep,
                         uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qimg = smem;
    char* doimg = smem + IMG_BYTES;
    float* lse_s = (float*)(smem + 2 * IMG_BYTES);
    float* dl_s = lse_s + ROWS_PAD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int half = blockIdx.x % nhalf;
    const int bh = blockIdx.x / nhalf, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const bf16* base = qkv + (size_t)b * N * ld + h * HD;
    load_image(qimg, base, ld, N, tid, BWD_WAVES * 64);
    load_image(doimg, d_o + (size_t)b * N * C + h * HD, C, N, tid, BWD_WAVES * 64);
    for (int i = tid; i < ROWS_PAD; i += BWD_WAVES * 64) {
        lse_s[i] = i < N ? lse[(size_t)bh * N + i] : 0.f;
        dl_s[i] = i < N ? delta[(size_t)bh * N + i] : 0.f;
    }
    __syncthreads();
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;
    const int kt = half * BWD_WAVES + wave;
    if (kt >= nt) return;
    const int key = kt * 16 + li;
    const int kr = key < N ? key : N - 1;
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        kf[kk] = *(const bf16x8*)(base + C + (size_t)kr * ld + kk * 32 + g * 8);
        vf[kk] = *(const bf16x8*)(base + 2 * C + (size_t)kr * ld + kk * 32 + g * 8);
    }
    const float c = scale * LOG2E;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int qs = 0; qs < nt2; ++qs) {
        float pdv[2][4], dsv[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int qt = 2 * qs + tt;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pdv[tt][r] = 0.f; dsv[tt][r] = 0.f; }
            if (qt < nt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(qimg, qt * 16 + li, kk * 4 + g), kf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag(doimg, qt * 16 + li, kk * 4 + g), vf[kk], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded query rows have zero Q / dO / delta / lse and a zero bias row: p stays finite and its
                    // products vanish; padded keys (bias -1e30) give p = 0
                    const int q = qt * 16 + 4 * g + r;
                    float bv;
                    if constexpr (HAS_BIAS) bv = biasP[((size_t)h * NP + q) * NP + key]; else bv = key < N ? 0.f : NEG_BIG;
                    const float p = __builtin_amdgcn_exp2f(s[r] * c + bv - lse_s[q]);
                    float dmul = 1.0f;
                    if (drop_thr) dmul = keep1(drop_key, ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1), key, drop_thr) ? inv_keep : 0.f;
                    pdv[tt][r] = p * dmul;
                    dsv[tt][r] = p * (dmul * dp[r] - dl_s[q]);
                }
            }
        }
        const bf16x8 pdf = pack8(pdv[0], pdv[1]);
        const bf16x8 dsf = pack8(dsv[0], dsv[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 dot = col_frag(doimg, 2 * qs * 16, (2 * qs + 1) * 16, dt * 16, lane);
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pdf, dv[dt], 0, 0, 0);
            const bf16x8 qtf = col_frag(qimg, 2 * qs * 16, (2 * qs + 1) * 16, dt * 16, lane);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dk[dt], 0, 0, 0);
        }
    }
    if (key < N) {
        bf16* dst = dqkv + ((size_t)b * N + key) * ld + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 kv = {f2bf(dk[dt][0] * scale), f2bf(dk[dt][1] * scale), f2bf(dk[dt][2] * scale), f2bf(dk[dt][3] * scale)};
            bf16x4 vv = {f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
            *(bf16x4*)(dst + C + dt * 16) = kv;
            *(bf16x4*)(dst + 2 * C + dt * 16) = vv;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
#define FWD_WAVES 7
static bool g_attn_attr = false;
static void attn_init_once() {
    if (g_attn_attr) return;
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<FWD_WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<FWD_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES + 2 * ROWS_PAD * 4);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES + 2 * ROWS_PAD * 4);
    g_attn_attr = true;
}

static int attn_check(int B, int H, int N, int head_dim) {
    if (head_dim != HD || B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16) return UVIT_ERR_SHAPE;
    return UVIT_OK;
}

int uvit_attn_fwd_launch(const void* qkv, const float* biasP, void* out, float* lse, int B, int H, int N, int NP,
                         float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s) {
    int rc = attn_check(B, H, N, HD); if (rc) return rc;
    attn_init_once();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    if (biasP) hipLaunchKernelGGL((attn_fwd_kernel<FWD_WAVES, true>), dim3(B * H), dim3(FWD_WAVES * 64), 2 * IMG_BYTES, s, (const bf16*)qkv,
                                  biasP, (bf16*)out, lse, H, N, NP, scale, thr, inv_keep, uvit_layer_key(seed, layer));
    else hipLaunchKernelGGL((attn_fwd_kernel<FWD_WAVES, false>), dim3(B * H), dim3(FWD_WAVES * 64), 2 * IMG_BYTES, s, (const bf16*)qkv,
                            biasP, (bf16*)out, lse, H, N, NP, scale, thr, inv_keep, uvit_layer_key(seed, layer));
    return uvit_check_launch();
}

int uvit_attn_bwd_launch(const void* qkv, const void* o_fwd, const void* d_o, const float* biasP, const float* lse,
                         float* delta, void* dqkv, float* dbias_slab, int accumulate_slab, int chunk, int B, int H,
                         int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s) {
    int rc = attn_check(B, H, N, HD); if (rc) return rc;
    if (chunk <= 0) return UVIT_ERR_ARG;
    attn_init_once();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t key = uvit_layer_key(seed, layer);
    const int nt = (N + 15) / 16, nhalf = nt > BWD_WAVES ? 2 : 1;
    const int nchunk = (B + chunk - 1) / chunk;
#define DQ_ARGS dim3(H * nchunk * nhalf), dim3(BWD_WAVES * 64), 2 * IMG_BYTES, s, (const bf16*)qkv, (const bf16*)o_fwd, (const bf16*)d_o, \
        biasP, lse, delta, (bf16*)dqkv, dbias_slab, accumulate_slab, B, H, N, NP, chunk, nhalf, scale, thr, inv_keep, key
    if (biasP) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, DQ_ARGS); else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, DQ_ARGS);
#undef DQ_ARGS
    rc = uvit_check_launch(); if (rc) return rc;
#define DKV_ARGS dim3(B * H * nhalf), dim3(BWD_WAVES * 64), 2 * IMG_BYTES + 2 * ROWS_PAD * 4, s, (const bf16*)qkv, (const bf16*)d_o, \
        biasP, lse, delta, (bf16*)dqkv, H, N, NP, nhalf, scale, thr, inv_keep, key
    if (biasP) hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, DKV_ARGS); else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, DKV_ARGS);
#undef DKV_ARGS
    return uvit_check_launch();
}
