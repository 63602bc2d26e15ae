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
// Same machinery as attention.hip: [208][64] bf16 LDS images (swizzled 128-B rows) read by rows
// (ds_read_b128) and by columns (ds_read_b64_tr_b16), accumulator tiles reused as MFMA operands,
// log2-unit scores (biasP = bias*log2e, -1e30 in padded key columns), pair-hash dropout.
//   fwd : 13 waves, one query tile each; images Bm, Bc (keys, transformed on the way in), V, CV (LDS-DMA)
//   bwd : ONE fused kernel (round 4): query-side images Am, Ac, dMean, dCov whole, key-side rows in a 2-slot ring, 16-key steps
#include <mutex>
#include <type_traits>
#include "common.h"
#include "uvit_internal.h"

// hand-placed MFMA -> VALU wait states where a branch follows an MFMA chain (tools/check_mfma_hazard.py is the build-time guard;
// -DATTN_NO_HAZARD_PAD builds the deliberately broken variant the guard must flag)
#ifdef ATTN_NO_HAZARD_PAD
#define HAZARD_PAD()
#else
#define HAZARD_PAD() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")
#endif
#define HD 64
#define NT_MAX 13
#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1e30f)

__device__ __forceinline__ int img_off2(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

enum { TR_NONE = 0, TR_SIG = 1, TR_SQRT_SIG = 2 };

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

__device__ __forceinline__ void dma_rows8_2(char* img, int rb, const bf16* src, size_t stride, int row0, int n_valid, int lane) {
    const int row = row0 + 8 * rb + (lane >> 3);
    const int chunk = (lane & 7) ^ (lane >> 3);
    const int r = row < n_valid ? row : n_valid - 1;
    __builtin_amdgcn_global_load_lds(GLB_PTR(void, src + (size_t)r * stride + chunk * 8), LDS_PTR(void, img + rb * 1024), 16, 0, 0);
}
// one 16-B chunk of a token row through the transform: returns the transformed chunk and this chunk's part of the row term
template <int TR>
__device__ __forceinline__ bf16x8 tr_chunk(const bf16x8 v, float pre_scale, float& part) {
    bf16x8 o;
    part = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = bf2f(v[j]);
        if constexpr (TR == TR_SIG) { x = sigm(x * pre_scale); part += x * x; }
        else { x = sigm(x); part += x; x = sqrtf(x); }
        o[j] = f2bf(x);
    }
    return o;
}

// ------------------------------------------------------------------------------------------
// forward (round 4: 13 waves, one query tile each, <= 128 VGPRs; was 7 waves x 2 tiles at ~200 VGPRs)
//
// One 13-wave workgroup per (batch, head).  The key-side operands B_m = sigmoid(k), B_c = sqrt(sigmoid(cov_k)) are transformed on
// the way into their LDS images (4 chunk tasks per lane; the column term c_j leaves through 8-lane DPP sums, no LDS atomics); V and
// cov_v arrive by LDS-DMA; the wave's own query operands A_m, A_c and the row term r_i are built in registers.  Scores of the
// wave's 16 queries against all 13 key tiles stay in registers (52 values per lane, the bias rows are requested into them up
// front), the softmax row is 2 shuffles, P and P^2 feed the two PV products straight from the accumulators.
// ------------------------------------------------------------------------------------------
#define FW2_WAVES 13
#define FW2_ROWS (NT_MAX * 16)
#define FW2_IMG (FW2_ROWS * 128)
#define FW2_LDS (4 * FW2_IMG + FW2_ROWS * 4)

template <bool HAS_BIAS, int NT_C>
__global__ __launch_bounds__(FW2_WAVES * 64)
void attn2_fwd_kernel(const bf16* __restrict__ qkv_m, const bf16* __restrict__ qkv_c, const float* __restrict__ biasP,
                      bf16* __restrict__ out_m, bf16* __restrict__ out_c, float* __restrict__ lse, int H, int N, int NP,
                      float scale, uint32_t drop_thr, float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *bm = smem, *bc = smem + FW2_IMG, *vimg = smem + 2 * FW2_IMG, *cvimg = smem + 3 * FW2_IMG;
    float* cj = (float*)(smem + 4 * FW2_IMG);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const bf16* base_m = qkv_m + (size_t)b * N * ld + h * HD;
    const bf16* base_c = qkv_c + (size_t)b * N * ld + h * HD;
    const int nt = NT_C ? NT_C : (N + 15) >> 4, nt2 = (nt + 1) >> 1;
    const bool active = NT_C ? true : wave < nt;
    const int q = wave * 16 + li, qr = q < N ? q : N - 1;

    // ---- every global request first: V / CV by LDS-DMA, this wave's raw query rows, the raw key chunks of its 4 staging tasks
    for (int p = wave; p < 2 * (FW2_ROWS / 8); p += FW2_WAVES) {
        const int img = p & 1, rb = p >> 1;
        if (rb * 8 >= nt * 16) continue;
        dma_rows8_2(img ? cvimg : vimg, rb, (img ? base_c : base_m) + 2 * C, ld, 0, N, lane);
    }
    // staging tasks: task = wave + 13 j covers (tensor, 8-row block): 2 x 26 row blocks of 8 rows x 8 chunks
    bf16x8 kraw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int task = wave + FW2_WAVES * j, tens = task & 1, rb = task >> 1;
        const int row = rb * 8 + (lane >> 3), rr = row < N ? row : N - 1;
        kraw[j] = *(const bf16x8*)((tens ? base_c : base_m) + C + (size_t)rr * ld + (lane & 7) * 8);
    }
    TokFrags A;
    float ri = 0.f;
    if (active) {
        A = load_tok(base_m + (size_t)qr * ld, base_c + (size_t)qr * ld, g, scale);
        ri = gsum4(A.side);
    }
    for (int i = tid; i < FW2_ROWS; i += FW2_WAVES * 64) cj[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int task = wave + FW2_WAVES * j, tens = task & 1, rb = task >> 1;
        const int row = rb * 8 + (lane >> 3), ch = lane & 7;
        if (rb * 8 >= nt * 16) continue;
        float part;
        bf16x8 o = tens ? tr_chunk<TR_SQRT_SIG>(kraw[j], 1.0f, part) : tr_chunk<TR_SIG>(kraw[j], 1.0f, part);
        if (row >= N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(0.f);
        }
        *(bf16x8*)((tens ? bc : bm) + img_off2(row, ch)) = o;
        part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x101, 0xf, 0xf, true));   // row_shl:1
        part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x102, 0xf, 0xf, true));   // row_shl:2
        part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x104, 0xf, 0xf, true));   // row_shl:4
        if (ch == 0) atomicAdd(&cj[row], part);           // two adders per row (the two tensors): 16 lanes per wave-instruction
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;

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
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float bb;
                if constexpr (HAS_BIAS) bb = s[t][r]; else bb = (t * 16 + 4 * g + r) < N ? 0.f : NEG_BIG;
                const float v = sigm(2.0f * a[r] - ri - cc[r]) * LOG2E + bb;     // sigmoid(-W) + bias, log2 units
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
            const int r1 = t1 < nt ? t1 * 16 : t0 * 16;          // no second tile: its p is 0 (bias -1e30 / zeroed s), re-read the first
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                om[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(vimg, t0 * 16, r1, dt * 16, lane), pf, om[dt], 0, 0, 0);
                oc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(colf(cvimg, t0 * 16, r1, dt * 16, lane), pf2, oc[dt], 0, 0, 0);
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
// backward, FUSED (round 4): every gradient of the two-stream attention from ONE recomputation of P.
//
// One 13-wave workgroup per (batch, head), the base kernel's structure (attention.hip, attn_bwd_fused_kernel) with the
// Wasserstein score and its ten products.  Wave w owns the 16 queries 16 w .. 16 w + 15 ("query on the MFMA lane":
// S^T = B.A^T with A = [m1 | sqrt c1] of the queries and B = [m2 | sqrt c2] of the keys) and the workgroup walks the keys in
// steps of ONE 16-key tile.  Per step i:
//   A_i (every wave): S^T (K = 128), dM.V^T, dC.CV^T of its queries against the step's key tile -> sigmoid(-W), P, PD, dS and
//        gW = dL/dW in registers -> dA^T += B^T.gW^T straight from the accumulators (v_mfma_f32_16x16x16_bf16: the contraction is
//        the step's 16 keys); PD, PD^2 and gW are written ONCE, as bf16, into the step buffers [208 queries][16 keys]; dS leaves
//        for HBM from the registers (8 B per lane, 512 contiguous bytes per wave) for the relative-position-bias gradient;
//   barrier;
//   B_i (waves 0..7, in program order in front of A_{i+1}): the step's key-side gradients, contracted over ALL queries by
//        transposed reads of the four query-side images and of the step buffers -- waves 0..3: dB_m, dB_c (one 16-feature tile
//        each) and the column sum of gW (an all-ones A operand) from the gW buffer, waves 4..7: dV, dCV from PD / PD^2 -- then the
//        chain rule back to the pre-sigmoid / pre-ELU inputs in fp32 and 8-B stores;
//   waves 8..12 meanwhile fill the key ring: the step's 16 rows of k, cov_k are fetched TWO steps ahead into registers,
//        transformed (sigmoid, sqrt sigmoid; the column term c_j on the way) and written one step ahead; v, cov_v by LDS-DMA.
// Why steps of 16 keys: the B phase needs all four query-side operands whole (dM, dC, A_m, A_c: 104 KiB), and P, P^2, gW staged
// for every query of a step; with 32-key steps (the base kernel's) that is 104 + 78 KiB + the key rows.  At 16 keys: 104 KiB +
// 2 x 3 x 6.5 KiB of step buffers + 2 x 8 KiB of key rows = 159 KiB.
// LDS images use the forward's swizzle (img_off2); step buffer rows are 32 B = four 8-B slots (4 keys each), slot s of row q
// at s ^ ((q >> 2) & 3): the 8-B writes of a 16-lane group (16 consecutive rows, one slot) touch every bank once, and a
// transposed read's 32-lane half covers 8 whole rows = 256 contiguous bytes.
// ------------------------------------------------------------------------------------------
#define F2_WAVES 13
#define F2_ROWS (NT_MAX * 16)                 // 208
#define F2_IMG (F2_ROWS * 128)                // 26,624 B
#define F2_SLOT (4 * 16 * 128)                // 8,192 B: Bm, Bc, V, CV rows of one 16-key step
#define F2_SB (F2_ROWS * 32)                  // 6,656 B
#define F2_LDS (4 * F2_IMG + 2 * F2_SLOT + 6 * F2_SB + 2 * 16 * 4)      // 162,944 B
// Diagnostic build only (-DATTN2_STAMP, tools/stamp_attn2.py): s_memtime stamps of waves 0, 4 and 8 of every workgroup; never in libuvit.so
#ifdef ATTN2_STAMP
#define F2S_WG 2048
__device__ unsigned long long g_attn2_stamps[F2S_WG * 3 * 32];
#define F2_WSLOT (wave == 0 ? 0 : wave == 4 ? 1 : wave == 8 ? 2 : -1)
#define F2STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if ((threadIdx.x & 63) == 0 && F2_WSLOT >= 0 && blockIdx.x < F2S_WG) g_attn2_stamps[(blockIdx.x * 3 + F2_WSLOT) * 32 + (k)] = t_; } while (0)
#define F2SUB(k) do { if (i == 6) F2STAMP(k); } while (0)
extern "C" int uvit_debug_attn2_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn2_stamps), sizeof(g_attn2_stamps)) == hipSuccess ? 0 : -3;
}
#else
#define F2STAMP(k)
#define F2SUB(k)
#endif
#ifndef F2_UNROLL
#define F2_UNROLL 7
#endif

__device__ __forceinline__ int sb16_off(int q, int slot) { return q * 32 + ((slot ^ ((q >> 2) & 3)) << 3); }
// B operand of the B phase (K = 32 queries): element j of lane (g, i) = buf[query (j<4 ? r_lo : r_hi) + 4g + (j&3)][key i]
__device__ __forceinline__ bf16x8 sb16_col_frag(const char* sb, int r_lo, int r_hi, bool has_hi, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sb + sb16_off(r_lo + 4 * g + q, p)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sb + sb16_off(r_hi + 4 * g + q, p)));
    if (!has_hi) hi = s16x4{0, 0, 0, 0};
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// A operand of the K = 16 MFMA: lane (g, i) = img[row 4g + e][col0 + i], e = 0..3 (rows = the step's keys)
__device__ __forceinline__ s16x4 col_frag16(const char* img, int col0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + img_off2(4 * g + q, (col0 >> 3) + (p >> 1)) + ((p & 1) << 3)));
}
template <bool HAS_BIAS, int NT_C>
__global__ __launch_bounds__(F2_WAVES * 64)
void attn2_bwd_fused_kernel(const bf16* __restrict__ qkv_m, const bf16* __restrict__ qkv_c, const bf16* __restrict__ o_m,
                            const bf16* __restrict__ o_c, const bf16* __restrict__ d_m, const bf16* __restrict__ d_c,
                            const float* __restrict__ biasP, const float* __restrict__ lse, float* __restrict__ delta,
                            bf16* __restrict__ dqkv_m, bf16* __restrict__ dqkv_c, uint2* __restrict__ ds_out, int H, int N, int NP,
                            float scale, uint32_t drop_thr, float inv_keep, uint32_t drop_key) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qa = smem;                        // A_m: sigmoid(q * scale)
    char* const qc = smem + F2_IMG;               // A_c: sqrt(sigmoid(cov_q))
    char* const dmimg = smem + 2 * F2_IMG;
    char* const dcimg = smem + 3 * F2_IMG;
    char* const ring = smem + 4 * F2_IMG;         // 2 slots x {Bm, Bc, V, CV}[16 rows]
    char* const sbuf = ring + 2 * F2_SLOT;        // 2 x {PD, PD^2, gW}
    float* const cjr = (float*)(sbuf + 6 * F2_SB);  // 2 x 16 column terms
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const int nt = NT_C ? NT_C : (N + 15) >> 4;
    const bf16* base_m = qkv_m + (size_t)b * N * ld + h * HD;
    const bf16* base_c = qkv_c + (size_t)b * N * ld + h * HD;
    const bf16* dmbase = d_m + (size_t)b * N * C + h * HD;
    const bf16* dcbase = d_c + (size_t)b * N * C + h * HD;
    const bool active = NT_C ? true : wave < nt;
    const int q = wave * 16 + li, qr = q < N ? q : N - 1;
    F2STAMP(0);

    // ---- key ring: a step's 16 rows of k, cov_k are 256 chunk tasks (tensor, row, 16-B chunk), one per lane of waves 0..3 (the role with the shortest B chain): fetched
    //      at the start of the iteration before, transformed (sigmoid / sqrt sigmoid, column term c_j on the way) and written at its
    //      end; the step's v, cov_v rows arrive by LDS-DMA (wave 12: 4 pieces of 8 rows)
    auto k_fetch = [&](int step, int lane) -> bf16x8 {
        const int l_t = lane >> 5, l_row = 4 * (wave & 3) + ((lane >> 3) & 3), l_ch = lane & 7;
        const int key = step * 16 + l_row;
        const int kr = key < N ? key : N - 1;
        return *(const bf16x8*)((l_t ? base_c : base_m) + C + (size_t)kr * ld + l_ch * 8);
    };
    auto k_write = [&](int step, const bf16x8 kraw, int lane) {
        const int l_t = lane >> 5, l_row = 4 * (wave & 3) + ((lane >> 3) & 3), l_ch = lane & 7;
        char* slot = ring + (step & 1) * F2_SLOT;
        const int key = step * 16 + l_row;
        float part;
        bf16x8 o = l_t ? tr_chunk<TR_SQRT_SIG>(kraw, 1.0f, part) : tr_chunk<TR_SIG>(kraw, 1.0f, part);
        if (key >= N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(0.f);
        }
        *(bf16x8*)(slot + l_t * 2048 + img_off2(l_row, l_ch)) = o;
        // sum over the row's 8 chunk lanes by DPP row shifts (lane 8 k of a row ends with lanes 8 k .. 8 k + 7), then the other tensor
        part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x101, 0xf, 0xf, true));   // row_shl:1
        part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x102, 0xf, 0xf, true));   // row_shl:2
        part += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, part), 0x104, 0xf, 0xf, true));   // row_shl:4
        part += __shfl_xor(part, 32, 64);
        if ((lane & 39) == 0) cjr[(step & 1) * 16 + l_row] = part;
    };
    auto v_dma = [&](int step, int lane) {
        char* slot = ring + (step & 1) * F2_SLOT;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc)
            dma_rows8_2(slot + 4096 + (pc >> 1) * 2048, pc & 1, ((pc >> 1) ? base_c : base_m) + 2 * C, ld, step * 16, N, lane);
    };

    // ---- preamble: every global request first (dM, dC images by LDS-DMA; this wave's raw q / cov_q rows, its dM / dC / mean / cov
    //      rows for delta; wave 12: step 0 of the key ring), then the transforms
    for (int p = wave; p < 2 * (F2_ROWS / 8); p += F2_WAVES) {
        const int img = p & 1, rb = p >> 1;
        if (rb * 8 >= nt * 16) continue;
        dma_rows8_2(img ? dcimg : dmimg, rb, img ? dcbase : dmbase, (size_t)C, 0, N, lane);
    }
    float ri = 0.f, dl = 0.f, lse_q = 1e30f;        // padded query lanes: p = exp2(.. - 1e30) = 0
    bf16x8 kraw0 = {};
    if (wave < 4) kraw0 = k_fetch(0, lane);
    if (wave == 12) v_dma(0, lane);
    if (active) {
        bf16x8 vm[2], vc[2], dmf[2], dcf[2], omf[2], ocf[2];
        const size_t orow = ((size_t)b * N + qr) * C + h * HD;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = wave * 16 + 8 * k + (lane >> 3);
            const int rr = row < N ? row : N - 1;
            vm[k] = *(const bf16x8*)(base_m + (size_t)rr * ld + (lane & 7) * 8);
            vc[k] = *(const bf16x8*)(base_c + (size_t)rr * ld + (lane & 7) * 8);
            dmf[k] = *(const bf16x8*)(d_m + orow + k * 32 + g * 8); dcf[k] = *(const bf16x8*)(d_c + orow + k * 32 + g * 8);
            omf[k] = *(const bf16x8*)(o_m + orow + k * 32 + g * 8); ocf[k] = *(const bf16x8*)(o_c + orow + k * 32 + g * 8);
        }
        if (q < N) lse_q = lse[(size_t)bh * N + q];
        float side[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = wave * 16 + 8 * k + (lane >> 3), ch = lane & 7;
            float pm_, pc_;
            bf16x8 om_ = tr_chunk<TR_SIG>(vm[k], scale, pm_), oc_ = tr_chunk<TR_SQRT_SIG>(vc[k], 1.0f, pc_);
            if (row >= N) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { om_[j] = f2bf(0.f); oc_[j] = f2bf(0.f); }
            }
            *(bf16x8*)(qa + img_off2(row, ch)) = om_;
            *(bf16x8*)(qc + img_off2(row, ch)) = oc_;
            float s_ = pm_ + pc_;
            s_ += __shfl_xor(s_, 1, 64); s_ += __shfl_xor(s_, 2, 64); s_ += __shfl_xor(s_, 4, 64);
            side[k] = s_;                                              // row term of row 8 k + (lane >> 3)
        }
        // lane (g, li) wants the row term of query li: rows 0..7 sit in side[0] of lanes 8 r, rows 8..15 in side[1]
        const float s0 = __shfl(side[0], (li & 7) * 8, 64), s1 = __shfl(side[1], (li & 7) * 8, 64);
        ri = li < 8 ? s0 : s1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(dmf[kk][j]) * bf2f(omf[kk][j]) + 2.0f * bf2f(dcf[kk][j]) * bf2f(ocf[kk][j]);
        dl = gsum4(dl);                                               // delta_i = dM.mean + 2 dC.cov
        if (q < N && g == 0) delta[(size_t)bh * N + q] = dl;
    }
    if (wave < 4) k_write(0, kraw0, lane);
    float4 bnext;
    auto bias_fetch = [&](int t, int lane) {
        const int g = lane >> 4, q = wave * 16 + (lane & 15);
        if constexpr (HAS_BIAS) {
            const float* brow = biasP + ((size_t)h * NP + q) * NP + 4 * g;      // q < 208 <= NP
            bnext = (t < nt && active) ? *(const float4*)(brow + t * 16) : make_float4(NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG);
        } else {
            const int k0 = t * 16 + 4 * g;
            bnext = make_float4(k0 < N ? 0.f : NEG_BIG, k0 + 1 < N ? 0.f : NEG_BIG, k0 + 2 < N ? 0.f : NEG_BIG, k0 + 3 < N ? 0.f : NEG_BIG);
        }
    };
    bias_fetch(0, lane);
    F2STAMP(1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    F2STAMP(2);
    __syncthreads();
    F2STAMP(3);

    f32x4 dam[4], dac[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dam[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dac[dt] = dam[dt]; }
    float dside = 0.f;

    // ================= A_i: this wave's 16 queries against the step's key tile
    auto A_step = [&](int i, int lane) __attribute__((always_inline)) {
        // per-lane offsets are re-derived from an opaque copy of the lane id every step: hoisted out of the loop they cost ~20 VGPRs
        const int g = lane >> 4, li = lane & 15, q = wave * 16 + li;
        const char* slot = ring + (i & 1) * F2_SLOT;
        const char *bm = slot, *bc = slot + 2048, *vimg = slot + 4096, *cvimg = slot + 6144;
        char* pbuf = sbuf + (i & 1) * 3 * F2_SB;
        const float4 bcur = bnext;
        if (i + 1 < nt) bias_fetch(i + 1, lane);
        // the dropout draw comes BEFORE the MFMAs: no wave-uniform branch between an MFMA and the first VALU read of its result
        // (hipcc pads the MFMA -> VALU wait states along the fall-through path only; attention.hip)
        bool k4[4] = {true, true, true, true};
        if (drop_thr) keep4b(drop_key, ((uint32_t)bh * N + q) * (uint32_t)(NP >> 1), i * 16 + 4 * g, drop_thr, k4);
        const float4 cv4 = *(const float4*)(cjr + (i & 1) * 16 + 4 * g);
        F2SUB(24);
        f32x4 sacc, pm, pc;
        {
            const bf16x8 k0 = rowf(bm, li, g), k1 = rowf(bm, li, 4 + g);
            const bf16x8 a0 = rowf(qa, q, g), a1 = rowf(qa, q, 4 + g);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, a0, z, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, a1, sacc, 0, 0, 0);
        }
        {
            const bf16x8 k0 = rowf(bc, li, g), k1 = rowf(bc, li, 4 + g);
            const bf16x8 a0 = rowf(qc, q, g), a1 = rowf(qc, q, 4 + g);
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, a0, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, a1, sacc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        F2SUB(25);
        {
            const bf16x8 v0 = rowf(vimg, li, g), v1 = rowf(vimg, li, 4 + g);
            const bf16x8 d0 = rowf(dmimg, q, g), d1 = rowf(dmimg, q, 4 + g);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, d0, z, 0, 0, 0);
            pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, d1, pm, 0, 0, 0);
        }
        {
            const bf16x8 v0 = rowf(cvimg, li, g), v1 = rowf(cvimg, li, 4 + g);
            const bf16x8 d0 = rowf(dcimg, q, g), d1 = rowf(dcimg, q, 4 + g);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            pc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, d0, z, 0, 0, 0);
            pc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, d1, pc, 0, 0, 0);
        }
        F2SUB(26);
        const float bb[4] = {bcur.x, bcur.y, bcur.z, bcur.w};
        const float cc[4] = {cv4.x, cv4.y, cv4.z, cv4.w};
        float pdv[4], p2v[4], gwv[4], dsv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sg = sigm(2.0f * sacc[r] - ri - cc[r]);                       // sigmoid(-W)
            const float p = __builtin_amdgcn_exp2f(sg * LOG2E + bb[r] - lse_q);
            const float pd = k4[r] ? p * inv_keep : 0.f;
            // dPD = dM.v + 2 PD (dC.cv);  dP = D dPD;  dS = P (dP - delta);  dL/dW = -dS sg (1 - sg)
            const float dpd = pm[r] + 2.0f * pd * pc[r];
            const float ds = p * ((k4[r] ? dpd * inv_keep : 0.f) - dl);
            const float gv = -ds * sg * (1.0f - sg);
            pdv[r] = pd; p2v[r] = pd * pd; gwv[r] = gv; dsv[r] = ds;
            dside += gv;
        }
        const bf16x4 pv = {f2bf(pdv[0]), f2bf(pdv[1]), f2bf(pdv[2]), f2bf(pdv[3])};
        const bf16x4 p2 = {f2bf(p2v[0]), f2bf(p2v[1]), f2bf(p2v[2]), f2bf(p2v[3])};
        const bf16x4 gw = {f2bf(gwv[0]), f2bf(gwv[1]), f2bf(gwv[2]), f2bf(gwv[3])};
        *(bf16x4*)(pbuf + sb16_off(q, g)) = pv;
        *(bf16x4*)(pbuf + F2_SB + sb16_off(q, g)) = p2;
        *(bf16x4*)(pbuf + 2 * F2_SB + sb16_off(q, g)) = gw;
        if (ds_out) {
            const bf16x4 dv = {f2bf(dsv[0]), f2bf(dsv[1]), f2bf(dsv[2]), f2bf(dsv[3])};
            ds_out[(((size_t)bh * nt + i) * nt + wave) * 64 + lane] = __builtin_bit_cast(uint2, dv);
        }
        const s16x4 gwf = __builtin_bit_cast(s16x4, gw);
        __builtin_amdgcn_sched_barrier(0);
        F2SUB(27);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dam[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(col_frag16(bm, dt * 16, lane), gwf, dam[dt], 0, 0, 0);
            dac[dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(col_frag16(bc, dt * 16, lane), gwf, dac[dt], 0, 0, 0);
        }
        F2SUB(28);
    };

    // ================= B_i (waves 0..11): key-side gradients of the step's 16 keys, contracted over every query.
    // role 0 (waves 0..3): dB_m tile + the column sum of gW;  role 1 (waves 4..7): dB_c tile + the same column sum;
    // role 2 (waves 8..11): dV and dCV tiles.  (The column sum d c_j = sum_i gW_ij is an all-ones A operand against the gW
    // fragments the wave reads anyway: cheaper than handing it from one wave to another.)
    const int dtj = wave & 3;
    bf16x4 xnext = {};                    // the raw input B_i's chain rule needs, requested a whole iteration ahead
    auto x_fetch = [&](int i, int lane) {
        const int g = lane >> 4, key = i * 16 + (lane & 15);
        const int kr = key < N ? key : N - 1;
        const int role = wave >> 2;
        xnext = *(const bf16x4*)((role == 0 ? base_m + C : role == 1 ? base_c + C : base_c + 2 * C) + (size_t)kr * ld + dtj * 16 + 4 * g);
    };
    auto B_step = [&](auto role_c, int i, int lane, const bf16x4 x0) __attribute__((always_inline)) {
        constexpr int role = decltype(role_c)::value;
        const int g = lane >> 4, li = lane & 15;
        const char* pbuf = sbuf + (i & 1) * 3 * F2_SB;
        const int key = i * 16 + li;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const char* img0 = role == 0 ? qa : role == 1 ? qc : dmimg;
        const char* sb0 = role == 2 ? pbuf : pbuf + 2 * F2_SB;
        const bf16 one = f2bf(1.0f);
        const bf16x8 ones = {one, one, one, one, one, one, one, one};
        auto kstep = [&](int ks) __attribute__((always_inline)) {
            const bool hk = 2 * ks + 1 < nt;
            const int r_lo = 32 * ks, r_hi = hk ? r_lo + 16 : r_lo;
            const bf16x8 a0 = colf(img0, r_lo, r_hi, dtj * 16, lane);
            const bf16x8 b0 = sb16_col_frag(sb0, r_lo, r_hi, hk, lane);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc0, 0, 0, 0);
            if constexpr (role == 2) {
                const bf16x8 a1 = colf(dcimg, r_lo, r_hi, dtj * 16, lane);
                const bf16x8 b1 = sb16_col_frag(pbuf + F2_SB, r_lo, r_hi, hk, lane);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc1, 0, 0, 0);
            } else {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b0, acc1, 0, 0, 0);
            }
        };
        if constexpr (NT_C != 0) {
#pragma unroll F2_UNROLL
            for (int ks = 0; ks < (NT_C + 1) / 2; ++ks) kstep(ks);
        } else {
            const int nk = (nt + 1) >> 1;
            for (int ks = 0; ks < nk; ++ks) kstep(ks);
        }
        HAZARD_PAD();     // loop exit / exec-mask branch: pad the MFMA -> VALU wait states by hand
        if (i == 5) F2STAMP(22);
        // acc*[r] = gradient [key li][feature 16 dtj + 4 g + r];  roles 0, 1: acc1[.] = d c_j = sum_i dL/dW_ij
        if (key < N) {
            const size_t o = ((size_t)b * N + key) * ld + h * HD + dtj * 16 + 4 * g;
            bf16x4 r0, r1;
            if constexpr (role == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) r0[r] = f2bf(back_mean(-2.0f * acc0[r], acc1[0], bf2f(x0[r]), 1.0f));
                *(bf16x4*)(dqkv_m + o + C) = r0;
            } else if constexpr (role == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) r0[r] = f2bf(back_cov(-2.0f * acc0[r], acc1[0], bf2f(x0[r])));
                *(bf16x4*)(dqkv_c + o + C) = r0;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    r0[r] = f2bf(acc0[r]);
                    r1[r] = f2bf(acc1[r] * fminf(bf2f(x0[r]), 1.0f));                // ELU' of the cov_v pre-activation
                }
                *(bf16x4*)(dqkv_m + o + 2 * C) = r0;
                *(bf16x4*)(dqkv_c + o + 2 * C) = r1;
            }
        }
    };
    // the role is wave-uniform but a run-time value: three instantiations, chosen in front of the chain (a branch BETWEEN the MFMAs of
    // one chain is what tools/check_mfma_hazard.py flagged in the single-instantiation form)
    auto B_any = [&](int i, int lane, const bf16x4 x0) __attribute__((always_inline)) {
        if (wave < 4) B_step(std::integral_constant<int, 0>{}, i, lane, x0);
        else if (wave < 8) B_step(std::integral_constant<int, 1>{}, i, lane, x0);
        else B_step(std::integral_constant<int, 2>{}, i, lane, x0);
    };
    if (wave < F2_WAVES - 1) x_fetch(0, lane);

    // iteration i: B_{i-1} (waves 0..11), then A_i, then (waves 0..3) the transformed key rows of step i + 1, then the step's barrier.
    // Only the LDS traffic is waited for at the barrier (plus wave 12's DMA): the dS and gradient stores drain behind the next steps.
#pragma unroll 1
    for (int i = 0; i < nt; ++i) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        F2SUB(21);
        const bool more = i + 1 < nt;
        if (wave < F2_WAVES - 1) {
            const bf16x4 xcur = xnext;
            x_fetch(i, ln);                                   // for B_i, one iteration from now
            bf16x8 kraw = {};
            if (wave < 4 && more) kraw = k_fetch(i + 1, ln);
            if (i > 0) B_any(i - 1, ln, xcur);
            __builtin_amdgcn_sched_barrier(0);
            F2SUB(23);
            if (active) A_step(i, ln);
            __builtin_amdgcn_sched_barrier(0);
            if (wave < 4 && more) k_write(i + 1, kraw, ln);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            if (more) v_dma(i + 1, ln);
            __builtin_amdgcn_sched_barrier(0);
            F2SUB(23);
            if (active) A_step(i, ln);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        F2SUB(29);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        F2SUB(30);
        F2STAMP(4 + (i < 14 ? i : 14));
    }
    // ---- tail: the raw q / cov_q values the query-side chain rule needs are requested in front of the last B step
    bf16x4 xq[4], xcq[4];
    if (active && q < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            xq[dt] = *(const bf16x4*)(base_m + (size_t)q * ld + dt * 16 + 4 * g);
            xcq[dt] = *(const bf16x4*)(base_c + (size_t)q * ld + dt * 16 + 4 * g);
        }
    }
    if (wave < F2_WAVES - 1) B_any(nt - 1, lane, xnext);
    F2STAMP(18);
    if (active) {
        dside = gsum4(dside);                                  // d r_i = sum_j dL/dW_ij
        if (q < N) {
            bf16* om = dqkv_m + ((size_t)b * N + q) * ld + h * HD + 4 * g;
            bf16* oc = dqkv_c + ((size_t)b * N + q) * ld + h * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 rm, rc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {       // dA = -2 sum_j gW B_j
                    rm[r] = f2bf(back_mean(-2.0f * dam[dt][r], dside, bf2f(xq[dt][r]), scale));
                    rc[r] = f2bf(back_cov(-2.0f * dac[dt][r], dside, bf2f(xcq[dt][r])));
                }
                *(bf16x4*)(om + dt * 16) = rm;
                *(bf16x4*)(oc + dt * 16) = rc;
            }
        }
    }
    F2STAMP(19);
}

// Bias gradient from the dS tiles the fused kernel streamed out: slab[h][key][q] += sum_b dS_b[h][q][key].
// ds = [B * H][nt key tiles][nt query tiles][64 lanes] x 8 B: lane (g, li) of tile (t, w) holds dS[q = 16 w + li][keys 16 t + 4 g ..+3].
// One 256-thread workgroup per (head, key tile, 4 query tiles, batch part): wave k streams tile (t, 4 wq + k) of every sample of the
// part (512 contiguous bytes per wave and sample), the [16 keys][64 q] partial tile is turned through LDS and added with one fp32
// atomic per lane, 256 contiguous bytes per wave-instruction.
#define DBR2_PARTS 4
__global__ __launch_bounds__(256)
void attn2_dbias_reduce_kernel(const uint2* __restrict__ ds, float* __restrict__ slab, int B, int H, int N, int NP) {
    __shared__ float tile[16][65];
    const int nt = (N + 15) >> 4, nq4 = (nt + 3) >> 2;
    const int wq = blockIdx.x % nq4, ht = blockIdx.x / nq4, t = ht % nt, h = ht / nt;
    const int tid = threadIdx.x, lane = tid & 63, k = tid >> 6, g = lane >> 4, li = lane & 15;
    const int w = 4 * wq + k;
    const int per = (B + DBR2_PARTS - 1) / DBR2_PARTS, b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (w < nt) {
        const uint2* p = ds + (((size_t)h * nt + t) * nt + w) * 64 + lane;
        const size_t bstride = (size_t)H * nt * nt * 64;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) {
            const bf16x4 v = __builtin_bit_cast(bf16x4, p[b * bstride]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += bf2f(v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[4 * g + j][16 * k + li] = acc[j];
    __syncthreads();
    const int qo = 64 * wq + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = 16 * t + 4 * k + j;
        if (key < N && qo < N) atomicAdd(slab + ((size_t)h * NP + key) * NP + qo, tile[4 * k + j][lane]);
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static std::once_flag g_attr2_once;
static void init2_impl() {
#define SETA(K, B) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, B)
    SETA((attn2_fwd_kernel<true, 0>), FW2_LDS); SETA((attn2_fwd_kernel<false, 0>), FW2_LDS);
    SETA((attn2_fwd_kernel<true, NT_MAX>), FW2_LDS); SETA((attn2_fwd_kernel<false, NT_MAX>), FW2_LDS);
    SETA((attn2_bwd_fused_kernel<true, 0>), F2_LDS); SETA((attn2_bwd_fused_kernel<false, 0>), F2_LDS);
    SETA((attn2_bwd_fused_kernel<true, NT_MAX>), F2_LDS); SETA((attn2_bwd_fused_kernel<false, NT_MAX>), F2_LDS);
#undef SETA
}
static void init2() { std::call_once(g_attr2_once, init2_impl); }

int uvit_attn2_fwd_launch(const void* qkv_m, const void* qkv_c, const float* biasP, void* out_m, void* out_c, float* lse,
                          int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s) {
    if (B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16) return UVIT_ERR_SHAPE;
    if (NP < NT_MAX * 16 || (NP & 3)) return UVIT_ERR_ARG;
    init2();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
#define FA dim3(B * H), dim3(FW2_WAVES * 64), FW2_LDS, s, (const bf16*)qkv_m, (const bf16*)qkv_c, biasP, (bf16*)out_m, (bf16*)out_c, lse, \
        H, N, NP, scale, thr, inv_keep, uvit_layer_key(seed, layer)
    if ((N + 15) / 16 == NT_MAX) {
        if (biasP) hipLaunchKernelGGL((attn2_fwd_kernel<true, NT_MAX>), FA); else hipLaunchKernelGGL((attn2_fwd_kernel<false, NT_MAX>), FA);
    } else {
        if (biasP) hipLaunchKernelGGL((attn2_fwd_kernel<true, 0>), FA); else hipLaunchKernelGGL((attn2_fwd_kernel<false, 0>), FA);
    }
#undef FA
    return uvit_check_launch();
}

size_t uvit_attn2_bwd_ws_bytes(int B, int H, int N) {
    const size_t nt = (size_t)(N + 15) / 16;
    return (size_t)B * H * nt * nt * 512;
}

// ds_ws: uvit_attn2_bwd_ws_bytes(B, H, N) bytes of bf16 dS tiles, written when want_ds != 0 (the bias gradient needs them)
int uvit_attn2_bwd_launch(const void* qkv_m, const void* qkv_c, const void* o_m, const void* o_c, const void* d_m, const void* d_c,
                          const float* biasP, const float* lse, float* delta, void* dqkv_m, void* dqkv_c, void* ds_ws, int want_ds,
                          int B, int H, int N, int NP, float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s) {
    if (B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16) return UVIT_ERR_SHAPE;
    if (NP < NT_MAX * 16 || (NP & 3) || (want_ds && !ds_ws)) return UVIT_ERR_ARG;
    init2();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t key = uvit_layer_key(seed, layer);
    uint2* dsw = want_ds ? (uint2*)ds_ws : nullptr;
#define BA dim3(B * H), dim3(F2_WAVES * 64), F2_LDS, s, (const bf16*)qkv_m, (const bf16*)qkv_c, (const bf16*)o_m, (const bf16*)o_c, \
        (const bf16*)d_m, (const bf16*)d_c, biasP, lse, delta, (bf16*)dqkv_m, (bf16*)dqkv_c, dsw, H, N, NP, scale, thr, inv_keep, key
    if ((N + 15) / 16 == NT_MAX) {
        if (biasP) hipLaunchKernelGGL((attn2_bwd_fused_kernel<true, NT_MAX>), BA); else hipLaunchKernelGGL((attn2_bwd_fused_kernel<false, NT_MAX>), BA);
    } else {
        if (biasP) hipLaunchKernelGGL((attn2_bwd_fused_kernel<true, 0>), BA); else hipLaunchKernelGGL((attn2_bwd_fused_kernel<false, 0>), BA);
    }
#undef BA
    return uvit_check_launch();
}

// dbias_slab = ONE [H][NP][NP] slab laid out [h][key][q]; accumulate = 0 overwrites it (zero fill first), 1 adds
int uvit_attn2_dbias_reduce_launch(const void* ds_ws, float* dbias_slab, int accumulate, int B, int H, int N, int NP, hipStream_t s) {
    if (B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16) return UVIT_ERR_SHAPE;
    if (!ds_ws || !dbias_slab || NP < NT_MAX * 16) return UVIT_ERR_ARG;
    if (!accumulate) { int rc = uvit_zero_launch(dbias_slab, (size_t)H * NP * NP * sizeof(float), s); if (rc) return rc; }
    const int nt = (N + 15) / 16, nq4 = (nt + 3) / 4;
    hipLaunchKernelGGL(attn2_dbias_reduce_kernel, dim3(H * nt * nq4, DBR2_PARTS), dim3(256), 0, s, (const uint2*)ds_ws, dbias_slab, B, H, N, NP);
    return uvit_check_launch();
}

#ifdef ATTN2_STAMP
extern "C" int uvit_debug_attn2_bwd(const void* qkv_m, const void* qkv_c, const void* o_m, const void* o_c, const void* d_m, const void* d_c,
                                    const float* biasP, const float* lse, float* delta, void* dqkv_m, void* dqkv_c, void* ds_ws, int B, int H,
                                    int N, float p_drop, void* stream) {
    return uvit_attn2_bwd_launch(qkv_m, qkv_c, o_m, o_c, d_m, d_c, biasP, lse, delta, dqkv_m, dqkv_c, ds_ws, ds_ws != nullptr, B, H, N, 208,
                                 0.125f, p_drop, 1u, 0u, (hipStream_t)stream);
}
#endif
