// bf16 MFMA GEMMs for the ViT step, gfx950.
//
//   NT :  C[M,N]  = A[M,K] . W[N,K]^T          (forward Linear, and dgrad with a W^T copy)
//   TN :  C[N,K]  = dY[M,N]^T . X[M,K]         (wgrad; reduction over the token dimension)
//
// Both use a 128x128 block tile, BK = 64, 4 waves (2x2) of 64x64 each, 16x16x32 bf16 MFMA with
// fp32 accumulation, operands staged HBM -> LDS with global_load_lds (16 B/lane, no VGPR round
// trip), two LDS stages, XOR-swizzled so the fragment reads are bank-conflict free:
//   NT  tiles are [rows][64 k]  (128-B rows), read with ds_read_b128;
//   TN  tiles are [64 m][128 c] (256-B rows), read with ds_read_b64_tr_b16 (hardware transpose).
// The MFMA is issued "swapped" (weight rows as the A operand) so that each lane ends up with 4
// consecutive output columns of one output row -> 8/16-byte vector epilogue loads and stores.
// Epilogues fuse bias, GELU, LayerScale*DropPath*residual, GELU', mask-token blend.
#include <mutex>
#include <cstdlib>

#include "common.h"
#include "uvit_internal.h"

#define BM 128
#define BN 128
#define BK 64
#define GEMM_THREADS 256
#define STAGE_BYTES (BM * BK * 2)   // 16 KiB per operand per stage

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(void, g), LDS_PTR(void, lds_wave_base), 16, 0, 0);
}

// ------------------------------------------------------------------------------------------
// epilogue: acc[j] holds C[m][n..n+3].  Per-column operands (bias, gamma) are loaded once per
// column group into a ColVals before the row loops (no dependent load chain in the store tail).
// ------------------------------------------------------------------------------------------
struct ColVals { float4 b; float4 g; };

template <int MODE>
__device__ __forceinline__ ColVals load_cols(const GemmEpi& e, int n, int N) {
    ColVals c;
    c.b = make_float4(0.f, 0.f, 0.f, 0.f);
    c.g = c.b;
    if constexpr (MODE == EPI_QKV || MODE == EPI_QKV_ELU) {
        // bias = cat(q_bias, 0, v_bias)  (modeling_finetune.py:149-151; cov stream: modeling_finetune_dist.py:116-127)
        const int C = N / 3;
        if (n < C) c.b = *(const float4*)(e.bias + n);
        else if (n >= 2 * C) c.b = *(const float4*)(e.bias2 + (n - 2 * C));
    } else if constexpr (MODE == EPI_BF16 || MODE == EPI_F32) {
        if (e.bias) c.b = *(const float4*)(e.bias + n);
    } else if constexpr (MODE == EPI_GELU || MODE == EPI_GELU_DG || MODE == EPI_PATCH) {
        c.b = *(const float4*)(e.bias + n);
        if constexpr (MODE == EPI_PATCH) c.g = *(const float4*)(e.mask_token + n);
    } else if constexpr (MODE == EPI_RESID) {
        c.b = *(const float4*)(e.bias + n);
        c.g = *(const float4*)(e.gamma + n);
    }
    return c;
}

// Exact (erf) GELU and its derivative for bf16 outputs, two elements per instruction.
// The fused GELU epilogues are VALU-bound: in-kernel stamps of the fc1 launch (tools/stamp_gemm.py) put 27.5k of a tile's 63k
// cycles in the GELU + GELU' epilogue against 7k for a plain bf16 store, at 4 cycles per wave-instruction and two waves per
// SIMD.  So the formulation minimises instruction ISSUES: no transcendental (v_rcp / v_exp cost two issue slots each), and
// everything on v_pk_fma_f32 / v_pk_mul_f32, which process two fp32 values per lane in one slot:
//     xc = clamp(x, -4, 4), u = xc^2
//     Phi(x)   ~ 0.5 + xc P(u)          (P, R: degree-8 least-squares fits on Chebyshev nodes of [0, 16], tools/fit_gelu.py)
//     gelu(x)  = x Phi(xc)              (x > 4: x (1 - 3e-5);  x < -4: |x| 3e-5)
//     gelu'(x) ~ 0.5 + xc R(u)          (R fits (Phi(x) - 0.5) / x + phi(x), i.e. gelu' = Phi + x phi directly)
// max |error| on [-4, 4], fp32 Horner: Phi 6.3e-6 (2.6e-5 with the clamp tail), gelu 2.5e-5, gelu' 8.0e-5; beyond the clamp gelu <= 2.1e-4 (|x| <= 8),
// gelu' <= 5.5e-4 -- all far below the bf16 half-ulp of the outputs (2e-3 at 1).
// Issue slots per element: clamp 1 + u 0.5 + P 4 + Phi 0.5 + gelu 0.5 = 6.5; gelu' + R 4 + 0.5 = 11 for both
// (round 2 first half: Abramowitz-Stegun 7.1.25 with v_rcp + v_exp: 13 / 18 slots; round 1: 16 / 20 ops + 2 transcendentals).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define GELU_HORNER(acc, u, c8, c7, c6, c5, c4, c3, c2, c1, c0) do { \
        acc = f32x2{c8, c8}; \
        acc = __builtin_elementwise_fma(acc, u, f32x2{c7, c7}); acc = __builtin_elementwise_fma(acc, u, f32x2{c6, c6}); \
        acc = __builtin_elementwise_fma(acc, u, f32x2{c5, c5}); acc = __builtin_elementwise_fma(acc, u, f32x2{c4, c4}); \
        acc = __builtin_elementwise_fma(acc, u, f32x2{c3, c3}); acc = __builtin_elementwise_fma(acc, u, f32x2{c2, c2}); \
        acc = __builtin_elementwise_fma(acc, u, f32x2{c1, c1}); acc = __builtin_elementwise_fma(acc, u, f32x2{c0, c0}); } while (0)
#define GELU_P(acc, u) GELU_HORNER(acc, u, 8.08658365e-11f, -7.02476654e-09f, 2.72402532e-07f, -6.31025083e-06f, 9.90762748e-05f, \
                                   -1.13498475e-03f, 9.88112355e-03f, -6.64164935e-02f, 3.98925811e-01f)
#define GELU_R(acc, u) GELU_HORNER(acc, u, 9.78443275e-10f, -8.21882605e-08f, 3.03107588e-06f, -6.50489355e-05f, 9.08658221e-04f, \
                                   -8.72635215e-03f, 5.84950522e-02f, -2.64895682e-01f, 7.97648736e-01f)
__device__ __forceinline__ f32x2 gelu_clamp2(f32x2 x) {
    return f32x2{__builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f), __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f)};
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 xc = gelu_clamp2(x), u = xc * xc;
    f32x2 p; GELU_P(p, u);
    return x * __builtin_elementwise_fma(xc, p, f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    const f32x2 xc = gelu_clamp2(x), u = xc * xc;
    f32x2 r; GELU_R(r, u);
    return __builtin_elementwise_fma(xc, r, f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ void gelu_and_grad_fast2(f32x2 x, f32x2& gelu, f32x2& grad) {
    const f32x2 xc = gelu_clamp2(x), u = xc * xc;
    f32x2 p, r; GELU_P(p, u); GELU_R(r, u);
    gelu = x * __builtin_elementwise_fma(xc, p, f32x2{0.5f, 0.5f});
    grad = __builtin_elementwise_fma(xc, r, f32x2{0.5f, 0.5f});
}

// ------------------------------------------------------------------------------------------
// LDS-staged epilogue.  The MFMA result layout gives a lane 4 consecutive columns of one row, i.e. 8-byte
// bf16 stores in 32..64-byte row segments: the store path, not HBM, then sets the epilogue time.  Instead a
// wave re-lays every 16-row x 64-column block of its accumulators through a private LDS slot (no barrier:
// one wave's LDS operations execute in order) and touches global memory with 16 B per lane, 128 or 256
// contiguous bytes per row:
//   bf16-staged (BF16, QKV, QKV_ELU, GELU): values are finished (bias, ELU) and rounded before staging;
//       slot = [16 rows][128 B], 16-B chunk c of row r at c ^ (r & 7), 8-B halves swapped for rows >= 8;
//       read-back lane -> row 8i + lane/8, columns 8 (lane % 8) .. +7.
//   fp32-staged (RESID, F32, PATCH: 4 columns per lane; DGELU: 8 columns per lane): raw accumulators;
//       slot = [16 rows][256 B], chunk c of row r at c ^ r.
// Both images are bank-conflict free for the ds_write / ds_read_b128 lane groups of gfx950.
// ------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ constexpr bool stage_f32() {
    return MODE == EPI_RESID || MODE == EPI_F32 || MODE == EPI_PATCH || MODE == EPI_DGELU || MODE == EPI_MULAUX;
}
#define EPI_SLOT_BF16 2048
#define EPI_SLOT_F32 4096
#define EPI_WAVE_BYTES 16384       // 8 bf16 blocks or 4 fp32 blocks in flight per wave

template <int MODE>
struct EpiCols {
    float4 pre[4];   // bf16-staged: bias at the MFMA-layout columns ct*16 + 4g .. +3
    float4 b, g;     // fp32-staged, 4 columns per lane: bias, and gamma (RESID) / mask token (PATCH), at the read-back columns
};

template <int MODE>
__device__ __forceinline__ EpiCols<MODE> epi_cols(const GemmEpi& e, int nb, int lane, int N) {
    EpiCols<MODE> c;
    c.b = make_float4(0.f, 0.f, 0.f, 0.f);
    c.g = c.b;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) c.pre[ct] = c.b;
    if constexpr (!stage_f32<MODE>()) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int n = nb + ct * 16 + 4 * (lane >> 4);
            c.pre[ct] = load_cols<MODE>(e, n < N ? n : 0, N).b;
        }
    } else if constexpr (MODE != EPI_DGELU && MODE != EPI_MULAUX) {
        const int n = nb + 4 * (lane & 15);
        const ColVals v = load_cols<MODE>(e, n < N ? n : 0, N);
        c.b = v.b; c.g = v.g;
    }
    return c;
}

// write one 16x64 block (a[ct] = columns ct*16 + 4g .. +3 of row li) into its slot
template <int MODE>
__device__ __forceinline__ void epi_stage(const EpiCols<MODE>& c, char* slot, int lane, const f32x4 a0, const f32x4 a1,
                                          const f32x4 a2, const f32x4 a3) {
    const int g = lane >> 4, li = lane & 15;
    const f32x4 a[4] = {a0, a1, a2, a3};
    if constexpr (!stage_f32<MODE>()) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float x0 = a[ct][0] + c.pre[ct].x, x1 = a[ct][1] + c.pre[ct].y, x2 = a[ct][2] + c.pre[ct].z, x3 = a[ct][3] + c.pre[ct].w;
            if constexpr (MODE == EPI_QKV_ELU) {
                // cov_qkv = ELU(x) + 1  (= x + 1 for x > 0, exp(x) otherwise)
                x0 = x0 > 0.f ? x0 + 1.f : __expf(x0); x1 = x1 > 0.f ? x1 + 1.f : __expf(x1);
                x2 = x2 > 0.f ? x2 + 1.f : __expf(x2); x3 = x3 > 0.f ? x3 + 1.f : __expf(x3);
            }
            const bf16x4 v = {f2bf(x0), f2bf(x1), f2bf(x2), f2bf(x3)};
            const int chunk = (ct * 2 + (g >> 1)) ^ (li & 7);
            const int half = (g & 1) ^ (li >> 3);
            *(bf16x4*)(slot + li * 128 + chunk * 16 + half * 8) = v;
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            *(f32x4*)(slot + li * 256 + (((ct * 4 + g) ^ li) << 4)) = a[ct];
    }
}

__device__ __forceinline__ void epi_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// row-indexed epilogue operands of one block (DGELU: h, RESID: the residual rows), requested before the block is
// staged so that their latency overlaps the LDS round trip instead of following it
template <int MODE>
struct EpiPre { bf16x8 h[2]; float4 r[4]; int mrow[4]; };     // mrow: RESID over a row list: the residual-stream row of each read-back row (-1: padding)

template <int MODE>
__device__ __forceinline__ EpiPre<MODE> epi_prefetch(const GemmEpi& e, int lane, int mb, int nb, int M, int N) {
    EpiPre<MODE> p;
    if constexpr (MODE == EPI_DGELU || MODE == EPI_MULAUX) {
        const int n = nb + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int m = mb + 8 * i + (lane >> 3); m = m < M ? m : M - 1;
            p.h[i] = *(const bf16x8*)((const bf16*)e.aux + (size_t)m * e.ldo + (n < N ? n : 0));
        }
    } else if constexpr (MODE == EPI_RESID) {
        const int n = nb + (lane & 15) * 4;
        const int nvalid = e.rowmap ? min(*e.rowcount, M) : M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = mb + 4 * i + (lane >> 4);
            p.mrow[i] = m;
            if (e.rowmap) { p.mrow[i] = m < nvalid ? e.rowmap[m] : -1; m = p.mrow[i] < 0 ? 0 : p.mrow[i]; }
            else m = m < M ? m : M - 1;
            p.r[i] = *(const float4*)(e.resid + (size_t)m * e.ldo + (n < N ? n : 0));
        }
    }
    return p;
}

// read one staged block back row-contiguously and finish it: block rows mb.., block columns nb..
template <int MODE>
__device__ __forceinline__ void epi_flush(const GemmEpi& e, const EpiCols<MODE>& c, const char* slot, int lane,
                                          int mb, int nb, int M, int N, const EpiPre<MODE>& pre) {
    if constexpr (!stage_f32<MODE>()) {
        const int cc = lane & 7, n = nb + cc * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * i + (lane >> 3), m = mb + row;
            bf16x8 v = *(const bf16x8*)(slot + row * 128 + ((cc ^ (lane >> 3)) << 4));
            if (i == 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
            if (m >= M || n >= N) continue;
            const size_t o = (size_t)m * e.ldo + n;
            if constexpr (MODE == EPI_GELU) {
                // the pre-activation is kept in bf16 for backward; GELU is evaluated on the rounded value so
                // forward and backward see the same h
                if (e.out2) *(bf16x8*)((bf16*)e.out2 + o) = v;
                bf16x8 av;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2 ge = gelu_fast2(f32x2{bf2f(v[j]), bf2f(v[j + 1])});
                    av[j] = f2bf(ge.x); av[j + 1] = f2bf(ge.y);
                }
                *(bf16x8*)((bf16*)e.out + o) = av;
            } else if constexpr (MODE == EPI_GELU_DG) {
                // gelu(h) and gelu'(h) from one erf / exp evaluation of the bf16-rounded h; backward multiplies by out2
                bf16x8 av, dv;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2 ge, gr; gelu_and_grad_fast2(f32x2{bf2f(v[j]), bf2f(v[j + 1])}, ge, gr);
                    av[j] = f2bf(ge.x); av[j + 1] = f2bf(ge.y);
                    dv[j] = f2bf(gr.x); dv[j + 1] = f2bf(gr.y);
                }
                if (e.out2) *(bf16x8*)((bf16*)e.out2 + o) = dv;
                *(bf16x8*)((bf16*)e.out + o) = av;
            } else {
                *(bf16x8*)((bf16*)e.out + o) = v;
            }
        }
    } else if constexpr (MODE == EPI_DGELU || MODE == EPI_MULAUX) {
        const int cc = lane & 7, n = nb + cc * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * i + (lane >> 3), m = mb + row;
            const f32x4 lo = *(const f32x4*)(slot + row * 256 + (((2 * cc) ^ row) << 4));
            const f32x4 hi = *(const f32x4*)(slot + row * 256 + (((2 * cc + 1) ^ row) << 4));
            if (m >= M || n >= N) continue;
            const size_t o = (size_t)m * e.ldo + n;
            const bf16x8 h = pre.h[i];
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                f32x2 ml = {bf2f(h[j]), bf2f(h[j + 1])}, mh = {bf2f(h[4 + j]), bf2f(h[5 + j])};
                if constexpr (MODE == EPI_DGELU) { ml = gelu_grad_fast2(ml); mh = gelu_grad_fast2(mh); }
                v[j] = f2bf(lo[j] * ml.x); v[j + 1] = f2bf(lo[j + 1] * ml.y);
                v[4 + j] = f2bf(hi[j] * mh.x); v[5 + j] = f2bf(hi[j + 1] * mh.y);
            }
            *(bf16x8*)((bf16*)e.out + o) = v;
        }
    } else {
        const int cc = lane & 15, n = nb + cc * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * i + (lane >> 4), m = mb + row;
            const f32x4 a = *(const f32x4*)(slot + row * 256 + ((cc ^ row) << 4));
            if (m >= M || n >= N) continue;
            const float y0 = a[0] + c.b.x, y1 = a[1] + c.b.y, y2 = a[2] + c.b.z, y3 = a[3] + c.b.w;
            if constexpr (MODE == EPI_F32) {
                *(float4*)((float*)e.out + (size_t)m * e.ldo + n) = make_float4(y0, y1, y2, y3);
            } else if constexpr (MODE == EPI_RESID) {
                // x_out = resid + droppath[b] * gamma * (acc + bias)   (modeling_finetune.py:295-298)
                const int mr = pre.mrow[i];                    // = m, or the residual-stream row of compact row m
                if (mr < 0) continue;
                const size_t o = (size_t)mr * e.ldo + n;
                const float4 r = pre.r[i];
                const float dp = e.rowscale ? e.rowscale[(mr + e.row0) / e.tokens] : 1.0f;
                if (e.out2) {
                    const bf16x4 yv = {f2bf(y0), f2bf(y1), f2bf(y2), f2bf(y3)};
                    *(bf16x4*)((bf16*)e.out2 + o) = yv;
                }
                *(float4*)((float*)e.out + o) = make_float4(r.x + dp * c.g.x * y0, r.y + dp * c.g.y * y1,
                                                            r.z + dp * c.g.z * y2, r.w + dp * c.g.w * y3);
            } else {   // EPI_PATCH
                // row m = b*P + p of the patch GEMM lands in token row b*(P+1) + 1 + p; masked patches take
                // the mask token (modeling_cyclical.py:179-182)
                const int b = m / e.patches, p_ = m - b * e.patches;
                const size_t orow = (size_t)(b * (e.patches + 1) + 1 + p_) * e.ldo + n;
                const bool masked = e.mask && e.mask[m] != 0;
                *(float4*)((float*)e.out + orow) = masked ? c.g : make_float4(y0, y1, y2, y3);
            }
        }
    }
}

// NB blocks of one wave: ACC(b, ct) names the accumulator of block b, column sub-tile ct; MB(b) its first row.
#define EPI_RUN_B(MODE, NB, region, nb_, ACC, MB, WBYTES) do { \
        const EpiCols<MODE> cols_ = epi_cols<MODE>(epi, (nb_), lane, N); \
        constexpr int SLOT_ = stage_f32<MODE>() ? EPI_SLOT_F32 : EPI_SLOT_BF16; \
        constexpr int GRP_ = (WBYTES) / SLOT_ < (NB) ? (WBYTES) / SLOT_ : (NB); \
        _Pragma("unroll") for (int b0_ = 0; b0_ < (NB); b0_ += GRP_) { \
            EpiPre<MODE> pre_[GRP_]; \
            _Pragma("unroll") for (int b_ = 0; b_ < GRP_; ++b_) if (b0_ + b_ < (NB)) pre_[b_] = epi_prefetch<MODE>(epi, lane, MB(b0_ + b_), (nb_), M, N); \
            _Pragma("unroll") for (int b_ = 0; b_ < GRP_; ++b_) if (b0_ + b_ < (NB)) \
                epi_stage<MODE>(cols_, (region) + b_ * SLOT_, lane, ACC(b0_ + b_, 0), ACC(b0_ + b_, 1), ACC(b0_ + b_, 2), ACC(b0_ + b_, 3)); \
            epi_sync(); \
            _Pragma("unroll") for (int b_ = 0; b_ < GRP_; ++b_) if (b0_ + b_ < (NB)) \
                epi_flush<MODE>(epi, cols_, (region) + b_ * SLOT_, lane, MB(b0_ + b_), (nb_), M, N, pre_[b_]); \
            epi_sync(); \
        } } while (0)
#define EPI_RUN(MODE, NB, region, nb_, ACC, MB) EPI_RUN_B(MODE, NB, region, nb_, ACC, MB, EPI_WAVE_BYTES)

// ------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(GEMM_THREADS, 2)
void gemm_nt_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K,
                    int lda, int ldw, GemmEpi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][A 16K | W 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    // tile order: XCD-contiguous chunks (T1); inside them N-groups of <= 8 column tiles walked for every row tile, so a
    // group's W panels stay in the XCD's L2 beside the streaming A panels (wide N would otherwise re-fetch W per row)
    int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int gw = tiles_n <= 8 ? tiles_n : (tiles_n + ((tiles_n + 7) / 8) - 1) / ((tiles_n + 7) / 8);
    int tn0 = 0;
    while (bid >= tiles_m * gw) { bid -= tiles_m * gw; tn0 += gw; gw = min(gw, tiles_n - tn0); }
    const int tm = bid / gw, tn = tn0 + (bid - tm * gw);
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, li = lane & 15;

    // staging: wave-instruction ii (0..15) fills rows ii*8..ii*8+7; lane -> (row, phys chunk)
    const int srow = lane >> 3, pchunk = lane & 7;
    const int schunk = pchunk ^ srow;                 // source chunk (swizzle on the SOURCE side)
    const bf16* a_src[4];
    const bf16* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 8 + srow;
        int ar = m0 + row; ar = ar < M ? ar : M - 1;
        int br = n0 + row; br = br < N ? br : N - 1;
        a_src[i] = A + (size_t)ar * lda + schunk * 8;
        w_src[i] = W + (size_t)br * ldw + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * (2 * STAGE_BYTES);
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ii = i * 4 + wave;
            glds16(a_src[i] + k0, base + ii * 1024);
            glds16(w_src[i] + k0, base + STAGE_BYTES + ii * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * (2 * STAGE_BYTES);
        const char* sw = sa + STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], wf[4];
            const int chunk = kk * 4 + g;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ra = wr * 64 + t * 16 + li;
                af[t] = *(const bf16x8*)(sa + ra * 128 + ((chunk ^ (ra & 7)) << 4));
                const int rw = wc * 64 + t * 16 + li;
                wf[t] = *(const bf16x8*)(sw + rw * 128 + ((chunk ^ (rw & 7)) << 4));
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    }
    // D layout (swapped operands): lane col (li) = m_local, rows 4g+r = n_local
    __syncthreads();                                   // every wave is done reading the operand stages
#define ACC0(b, ct) acc[b][ct]
#define MB0(b) (m0 + wr * 64 + (b) * 16)
    EPI_RUN(MODE, 4, smem + wave * EPI_WAVE_BYTES, n0 + wc * 64, ACC0, MB0);
#undef ACC0
#undef MB0
}

// ------------------------------------------------------------------------------------------
// NT kernel, large shapes: 256x256 block tile, BK = 64, 8 waves (2 M x 4 N), one block per CU.
//
// LDS = 2 K-tile buffers x 4 half-tiles {AL, AH, BL, BH} of [128 rows][64 k] bf16 (16 KiB each,
// 128 KiB total).  A wave owns output rows {wm*64.. in AL} u {wm*64.. in AH} and columns
// {wn*32.. in BL} u {wn*32.. in BH}, so each K-tile is consumed in 4 phases of 16 MFMAs
// (one 64x32 quadrant each) that read
//        p0: AL + BL      p1: BH      p2: AH      p3: -  (BL fragments stay in registers)
// A half-tile slot is read in exactly one phase per K-tile, so the loader can refill it soon after with
// the half-tile of a later K-tile: prefetch depth comes from the consumption order, not from more LDS.
// Never vmcnt(0) in the steady state.  The schedule itself is documented at the kernel below.
// ------------------------------------------------------------------------------------------
#define T_BM 256
#define T_BN 256
#define T_THREADS 512
#define HALF_BYTES (128 * BK * 2)          // 16 KiB
#define T_LDS_BYTES (8 * HALF_BYTES)       // 128 KiB
#define T5_LDS_BYTES (2 * (2 * 24 * 1024 + 2 * HALF_BYTES))   // 160 KiB: the 320-row tile (24 KiB A half-tile slots)

#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LDS_WAIT() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RAW_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// ------------------------------------------------------------------------------------------
// The two wave groups wm = 0 / wm = 1 (one wave of each per SIMD) run half a phase apart -- wm = 1
// takes one extra barrier up front -- so while one group issues its 16 MFMAs the other does its
// ds_reads, LDS-DMA issue and counted wait, and the MFMA pipe of every SIMD stays fed.
// The stagger moves the hazards by one barrier, so the half-tile schedule differs from the lockstep kernel:
//        phase:   p0            p1          p2          p3
//        reads:   AL(t) BL(t)   BH(t)       AH(t)       -
//        issues:  BH(t+1)       AH(t+1)     AL(t+2)     BL(t+2)
//        waits:   BH(t)         AH(t)       -           AL(t+1) BL(t+1)
// RAW: a half-tile is read one phase after the phase whose counted vmcnt (before that phase's first
// barrier) retired it -- the trailing group's wait is then also behind a barrier the reader has passed.
// WAR: every slot is refilled >= 2 phases after its last read (the trailing group's reads of phase p
// complete before barrier 2p+2; the leading group issues phase p+2's DMA after barrier 2p+3).
// 4 half-tiles (64 KiB) are in flight behind every wait: vmcnt(8), exact smaller counts in the K tail.
// ------------------------------------------------------------------------------------------
// MT = 16-row tiles per wave per A half: 4 -> 256-row tile, 5 -> 320-row tile (N = 768 on 256 CUs: 79 x 3 = 237 tiles
// are ONE round where 256-row tiles need 297 = 1.16 rounds).  For MT = 5 an A half-tile holds 160 rows in a 192-row
// (24 KiB) slot so that every wave still issues whole LDS-DMA instructions (3 per half-tile; the last 32 rows are
// padding): LDS = 2 x (24 + 24 + 16 + 16) KiB = 160 KiB, and the counted waits become LA = 3, LB = 2 loads per thread.
// Diagnostic build only (-DGEMM_STAMP, tools/stamp_gemm.py): per-workgroup s_memtime stamps + the hardware id of the CU, to
// lay the tiles of one launch out on a per-CU time line.  Never defined for libuvit.so.
#if defined(GEMM_STAMP) || defined(GEMM_DEBUG)
int uvit_gemm_nt_launch(int, const void*, const void*, int, int, int, int, int, const GemmEpi*, hipStream_t, const GemmTune*, int*);
extern "C" int uvit_debug_gemm_nt(int mode, int variant, const void* A, const void* W, int M, int N, int K, void* out, void* out2,
                                  const float* bias, const float* resid, const float* gamma, void* stream) {
    GemmEpi e; e.out = out; e.out2 = out2; e.bias = bias; e.resid = resid; e.gamma = gamma; e.ldo = N; e.tokens = 197;
    GemmTune t; t.nt_variant = variant % 100; t.nt_persist = variant < 100;
    return uvit_gemm_nt_launch(mode, A, W, M, N, K, K, K, &e, (hipStream_t)stream, &t, nullptr);
}
#endif
#ifdef GEMM_STAMP
__device__ unsigned long long g_gemm_stamps[4096 * 8];
#define GSTAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_gemm_stamps[blockIdx.x * 8 + (i)] = t_; } while (0)
// persistent workgroups: stamps of tile `seq` (0..7) of workgroup blockIdx.x < 256 at [2048 + blockIdx.x * 8 + seq] * 8 + i
#define GSTAMP_P(seq, i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (threadIdx.x == 0 && blockIdx.x < 256 && (seq) < 8) g_gemm_stamps[(2048 + blockIdx.x * 8 + (seq)) * 8 + (i)] = t_; } while (0)
#define GSTAMP_ID() do { if (threadIdx.x == 0 && blockIdx.x < 4096) { \
        g_gemm_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); \
        g_gemm_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); } } while (0)
extern "C" int uvit_debug_gemm_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), sizeof(g_gemm_stamps)) == hipSuccess ? 0 : -3;
}
// one K-tile of the workgroup's first tile, phase by phase: s_memtime WITHOUT a wait (the value lands in the SGPR pair behind
// the kernel's own lgkmcnt waits; it is only read after the K loop), so the phases are not perturbed by a drain
#define PSTAMP_DECL unsigned long long ps_[26]; for (int k_ = 0; k_ < 26; ++k_) ps_[k_] = 0
#define PSTAMP(k) do { if (t == 5 && first) asm volatile("s_memtime %0" : "=s"(ps_[k]) :: "memory"); } while (0)
#define PSTAMP_FLUSH() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if (lane == 0 && (wave & 3) == 0 && blockIdx.x < 4) for (int k_ = 0; k_ < 26; ++k_) \
            g_gemm_stamps[3500 * 8 + ((blockIdx.x * 2 + (wave >> 2)) * 32 + k_)] = ps_[k_]; } while (0)
#else
#define GSTAMP(i)
#define GSTAMP_P(seq, i)
#define GSTAMP_ID()
#define PSTAMP_DECL
#define PSTAMP(k)
#define PSTAMP_FLUSH()
#endif

// PERSIST (MT = 4 only): one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ... and the operand pipeline runs ACROSS tiles:
// the loads the schedule would issue for K-tiles nk, nk + 1 of a tile fetch K-tiles 0, 1 of the workgroup's NEXT tile, so when the
// K loop ends the next tile's prologue is already in LDS.  The epilogue stages through its own 32 KiB (4 KiB per wave) behind
// the operand buffers, the next K loop starts while its stores are still in flight, and no workgroup is retired between tiles.
// What that removes per tile (tools/stamp_gemm.py, fc1 shape, K = 768: K loop 35.0k cycles, epilogue 5.3k (bf16) / 16.5k
// (GELU + GELU')): the prologue wait for the first operands (3.7k / 4.9k) and the gap between a workgroup's last store and the
// next workgroup's first instruction (3.9k / 10.5k: store drain + dispatch).
// vmcnt is one in-order counter for loads and stores: every load the next tile's first K-tile needs is OLDER than the epilogue's
// stores and is waited for (vmcnt(0)) before the first store is issued, so K-tile 0 of a later tile needs no counted wait at all;
// from K-tile 1 on the counted waits are the steady-state ones (a wait then also covers the stores issued before its loads).
#define TP_EPI_BYTES 4096                              // persistent kernel: epilogue staging per wave
#define TP_LDS_BYTES (T_LDS_BYTES + 8 * TP_EPI_BYTES)  // 160 KiB

template <int MODE, int MT, bool PERSIST>
__global__ __launch_bounds__(T_THREADS, 2)
void gemm_nt256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K,
                        int lda, int ldw, GemmEpi epi) {
    static_assert(!PERSIST || MT == 4, "the 320-row tile leaves no LDS for a separate epilogue region");
    constexpr int BM_ = 64 * MT;                       // tile rows
    constexpr int AH_ROWS = 32 * MT;                   // rows of one A half-tile
    constexpr int LA = MT == 4 ? 2 : 3;                // LDS-DMA instructions per wave per A half-tile
    constexpr int A_HALF = LA * 8 * 1024;              // its LDS slot
    constexpr int BUF = 2 * A_HALF + 2 * HALF_BYTES;   // one K-tile buffer: [AL | AH | BL | BH]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GSTAMP(0); GSTAMP_ID();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N / T_BN, tiles_m = (M + BM_ - 1) / BM_, ntiles = tiles_m * tiles_n;
    const int wm = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;
    // tile order: XCD-contiguous chunks (T1) of the VIRTUAL block id (a persistent workgroup's ids blockIdx.x + k gridDim.x stay on
    // its XCD because the grid is a multiple of 8), column tiles grouped per row tile
    auto tile_origin = [&](int vb, int& m0_, int& n0_) {
        int bid = xcd_remap(vb, ntiles);
        const int gmax = epi.ngroup > 0 ? epi.ngroup : 6;   // column tiles walked per row tile before moving to the next row tile
        int gw = tiles_n <= gmax ? tiles_n : (tiles_n + ((tiles_n + gmax - 1) / gmax) - 1) / ((tiles_n + gmax - 1) / gmax);
        int tn0 = 0;
        while (bid >= tiles_m * gw) { bid -= tiles_m * gw; tn0 += gw; gw = min(gw, tiles_n - tn0); }
        const int tm = bid / gw, tn = tn0 + (bid - tm * gw);
        m0_ = tm * BM_; n0_ = tn * T_BN;
    };
    // Operand addressing: every LDS-DMA piece of a tile is ONE per-lane base offset (A: row m0 + 8 wave + srow, B: output
    // column n0 + 64 (wave / 4) + 8 (wave % 4) + srow) plus a wave-uniform row step times the leading dimension, so the loader
    // holds 2 VGPRs per tile and a piece costs one v_add in the K loop.  (Round 2 measured what the alternatives cost: ten
    // offset VGPRs spilled inside the 320-row kernel's K loop, and recomputing row * lda per piece put ~110 issue cycles on
    // every piece of its load blocks, profiles/round2_gemm_phase_timeline_320.txt.)  Only the ragged last row tile clamps rows
    // per lane.
    //   A piece j of half `kind`: tile row kind * AH_ROWS + 64 j + 8 wave + srow; in the 320-row tile the third piece of waves
    //   4..7 is slot padding (rows 160..191 of a 160-row half) and re-reads rows 128..159.
    //   B piece j of half `kind`: W row n0 + cb, cb = 64 (r / 32) + r % 32 (+ 32 for BH), r = 64 j + 8 wave + srow -- a wave's
    //   2 x 32 output columns are adjacent, which gives full 128-B lines per row in the staged epilogue.
    // The 256-row tile has the registers to keep one offset per piece (no in-loop address arithmetic at all: measured 2-3 %
    // faster there than base + step); the 320-row tile, at the 256-VGPR limit, uses base + step.
    struct TileSrc { uint32_t a, b; int m0; bool edge; uint32_t pa[2][LA], pb[2][2]; };
    auto tile_src = [&](int m0_, int n0_, TileSrc& ts) {
        ts.a = (uint32_t)(m0_ + wave * 8 + srow) * (uint32_t)lda + schunk * 8;
        ts.b = (uint32_t)(n0_ + (wave >> 2) * 64 + (wave & 3) * 8 + srow) * (uint32_t)ldw + schunk * 8;
        ts.m0 = m0_;
        ts.edge = m0_ + BM_ > M;
        if constexpr (MT != 5) {
#pragma unroll
            for (int kind = 0; kind < 2; ++kind)
#pragma unroll
                for (int j = 0; j < LA; ++j) {
                    int r = m0_ + kind * AH_ROWS + j * 64 + wave * 8 + srow; r = r < M ? r : M - 1;
                    ts.pa[kind][j] = (uint32_t)r * (uint32_t)lda + schunk * 8;
                    if (j < 2) ts.pb[kind][j] = ts.b + (uint32_t)(j * 128 + kind * 32) * (uint32_t)ldw;
                }
        }
    };
    // Tile assignment of the persistent form.  Fixed stride (tile_counter == nullptr): tiles blockIdx.x, + gridDim.x, ...  Dynamic
    // (round 4): XCD x = blockIdx.x % 8 owns the virtual ids x, x + 8, x + 16, ... and every workgroup takes the next unclaimed one of
    // ITS XCD from a per-XCD counter, one tile ahead of the one it computes (the operand pipeline prefetches across tiles).  With all
    // CUs present nothing changes; with CUs held by somebody else (RCCL's channel workgroups in a data-parallel run) the workgroups
    // that start late find few tiles left instead of a full fixed share, and the launch takes tiles / CUs longer, not twice as long.
    // (the ticket crosses the workgroup through the first word of the epilogue staging area, which is idle between two epilogues: the
    //  kernel's 160 KiB of dynamic LDS are the whole LDS of the CU, there is no room for a static word)
    volatile int* const s_ticket = (volatile int*)(smem + (PERSIST ? T_LDS_BYTES : 0));
    unsigned* const tcnt = PERSIST ? epi.tile_counter : nullptr;
    const int xcd = blockIdx.x & 7;
    auto claim = [&]() -> int {                        // thread 0 only: the next virtual tile id of this XCD (may be >= ntiles)
        return xcd + 8 * (int)__hip_atomic_fetch_add(tcnt + xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    int vb = blockIdx.x, vb_next = blockIdx.x + (int)gridDim.x;
    if constexpr (PERSIST) {
        if (tcnt) {
            if (tid == 0) { *s_ticket = claim(); }
            __syncthreads();
            vb = *s_ticket;
            __syncthreads();
            if (tid == 0) { *s_ticket = claim(); }
            __syncthreads();
            vb_next = *s_ticket;
            __syncthreads();
            if (vb >= ntiles) {                        // nothing left for this workgroup (it started late): leave through the exit counter
                if (tid == 0) {
                    const unsigned d = __hip_atomic_fetch_add(tcnt + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d == gridDim.x - 1) for (int i = 0; i < 9; ++i) __hip_atomic_store(tcnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
        }
    }
    int m0, n0;
    tile_origin(vb, m0, n0);
    TileSrc cur, nxt;
    tile_src(m0, n0, cur);
    nxt = cur;
    int pending = -1;                                  // thread 0: the claim issued at the top of the previous tile (for the tile after next)
    const int nk = K / BK;
    int par = 0;                                       // LDS buffer of the current tile's K-tile 0 (alternates when nk is odd)
    bool more = false;                                 // this workgroup has another tile after the current one
    auto issue = [&](int kind, int t) {                // kind: 0 AL, 1 AH, 2 BL, 3 BH; t >= nk: K-tile t - nk of the next tile
        const bool own = t < nk;
        if (own || more) {
            char* buf = smem + ((t + par) & 1) * BUF;
            const uint32_t k0 = (uint32_t)(own ? t : t - nk) * BK;
            const TileSrc& T = own ? cur : nxt;
            if (kind < 2) {
                char* dst = buf + kind * A_HALF + wave * 1024;
#pragma unroll
                for (int j = 0; j < LA; ++j) {
                    int roff = kind * AH_ROWS + j * 64;
                    if (MT == 5 && j == 2 && wave >= 4) roff -= 32;
                    uint32_t off;
                    if constexpr (MT != 5) off = T.pa[kind][j];
                    else if (!T.edge) off = T.a + (uint32_t)roff * (uint32_t)lda;
                    else { int r = T.m0 + roff + wave * 8 + srow; r = r < M ? r : M - 1; off = (uint32_t)r * (uint32_t)lda + schunk * 8; }
                    glds16(A + (off + k0), dst + j * 8 * 1024);
                }
            } else {
                char* dst = buf + 2 * A_HALF + (kind - 2) * HALF_BYTES + wave * 1024;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint32_t off;
                    if constexpr (MT != 5) off = T.pb[kind - 2][j];
                    else off = T.b + (uint32_t)(j * 128 + (kind - 2) * 32) * (uint32_t)ldw;
                    glds16(W + (off + k0), dst + j * 8 * 1024);
                }
            }
        }
    };
    const int sw0 = ((g) ^ (li & 7)) << 4, sw1 = ((4 + g) ^ (li & 7)) << 4;
    const int a_off = (wm * 16 * MT + li) * 128, b_off = (wn * 32 + li) * 128;

    f32x4 acc[2][2][MT][2];
    bf16x8 af[MT][2], b0f[2][2], b1f[2][2];

#define LOAD_A(half_base) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { \
        af[mt][0] = *(const bf16x8*)((half_base) + a_off + mt * 2048 + sw0); \
        af[mt][1] = *(const bf16x8*)((half_base) + a_off + mt * 2048 + sw1); }
#define LOAD_B(dst, half_base) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) { \
        dst[nt][0] = *(const bf16x8*)((half_base) + b_off + nt * 2048 + sw0); \
        dst[nt][1] = *(const bf16x8*)((half_base) + b_off + nt * 2048 + sw1); }
#define MMA(mq, nq, bfr) do { __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) \
            acc[mq][nq][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt][kk], af[mt][kk], acc[mq][nq][mt][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0); } while (0)
    // counted waits: loads per thread younger than the half-tile waited for (LA per A half-tile, 2 per B half-tile)
#define WAIT_FULL() do { if constexpr (MT == 4) VM_WAIT(8); else VM_WAIT(10); } while (0)      /* 2 LA + 2 LB */
#define WAIT_AB() do { if constexpr (MT == 4) VM_WAIT(4); else VM_WAIT(5); } while (0)         /* LA + LB */
#define WAIT_A() do { if constexpr (MT == 4) VM_WAIT(2); else VM_WAIT(3); } while (0)          /* LA */

    // ---- prologue of the workgroup's first tile: AL0 BL0 BH0 AH0 AL1 BL1 (BH1, AH1 are issued in phases 0, 1 of K-tile 0)
    issue(0, 0); issue(2, 0); issue(3, 0); issue(1, 0);
    issue(0, 1); issue(2, 1);
    if (nk >= 2) WAIT_FULL(); else WAIT_AB();          // AL0, BL0 landed (younger: BH0 AH0 [AL1 BL1])
    bool first = true;                                 // later tiles: everything the prologue loads has already landed
    PSTAMP_DECL;
#ifdef GEMM_STAMP
    int seq = 0;
#endif

    for (;;) {
        if constexpr (PERSIST) {
            more = vb_next < ntiles;
            if (more) { int m1, n1; tile_origin(vb_next, m1, n1); tile_src(m1, n1, nxt); }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < MT; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        RAW_BARRIER();
        GSTAMP(1); GSTAMP_P(seq, 1);
        if (wm == 1) RAW_BARRIER();                    // the stagger: wm = 1 runs one barrier behind

        for (int t = 0; t < nk; ++t) {
            const char* buf = smem + ((t + par) & 1) * BUF;
            const bool has1 = t + 1 < nk || more, has2 = t + 2 < nk || more;
            const bool landed = !first && t == 0;      // K-tile 0 (and AL1, BL1) of a later tile: waited for before the epilogue's stores
            // ---- phase 0: quadrant (0,0) <- AL, BL
            PSTAMP(0);
            LOAD_B(b0f, buf + 2 * A_HALF);
            LOAD_A(buf);
            issue(3, t + 1);                           // BH(t+1): slot last read in phase 1 of tile t-1
            PSTAMP(1);
            if (!landed) { if (has1) WAIT_FULL(); else WAIT_A(); }      // BH(t) landed (younger: AH(t) [AL BL BH](t+1))
            PSTAMP(2);
            RAW_BARRIER();
            PSTAMP(3);
            LDS_WAIT();
            PSTAMP(4);
            MMA(0, 0, b0f);
            PSTAMP(5);
            RAW_BARRIER();
            // ---- phase 1: quadrant (0,1) <- BH
            PSTAMP(6);
            LOAD_B(b1f, buf + 2 * A_HALF + HALF_BYTES);
            issue(1, t + 1);                           // AH(t+1): slot last read in phase 2 of tile t-1
            PSTAMP(7);
            if (!landed) { if (has1) WAIT_FULL(); else VM_WAIT(0); }    // AH(t) landed
            PSTAMP(8);
            RAW_BARRIER();
            PSTAMP(9);
            LDS_WAIT();
            PSTAMP(10);
            MMA(0, 1, b1f);
            PSTAMP(11);
            RAW_BARRIER();
            // ---- phase 2: quadrant (1,1) <- AH
            PSTAMP(12);
            LOAD_A(buf + A_HALF);
            issue(0, t + 2);                           // AL(t+2): slot last read in phase 0
            PSTAMP(13);
            RAW_BARRIER();
            PSTAMP(15);
            LDS_WAIT();
            PSTAMP(16);
            MMA(1, 1, b1f);
            PSTAMP(17);
            RAW_BARRIER();
            // ---- phase 3: quadrant (1,0), no LDS read
            PSTAMP(18);
            issue(2, t + 2);                           // BL(t+2): slot last read in phase 0
            PSTAMP(19);
            if (!landed) { if (has2) WAIT_FULL(); else if (has1) WAIT_AB(); }   // AL(t+1), BL(t+1) landed (younger: BH AH (t+1) [AL BL (t+2)])
            PSTAMP(20);
            RAW_BARRIER();
            PSTAMP(21);
            MMA(1, 0, b0f);
            PSTAMP(23);
            RAW_BARRIER();
            PSTAMP(24);
        }
        if (wm == 0) RAW_BARRIER();                    // matches the trailing group's last barrier
        if (first) PSTAMP_FLUSH();
        GSTAMP(2); GSTAMP_P(seq, 2);

#define ACC1(b, ct) acc[(b) / MT][(ct) >> 1][(b) % MT][(ct) & 1]
#define MB1(b) (m0 + ((b) / MT) * AH_ROWS + wm * 16 * MT + ((b) % MT) * 16)
        // the trailing wave group loses the VALU arbitration in the epilogue (round-2 stamps: it finishes 2-11k cycles after the leading
        // one, which then idles at the next tile's first barrier): it runs the epilogue at priority 1 (round 3 A/B, fc1 GELU+GELU'
        // 158.1 -> 153.4 us, fc1 GELU 144.8 -> 140.9 us, qkv 92.8 -> 91.4 us; priority 3 is no better).  Inside the two-stream step the gain
        // is within noise (same box, alternating: 25.31 / 25.36 / 25.41 ms with, 25.35 / 25.49 / 25.39 ms without).
        // K-loop variants measured beside it and dropped: static priority for this group without the per-block flips (+15-20 %),
        // flips with either group one level above the other (+10-15 %); profiles/round3_gemm_epilogue_priority_ab.txt
        if (wm == 1) __builtin_amdgcn_s_setprio(1);
        if constexpr (PERSIST) {
            // the next tile's K-tile 0 + AL1 BL1 were issued during the last two K-tiles: land them BEFORE the first store
            VM_WAIT(0);
            // the claim for the tile after next: issued HERE, behind the drain and in front of the epilogue's stores, and read at the top
            // of the next tile.  (Issued at the top of a tile it sat in wave 0's in-order vmcnt queue in front of the K loop's counted
            // waits, which then waited for an L2 atomic's round trip: +0.8 % on the step.)
            if (tcnt && tid == 0) pending = more ? claim() : ntiles;
            GSTAMP_P(seq, 4);
            EPI_RUN_B(MODE, 2 * MT, smem + T_LDS_BYTES + wave * TP_EPI_BYTES, n0 + wn * 64, ACC1, MB1, TP_EPI_BYTES);
        } else {
            EPI_RUN(MODE, 2 * MT, smem + wave * EPI_WAVE_BYTES, n0 + wn * 64, ACC1, MB1);
        }
        __builtin_amdgcn_s_setprio(0);
#undef ACC1
#undef MB1
        GSTAMP(3); GSTAMP_P(seq, 3);
        if (!PERSIST || !more) break;
        if (tcnt) {
            if (tid == 0) *s_ticket = pending;
            __syncthreads();
            vb = vb_next; vb_next = *s_ticket;
            __syncthreads();
        } else {
            vb = vb_next; vb_next += (int)gridDim.x;
        }
        tile_origin(vb, m0, n0);
        cur = nxt;
        par ^= nk & 1;
        first = false;
#ifdef GEMM_STAMP
        ++seq;
#endif
    }
    if constexpr (PERSIST) {
        if (tcnt && tid == 0) {                        // last workgroup out puts the counters back to zero for the next launch on this stream
            const unsigned d = __hip_atomic_fetch_add(tcnt + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == gridDim.x - 1) for (int i = 0; i < 9; ++i) __hip_atomic_store(tcnt + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef LOAD_A
#undef LOAD_B
#undef MMA
#undef WAIT_FULL
#undef WAIT_AB
#undef WAIT_A
}

// ------------------------------------------------------------------------------------------
// NT kernel "R" (ring): (16 MT) x 256 block tile, BK = 32, 4 waves side by side over the 256 columns (each wave: all
// rows x 64 columns), TWO workgroups per CU.
//
// Why it exists: with K = 768 the launches of the step are as much epilogue as main loop (fc1: 310 MB of stores against
// 61 us of MFMA work), and a one-workgroup-per-CU kernel runs the two back to back -- the MFMA pipe idles while its tile
// drains at the HBM rate.  Two INDEPENDENT workgroups per CU drift apart by themselves: one streams its finished tile out
// while the other is in its K loop, with no in-order vmcnt coupling between one's stores and the other's operand loads
// (CDNA4 counts loads and stores in one queue per wave).  The 128x128 kernel above has the same residency but a
// one-K-tile prefetch distance and twice the L2 traffic per flop; here the ring holds THREE K-steps (72-80 KB per
// workgroup: two 32-deep steps in flight behind every wait, never vmcnt(0) in the steady state) and the tile is twice as wide.
// LDS images are [rows][32 k] (64-B rows, 1 KB = 16 rows per LDS-DMA wave-instruction); 16-B chunk c of row r sits at
// c ^ F[(r >> 2) & 3], F = {0,2,3,1}: the four 16-lane groups of a ds_read_b128 then each cover all 16 slots of the 256-B
// bank row (a plain or (r >> 2)-XOR image is 2-way conflicted).  The swizzle is applied to the SOURCE address.
// ------------------------------------------------------------------------------------------
#define R_BN 256
#define R_BK 32
#define R_THREADS 256
#define R_STAGES 3
__host__ __device__ constexpr int r_stage_bytes(int mt) { return (16 * mt + R_BN) * R_BK * 2; }
__host__ __device__ constexpr int r_lds_bytes(int mt) {
    return R_STAGES * r_stage_bytes(mt) > 4 * EPI_WAVE_BYTES ? R_STAGES * r_stage_bytes(mt) : 4 * EPI_WAVE_BYTES;
}

template <int MODE, int MT>
__global__ __launch_bounds__(R_THREADS, 2)
void gemm_ntr_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, int M, int N, int K,
                     int lda, int ldw, GemmEpi epi) {
    constexpr int BM_ = 16 * MT;
    constexpr int STAGE = r_stage_bytes(MT);
    constexpr int LA_MAX = (MT + 3) / 4;                 // LDS-DMA instructions per wave for the A image (16 rows each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = N / R_BN, tiles_m = (M + BM_ - 1) / BM_;
    int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    int gw = tiles_n <= 6 ? tiles_n : (tiles_n + ((tiles_n + 5) / 6) - 1) / ((tiles_n + 5) / 6);
    int tn0 = 0;
    while (bid >= tiles_m * gw) { bid -= tiles_m * gw; tn0 += gw; gw = min(gw, tiles_n - tn0); }
    const int tm = bid / gw, tn = tn0 + (bid - tm * gw);
    const int m0 = tm * BM_, n0 = tn * R_BN;
    const int g = lane >> 4, li = lane & 15;
    // F = {0, 2, 3, 1} as a 2-bit lookup
    auto fswz = [](int q) { return (0x78 >> (2 * q)) & 3; };          // 0b01'11'10'00

    // ---- loader: instruction covers 16 rows; lane -> (row = lane >> 2, physical chunk = lane & 3)
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ fswz((lane >> 4) & 3);            // logical (source) chunk of this lane
    uint32_t srcA[LA_MAX], srcB[4];
    const int la = (MT - wave + 3) / 4;                                // A instructions of this wave: row groups wave, wave+4, ...
#pragma unroll
    for (int j = 0; j < LA_MAX; ++j) {
        int r = m0 + (j * 4 + wave) * 16 + lrow; r = r < M ? r : M - 1;
        srcA[j] = (uint32_t)r * (uint32_t)lda + lchunk * 8;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = n0 + (j * 4 + wave) * 16 + lrow;                 // N % 256 == 0: always in range
        srcB[j] = (uint32_t)r * (uint32_t)ldw + lchunk * 8;
    }
    const int nk = K / R_BK;
    auto issue = [&](int t) {
        char* st = smem + (t % R_STAGES) * STAGE;
        const uint32_t k0 = (uint32_t)t * R_BK;
#pragma unroll
        for (int j = 0; j < LA_MAX; ++j)
            if (j < la) glds16(A + (srcA[j] + k0), st + (j * 4 + wave) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16(W + (srcB[j] + k0), st + BM_ * 64 + (j * 4 + wave) * 1024);
    };
    // ---- fragment addresses: lane (g, li) reads row base + li, logical chunk g
    const int foff = li * 64 + ((g ^ fswz((li >> 2) & 3)) << 4);
    const int boff = BM_ * 64 + wave * 64 * 64 + foff;                  // this wave's 64 W rows (output columns)

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // loads per stage of this wave: la + 4.  The counted wait before K-step t leaves the next stage's loads in flight.
    issue(0);
    if (nk > 1) issue(1);
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) { if (la == LA_MAX) { if constexpr (LA_MAX == 2) VM_WAIT(6); else VM_WAIT(7); }
                          else { if constexpr (LA_MAX == 2) VM_WAIT(5); else VM_WAIT(6); } }
        else VM_WAIT(0);
        RAW_BARRIER();                                     // stage t landed for every wave; stage t-1 is no longer read
        if (t + 2 < nk) issue(t + 2);                      // refills the slot of stage t-1
        const char* st = smem + (t % R_STAGES) * STAGE;
        bf16x8 wf[4], af[MT];
        constexpr int H1 = MT / 2;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(st + boff + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < H1; ++mt) af[mt] = *(const bf16x8*)(st + foff + mt * 1024);
        LDS_WAIT();
        // the second half of the A fragments is read under the first half's MFMAs
#pragma unroll
        for (int mt = H1; mt < MT; ++mt) af[mt] = *(const bf16x8*)(st + foff + mt * 1024);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < H1; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        LDS_WAIT();
#pragma unroll
        for (int mt = H1; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();                                       // every wave is done reading the ring: reuse it for staging
#define ACCR(b, ct) acc[b][ct]
#define MBR(b) (m0 + (b) * 16)
    EPI_RUN(MODE, MT, smem + wave * EPI_WAVE_BYTES, n0 + wave * 64, ACCR, MBR);
#undef ACCR
#undef MBR
}

// ------------------------------------------------------------------------------------------
// TN kernel:  C[Nn,Kk] = Y[M,Nn]^T . X[M,Kk];  M (reduction) must be a multiple of 64 and the
// buffers readable (zero rows) up to it.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int col0, int kk, int lane) {
    // operand element j of lane (g, i): tile[row = kk*32 + 8g + j][col0 + i]
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int r0 = kk * 32 + 8 * g + q, r1 = r0 + 4;
    const int chunk = (col0 >> 3) + (p >> 1);
    const int a0 = r0 * 256 + ((chunk ^ tn_swz(r0)) << 4) + ((p & 1) << 3);
    const int a1 = r1 * 256 + ((chunk ^ tn_swz(r1)) << 4) + ((p & 1) << 3);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + a1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

#define TN_OUT_LD 132                                   // padded fp32 row of the staged output tile
#define TN_LDS_BYTES (BM * TN_OUT_LD * 4)               // 67584 B >= 4 * STAGE_BYTES

// grid = (output tiles, split): split s reduces token rows [s*steps_per_split*64, ...) and the
// partial tiles are combined with fp32 atomics (the gradient arena is zeroed once per step).
// The accumulators are re-laid out through LDS so every atomic wave-instruction covers 256
// contiguous bytes of one output row (the full-rate shape for global float atomics).
__global__ __launch_bounds__(GEMM_THREADS, 2)
void gemm_tn_kernel(const bf16* __restrict__ Y, const bf16* __restrict__ X, int M, int Nn, int Kk,
                    int ldy, int ldx, float* __restrict__ C, int ldc, int steps_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][Y tile 16K | X tile 16K], reused for the output
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_k = (Kk + BN - 1) / BN, tiles_n = (Nn + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, tiles_n * tiles_k);
    const int tn = bid / tiles_k, tk = bid - tn * tiles_k;
    const int n0 = tn * BM, k0 = tk * BN;
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int nm_total = M / BK;
    const int m_begin = blockIdx.y * steps_per_split;
    const int nm = min(steps_per_split, nm_total - m_begin);
    if (nm <= 0) return;

    // staging: wave-instruction ii (0..15) fills tile rows ii*4..ii*4+3 (256-B rows)
    const int srow = lane >> 4, pchunk = lane & 15;
    const bf16* y_src[4];
    const bf16* x_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 4 + srow;
        const int sch = pchunk ^ tn_swz(row);
        int yc = n0 / 8 + sch; yc = yc < Nn / 8 ? yc : Nn / 8 - 1;
        int xc = k0 / 8 + sch; xc = xc < Kk / 8 ? xc : Kk / 8 - 1;
        y_src[i] = Y + ((size_t)m_begin * BK + row) * ldy + yc * 8;
        x_src[i] = X + ((size_t)m_begin * BK + row) * ldx + xc * 8;
    }
    auto stage = [&](int buf, int mt) {
        char* base = smem + buf * (2 * STAGE_BYTES);
        const size_t mrow = (size_t)mt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ii = i * 4 + wave;
            glds16(y_src[i] + mrow * ldy, base + ii * 1024);
            glds16(x_src[i] + mrow * ldx, base + STAGE_BYTES + ii * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    for (int mt = 0; mt < nm; ++mt) {
        const int cur = mt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (mt + 1 < nm) stage(cur ^ 1, mt + 1);
        const char* sy = smem + cur * (2 * STAGE_BYTES);
        const char* sx = sy + STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 yf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                yf[t] = tr_frag(sy, wr * 64 + t * 16, kk, lane);
                xf[t] = tr_frag(sx, wc * 64 + t * 16, kk, lane);
            }
            // D[i = k_local][j = n_local] = sum_m X[m][k] Y[m][n]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[kt], yf[nt], acc[nt][kt], 0, 0, 0);
        }
    }
    if (gridDim.y == 1) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wr * 64 + nt * 16 + li;
            if (n >= Nn) continue;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int k = k0 + wc * 64 + kt * 16 + 4 * g;
                if (k < Kk) {
                    const f32x4 a = acc[nt][kt];
                    *(float4*)(C + (size_t)n * ldc + k) = make_float4(a[0], a[1], a[2], a[3]);
                }
            }
        }
        return;
    }
    // split: stage the 128x128 fp32 tile in LDS, then row-contiguous atomics
    __syncthreads();
    float* ot = (float*)smem;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f32x4 a = acc[nt][kt];
            *(float4*)(ot + (wr * 64 + nt * 16 + li) * TN_OUT_LD + wc * 64 + kt * 16 + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
        }
    __syncthreads();
    for (int r = wave; r < BM; r += 4) {
        const int n = n0 + r;
        if (n >= Nn) break;
        float* dst = C + (size_t)n * ldc + k0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = h * 64 + lane;
            if (k0 + k < Kk) atomicAdd(dst + k, ot[r * TN_OUT_LD + k]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// TN kernel, 256x256 output tile: the staggered 8-wave structure of gemm_nt256_kernel with the token
// dimension as the reduction.  A K-tile is 64 tokens; its four half-tiles {YL, YH, XL, XH} are
// [64 tokens][128 columns] bf16 images (256-B rows, 16 KiB, chunk c of row r at c ^ tn_swz(r)) read with
// ds_read_b64_tr_b16, so the phase schedule, the vmcnt counts (2 LDS-DMA per thread per half-tile) and
// the hazard analysis are those of the NT kernel.  Output: C[n][k] (+)= sum_m Y[m][n] X[m][k], the
// accumulators re-laid out through wave-private LDS slots: float4 stores (no split) or one fp32 atomic per
// lane, 256 contiguous bytes per wave-instruction (split reduction).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 tr_pair(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p + 1024));     // token rows + 4
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// one segment of one output tile: K-tiles [t0, t0 + nk) of 64 tokens, rows n0.., columns k0..
__device__ __forceinline__ void tn256_segment(char* smem, const bf16* __restrict__ Y, const bf16* __restrict__ X,
                                              int ldy, int ldx, int n0, int k0, int t0, int nk,
                                              float* __restrict__ C, int ldc, bool atomic, int lane, int wave,
                                              float* __restrict__ bias = nullptr, bool accum = false) {
    // atomic: partial sums of a split reduction, combined with fp32 atomics; otherwise this workgroup owns the tile
    // and stores it (accum: adds it to the value already there).
    // bias != nullptr: also accumulate the column sums of this tile's 256 Y columns (the Linear's bias gradient),
    // bias[i] += sum_m Y[m][n0 + i], as MFMAs against an all-ones operand: wave (wm, wn) sums the 16-column group
    // mt = wn of each Y half beside its phases 1 and 3 (2 extra MFMAs per phase).
    const int wm = wave >> 2, wn = wave & 3;
    // ---- loader: a wave-instruction fills 4 token rows x 256 B; this wave issues instructions {wave, 8 + wave}
    const bf16* src[4][2];                             // [YL, YH, XL, XH][instruction]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 8 + wave) * 4 + (lane >> 4);
        const int sc = ((lane & 15) ^ tn_swz(row)) * 8;            // source column inside the half-tile
        const size_t mrow = (size_t)t0 * BK + row;
        // a wave's 2 x 32 output columns are adjacent: XL column c <-> k0 + 64 (c / 32) + c % 32, XH + 32
        const int xc = (sc >> 5) * 64 + (sc & 31);
        src[0][j] = Y + mrow * ldy + n0 + sc;
        src[1][j] = Y + mrow * ldy + n0 + 128 + sc;
        src[2][j] = X + mrow * ldx + k0 + xc;
        src[3][j] = X + mrow * ldx + k0 + xc + 32;
    }
    const size_t ystep = (size_t)BK * ldy, xstep = (size_t)BK * ldx;
    auto issue = [&](int kind, int t) {
        if (t < nk) {
            char* dst = smem + (t & 1) * (4 * HALF_BYTES) + kind * HALF_BYTES + wave * 1024;
            const size_t step = kind < 2 ? ystep : xstep;
            glds16(src[kind][0] + t * step, dst);
            glds16(src[kind][1] + t * step, dst + 8 * 1024);
        }
    };
    // ---- fragments: element j of lane (g, i) = tile[token 32 kk + 8g + j][col0 + i]
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = 8 * g + q, swz = tn_swz(r0);
    int yoff[4], xoff[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) yoff[mt] = r0 * 256 + (((wm * 8 + mt * 2 + (pp >> 1)) ^ swz) << 4) + ((pp & 1) << 3);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) xoff[nt] = r0 * 256 + (((wn * 4 + nt * 2 + (pp >> 1)) ^ swz) << 4) + ((pp & 1) << 3);

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 yf[4][2], x0f[2][2], x1f[2][2];
    f32x4 bacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bf16 one = (bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
#define BIAS_MMA(mq) do { if (bias) { \
        if (wn == 0) { bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[0][0], bacc[mq], 0, 0, 0); bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[0][1], bacc[mq], 0, 0, 0); } \
        else if (wn == 1) { bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[1][0], bacc[mq], 0, 0, 0); bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[1][1], bacc[mq], 0, 0, 0); } \
        else if (wn == 2) { bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[2][0], bacc[mq], 0, 0, 0); bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[2][1], bacc[mq], 0, 0, 0); } \
        else { bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[3][0], bacc[mq], 0, 0, 0); bacc[mq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf[3][1], bacc[mq], 0, 0, 0); } } } while (0)

#define LOAD_Y(half_base) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) { \
        yf[mt][0] = tr_pair((half_base) + yoff[mt]); yf[mt][1] = tr_pair((half_base) + yoff[mt] + 8192); }
#define LOAD_X(dst, half_base) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) { \
        dst[nt][0] = tr_pair((half_base) + xoff[nt]); dst[nt][1] = tr_pair((half_base) + xoff[nt] + 8192); }
    // D[i = k_local][j = n_local]: lane li <-> output row n, registers <-> 4 consecutive output columns k
#define MMA(mq, nq, xfr) do { __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) \
            acc[mq][nq][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xfr[nt][kk], yf[mt][kk], acc[mq][nq][mt][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0); } while (0)

    RAW_BARRIER();                                     // a previous segment's staging slots are drained
    issue(0, 0); issue(2, 0); issue(3, 0); issue(1, 0);
    issue(0, 1); issue(2, 1);
    if (nk >= 2) VM_WAIT(8); else VM_WAIT(4);          // YL0, XL0 landed
    RAW_BARRIER();
    if (wm == 1) RAW_BARRIER();                        // the stagger

    for (int t = 0; t < nk; ++t) {
        const char* buf = smem + (t & 1) * (4 * HALF_BYTES);
        const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
        // ---- phase 0: quadrant (0,0) <- YL, XL
        LOAD_X(x0f, buf + 2 * HALF_BYTES);
        LOAD_Y(buf);
        issue(3, t + 1);
        if (has1) VM_WAIT(8); else VM_WAIT(2);
        RAW_BARRIER();
        LDS_WAIT();
        MMA(0, 0, x0f);
        RAW_BARRIER();
        // ---- phase 1: quadrant (0,1) <- XH
        LOAD_X(x1f, buf + 3 * HALF_BYTES);
        issue(1, t + 1);
        if (has1) VM_WAIT(8); else VM_WAIT(0);
        RAW_BARRIER();
        LDS_WAIT();
        MMA(0, 1, x1f);
        BIAS_MMA(0);
        RAW_BARRIER();
        // ---- phase 2: quadrant (1,1) <- YH
        LOAD_Y(buf + HALF_BYTES);
        issue(0, t + 2);
        RAW_BARRIER();
        LDS_WAIT();
        MMA(1, 1, x1f);
        RAW_BARRIER();
        // ---- phase 3: quadrant (1,0)
        issue(2, t + 2);
        if (has2) VM_WAIT(8); else if (has1) VM_WAIT(4);
        RAW_BARRIER();
        MMA(1, 0, x0f);
        BIAS_MMA(1);
        RAW_BARRIER();
    }
    if (wm == 0) RAW_BARRIER();
#undef LOAD_Y
#undef LOAD_X
#undef MMA
#undef BIAS_MMA

    // every D row of the ones-product holds the column sums: lane (g = 0, li) register 0 <-> Y column ... + li
    if (bias && lane < 16) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
            float* dst = bias + mq * 128 + wm * 64 + wn * 16 + lane;
            if (atomic) atomicAdd(dst, bacc[mq][0]); else *dst = bacc[mq][0] + (accum ? *dst : 0.f);
        }
    }
    // ---- epilogue: 8 blocks of 16 rows x 64 columns per wave, 4 fp32 slots at a time
    char* region = smem + wave * EPI_WAVE_BYTES;
    const int li = lane & 15;
    const int kb = k0 + wn * 64;
#pragma unroll
    for (int mq = 0; mq < 2; ++mq) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                *(f32x4*)(region + mt * EPI_SLOT_F32 + li * 256 + (((ct * 4 + g) ^ li) << 4)) = acc[mq][ct >> 1][mt][ct & 1];
        epi_sync();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const char* slot = region + mt * EPI_SLOT_F32;
            const int nrow = n0 + mq * 128 + wm * 64 + mt * 16;
            if (atomic) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = *(const float*)(slot + r * 256 + ((((lane >> 2) ^ r)) << 4) + ((lane & 3) << 2));
                    atomicAdd(C + (size_t)(nrow + r) * ldc + kb + lane, v);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * i + (lane >> 4), cc = lane & 15;
                    const f32x4 v = *(const f32x4*)(slot + row * 256 + ((cc ^ row) << 4));
                    float4* dst = (float4*)(C + (size_t)(nrow + row) * ldc + kb + 4 * cc);
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (accum) { const float4 c0 = *dst; o.x += c0.x; o.y += c0.y; o.z += c0.z; o.w += c0.w; }
                    *dst = o;
                }
            }
        }
        epi_sync();
    }
}

__global__ __launch_bounds__(T_THREADS, 2)
void gemm_tn256_kernel(const bf16* __restrict__ Y, const bf16* __restrict__ X, int M, int Nn, int Kk,
                       int ldy, int ldx, float* __restrict__ C, int ldc, int steps_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_k = Kk / T_BN, tiles_n = Nn / T_BM;
    const int bid = xcd_remap(blockIdx.x, tiles_n * tiles_k);
    const int tn = bid / tiles_k, tk = bid - tn * tiles_k;
    const int t0 = blockIdx.y * steps_per_split;
    const int nk = min(steps_per_split, M / BK - t0);
    if (nk <= 0) return;
    tn256_segment(smem, Y, X, ldy, ldx, tn * T_BM, tk * T_BN, t0, nk, C, ldc, gridDim.y > 1, lane, wave);
}

// All wgrad GEMMs of one transformer layer in one launch.  An item = (problem, output tile, token chunk); items
// are ordered chunk-major so the workgroups resident at any time reduce the same token range of every problem
// (their Y / X panels are shared through L2 / Infinity Cache), and the four launches' partial last rounds
// collapse into one.  Chunks of one tile are combined with fp32 atomics (the gradient arena is zero).
__global__ __launch_bounds__(T_THREADS, 2)
void gemm_tn256_group_kernel(const TnGroup G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-contiguous item ranges (T1): the ~27 items one XCD's CUs work on at a time are neighbouring tiles of one
    // problem, so the Y panels of a tile row and the X panels of a tile column are fetched into that L2 once
    // (without it the launch moves 3.7x its operand bytes through the fabric and is bandwidth-bound: profiles/)
    int b = xcd_remap(blockIdx.x, gridDim.x), c = 0, p = 0;
    bool found = false;
    for (c = 0; c < G.max_chunks && !found; ++c)
        for (p = 0; p < G.nprob; ++p) {
            if (c >= G.p[p].chunks) continue;
            const int nt = G.p[p].tiles_n * G.p[p].tiles_k;
            if (b < nt) { found = true; break; }
            b -= nt;
        }
    if (!found) return;
    c -= 1;                                            // the for statement stepped once more after the hit
    const TnProb& P = G.p[p];
    // tile order inside a problem: walk the SHORTER tile dimension fastest, so that a contiguous item range (one XCD's share) touches few
    // panels of the longer one -- fc2 (3 x 12 tiles): 27 consecutive items then read 3 Y + 9 X panels instead of 3 + 12
    int tn, tk;
    if (P.tiles_k > P.tiles_n) { tk = b / P.tiles_n; tn = b - tk * P.tiles_n; }
    else { tn = b / P.tiles_k; tk = b - tn * P.tiles_k; }
    const int t0 = c * P.chunk_steps;
    const int nk = min(P.chunk_steps, P.nm - t0);
    if (nk <= 0) return;
    float* bias = nullptr;
    if (tk == 0 && P.bias) {
        const int n0 = tn * T_BM;                      // bias segments are multiples of 256 columns
        // stacked two-stream rows: a chunk lies in one stream (the launcher only plans such chunk lengths) and sums into that stream's bias
        const bool s1 = P.s1_row > 0 && t0 * BK >= P.s1_row;
        float* b1 = s1 ? P.bias_s1 : P.bias;
        float* b2 = s1 ? P.bias2_s1 : P.bias2;
        if (n0 < P.bias_end) bias = b1 + n0;
        else if (b2 && n0 >= P.bias2_begin) bias = b2 + (n0 - P.bias2_begin);
    }
    tn256_segment(smem, (const bf16*)P.Y, (const bf16*)P.X, P.ldy, P.ldx, tn * T_BM, tk * T_BN, t0, nk, P.C, P.ldc,
                  P.chunks > 1, lane, wave, bias, true);
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
// Tuning knobs travel with the caller (engine object / operator call): no mutable process-wide launcher state.
// The only process-wide data are write-once caches of device facts (CU count, LDS opt-in), set under std::call_once.
static std::once_flag g_init_flag;
static int g_num_cu = 256;
static const GemmTune g_default_tune;
template <typename F>
static void allow_lds(F f) { (void)hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES); }

static void gemm_init_impl() {
    allow_lds(gemm_nt_kernel<EPI_BF16>); allow_lds(gemm_nt_kernel<EPI_QKV>); allow_lds(gemm_nt_kernel<EPI_GELU>);
    allow_lds(gemm_nt_kernel<EPI_RESID>); allow_lds(gemm_nt_kernel<EPI_F32>); allow_lds(gemm_nt_kernel<EPI_PATCH>);
    allow_lds(gemm_nt_kernel<EPI_DGELU>); allow_lds(gemm_nt_kernel<EPI_QKV_ELU>);
    allow_lds(gemm_nt_kernel<EPI_GELU_DG>); allow_lds(gemm_nt_kernel<EPI_MULAUX>);
#define ALLOW256(MODE) do { (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<MODE, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES); \
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<MODE, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS_BYTES); \
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<MODE, 5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS_BYTES); } while (0)
    ALLOW256(EPI_BF16); ALLOW256(EPI_QKV); ALLOW256(EPI_GELU); ALLOW256(EPI_RESID); ALLOW256(EPI_F32); ALLOW256(EPI_PATCH); ALLOW256(EPI_DGELU); ALLOW256(EPI_QKV_ELU); ALLOW256(EPI_GELU_DG); ALLOW256(EPI_MULAUX);
#undef ALLOW256
#define ALLOWR(MODE) do { (void)hipFuncSetAttribute((const void*)gemm_ntr_kernel<MODE, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, r_lds_bytes(8)); \
        (void)hipFuncSetAttribute((const void*)gemm_ntr_kernel<MODE, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, r_lds_bytes(10)); } while (0)
    ALLOWR(EPI_BF16); ALLOWR(EPI_QKV); ALLOWR(EPI_GELU); ALLOWR(EPI_RESID); ALLOWR(EPI_F32); ALLOWR(EPI_PATCH); ALLOWR(EPI_DGELU); ALLOWR(EPI_QKV_ELU); ALLOWR(EPI_GELU_DG); ALLOWR(EPI_MULAUX);
#undef ALLOWR
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm_tn256_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES);
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        g_num_cu = prop.multiProcessorCount;
}
static void gemm_init_once() { std::call_once(g_init_flag, gemm_init_impl); }

int uvit_gemm_nt_launch(int mode, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                        const GemmEpi* epi, hipStream_t s, const GemmTune* tune, int* tail_rows_out) {
    const GemmTune& tu = tune ? *tune : g_default_tune;
    const int nt_variant = tu.nt_variant;
    if (tail_rows_out) *tail_rows_out = 0;
    if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N % 8) || (lda % 8) || (ldw % 8) || (epi->ldo % 4))
        return UVIT_ERR_SHAPE;
    gemm_init_once();
    // variant: 0 = 128x128 (any shape, two workgroups per CU), 1 = 256x256 staggered (one per CU), 5 = 320x256 (same
    // kernel, 5 row tiles per wave), 3 = auto.
    // Auto takes the staggered kernel for every shape it supports.  In warm micro-benchmarks (operands resident in the
    // Infinity Cache) the 128x128 kernel wins the N = 768 shapes by 10-25 % (profiles/round1_gemm_variants_v2.txt),
    // but inside the step, where operands come from HBM, its one-K-tile prefetch distance costs it 28-44 % and the
    // deeper ring of the staggered kernel wins everywhere: A/B of the whole step on one box 30.2 -> 29.6 ms.
    // Tile height: 320 rows when that saves > 10 % of rounds x rows (N = 768 at M = 25216: 237 tiles = 1 round
    // instead of 297 = 1.16 rounds of 256-row tiles).
    const bool shape_ok = (N % 256) == 0 && M >= 1024 && K >= 128 && (K % 64) == 0 &&
                          (size_t)M * lda < 0xFFFFFFFFull && (size_t)N * ldw < 0xFFFFFFFFull;     // 32-bit operand offsets
    int variant = shape_ok ? nt_variant : 0;
    int mt = 4;
    // ring kernel (2 workgroups per CU): 6 = 128-row tiles, 7 = 160-row tiles
    const bool ring_ok = (N % R_BN) == 0 && M >= 128 && K >= 64 && (K % R_BK) == 0 &&
                         (size_t)M * lda < 0xFFFFFFFFull && (size_t)N * ldw < 0xFFFFFFFFull;
    // auto: the residual epilogue (fp32 stream read + write + bf16 branch copy: the heaviest epilogue per flop) on narrow
    // outputs goes to the ring kernel, whose two workgroups per CU overlap one's epilogue with the other's K loop -- unless the
    // 320-row tiles of the staggered kernel fill the CUs in ONE round (>= 85 % of them busy), where its stronger K loop wins
    // (tools/bench_gemm_variants.py, M = 25216, N = 768: K = 3072 122.8 vs 152.7 us, K = 768 54.0 vs 53.8 us -- since the
    // 320-row kernel no longer spills inside its K loop; ViT-L at bs = 64 gives 160 such tiles on 256 CUs and stays on the ring)
    const long tiles5 = (long)((M + 319) / 320) * (N / 256 > 0 ? N / 256 : 1);
    const bool one_full_round5 = shape_ok && tiles5 <= g_num_cu && tiles5 * 100 >= (long)g_num_cu * 85;
    // Round 4: auto no longer sends anything to the ring kernel.  Re-measured at the ViT-L shapes it was kept for (tools/bench_gemm.py --cold style,
    // residual epilogue, M = 12608 / 25216, N = 1024): K = 1024: ring 85.8 / 146.0 us against 67.7 / 117.9 us for the 256-row staggered kernel,
    // K = 4096: 191.4 / 305.6 against 125.1 / 239.9 us -- the staggered kernel's K loop and epilogue have moved since round 2, the ring's have not
    // (profiles/round4_gemm_large_resid_variants.txt).  UVIT_AUTO_RING=1 restores the old rule for A/B runs; variants 6 / 7 still select it.
    static const bool auto_ring_env = getenv("UVIT_AUTO_RING") && getenv("UVIT_AUTO_RING")[0] == '1';
    const bool auto_ring = auto_ring_env && nt_variant == 3 && mode == EPI_RESID && ring_ok && N <= 1024 && M >= 160 * 8 && !one_full_round5;
    if (((nt_variant == 6 || nt_variant == 7) && ring_ok) || auto_ring) {
        const int rmt = nt_variant == 6 ? 8 : 10;
        const int rgrid = ((M + 16 * rmt - 1) / (16 * rmt)) * (N / R_BN);
        const bf16* a_ = (const bf16*)A; const bf16* w_ = (const bf16*)W;
#define LR(MODE) do { if (rmt == 8) hipLaunchKernelGGL((gemm_ntr_kernel<MODE, 8>), dim3(rgrid), dim3(R_THREADS), r_lds_bytes(8), s, a_, w_, M, N, K, lda, ldw, *epi); \
        else hipLaunchKernelGGL((gemm_ntr_kernel<MODE, 10>), dim3(rgrid), dim3(R_THREADS), r_lds_bytes(10), s, a_, w_, M, N, K, lda, ldw, *epi); } while (0)
        switch (mode) {
            case EPI_BF16: LR(EPI_BF16); break;
            case EPI_QKV: if (N % 3) return UVIT_ERR_SHAPE; LR(EPI_QKV); break;
            case EPI_GELU: LR(EPI_GELU); break;
            case EPI_RESID: LR(EPI_RESID); break;
            case EPI_F32: LR(EPI_F32); break;
            case EPI_PATCH: LR(EPI_PATCH); break;
            case EPI_DGELU: LR(EPI_DGELU); break;
            case EPI_GELU_DG: LR(EPI_GELU_DG); break;
            case EPI_MULAUX: LR(EPI_MULAUX); break;
            case EPI_QKV_ELU: if (N % 3) return UVIT_ERR_SHAPE; LR(EPI_QKV_ELU); break;
            default: return UVIT_ERR_ARG;
        }
#undef LR
        return uvit_check_launch();
    }
    if (variant == 6 || variant == 7) variant = 3;          // shape not supported by the ring kernel: auto
    if (variant == 5) { variant = 1; mt = 5; }
    else if (variant == 3) {
        variant = 1;
        const int tn_ = N / T_BN;
        const long c4 = (long)((((M + 255) / 256) * tn_ + g_num_cu - 1) / g_num_cu) * 256;
        const long c5 = (long)((((M + 319) / 320) * tn_ + g_num_cu - 1) / g_num_cu) * 320;
        if (c5 * 10 < c4 * 9) mt = 5;
    }
    // 256-row tiles that overflow whole rounds of the CUs by only a few tiles would run a nearly empty last round: the
    // overflowing row tiles go to the 128x128 kernel instead (second launch below)
    int m_tail = 0;
    if (variant == 1 && mt == 4 && nt_variant == 3 && mode != EPI_PATCH && !epi->rowmap) {      // (a row list does not split: its rows are not an offset apart)
        const int tiles_n = N / T_BN, tiles = ((M + T_BM - 1) / T_BM) * tiles_n;
        const int rounds = tiles / g_num_cu, over = tiles - rounds * g_num_cu;
        // (round 3 tried half a round of overflow -- QKV at bs = 128: 891 tiles = 3 rounds + 123: 93.2 -> 88.9 us alone, but inside the
        //  step the 3-round launch + its 128x128 tail take the same 91.7 us as the 4-round launch: profiles/round3_gemm_pair_kernel_experiment.txt)
        if (rounds >= 1 && over > 0 && over * 4 <= g_num_cu) {
            const int rows_a = (rounds * g_num_cu / tiles_n) * T_BM;
            if (rows_a > 0 && rows_a < M) { m_tail = M - rows_a; M = rows_a; }
        }
    }
    GemmEpi epi_g = *epi;
    epi_g.ngroup = tu.nt_group;
    epi = &epi_g;
    const int bm = mt == 5 ? 320 : T_BM;
    const int grid = variant == 1 ? ((M + bm - 1) / bm) * (N / T_BN) : ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const size_t lds = 4 * STAGE_BYTES;
    const bf16* a = (const bf16*)A; const bf16* w = (const bf16*)W;
    // persistent form of the 256-row kernel whenever a CU would otherwise run several workgroups back to back
    const int pgrid = g_num_cu & ~7;
    // (not for the epilogues that multiply by a row operand: with the persistent form's 4-KiB staging area their fp32 blocks go
    // through one at a time and they lose 2-3 %, tools/bench_gemm.py --persist: mulaux 153 vs 149 us, dgelu 169 vs 164 us;
    // all other epilogues gain 6-10 %)
    const bool persist = variant == 1 && mt == 4 && tu.nt_persist && pgrid >= 8 && grid > pgrid && mode != EPI_MULAUX && mode != EPI_DGELU;
#define L(MODE) do { if (variant == 1 && mt == 5) hipLaunchKernelGGL((gemm_nt256_kernel<MODE, 5, false>), dim3(grid), dim3(T_THREADS), T5_LDS_BYTES, s, a, w, M, N, K, lda, ldw, *epi); \
        else if (persist) hipLaunchKernelGGL((gemm_nt256_kernel<MODE, 4, true>), dim3(pgrid), dim3(T_THREADS), TP_LDS_BYTES, s, a, w, M, N, K, lda, ldw, *epi); \
        else if (variant == 1) hipLaunchKernelGGL((gemm_nt256_kernel<MODE, 4, false>), dim3(grid), dim3(T_THREADS), T_LDS_BYTES, s, a, w, M, N, K, lda, ldw, *epi); \
        else hipLaunchKernelGGL(gemm_nt_kernel<MODE>, dim3(grid), dim3(GEMM_THREADS), lds, s, a, w, M, N, K, lda, ldw, *epi); } while (0)
    switch (mode) {
        case EPI_BF16: L(EPI_BF16); break;
        case EPI_QKV: if (N % 3) return UVIT_ERR_SHAPE; L(EPI_QKV); break;
        case EPI_GELU: L(EPI_GELU); break;
        case EPI_RESID: L(EPI_RESID); break;
        case EPI_F32: L(EPI_F32); break;
        case EPI_PATCH: L(EPI_PATCH); break;
        case EPI_DGELU: L(EPI_DGELU); break;
        case EPI_GELU_DG: L(EPI_GELU_DG); break;
        case EPI_MULAUX: L(EPI_MULAUX); break;
        case EPI_QKV_ELU: if (N % 3) return UVIT_ERR_SHAPE; L(EPI_QKV_ELU); break;
        default: return UVIT_ERR_ARG;
    }
#undef L
    int rc = uvit_check_launch();
    if (rc == UVIT_OK && m_tail > 0) {
        // remaining rows: every row-indexed epilogue operand moves with the row offset
        GemmEpi t = *epi;
        const size_t r0 = (size_t)M;
        const bool f32out = mode == EPI_RESID || mode == EPI_F32 || mode == EPI_PATCH;
        t.out = f32out ? (void*)((float*)t.out + r0 * t.ldo) : (void*)((bf16*)t.out + r0 * t.ldo);
        if (t.out2) t.out2 = (void*)((bf16*)t.out2 + r0 * t.ldo);
        if (t.resid) t.resid = t.resid + r0 * t.ldo;
        if (t.aux) t.aux = (const void*)((const bf16*)t.aux + r0 * t.ldo);
        t.row0 = epi->row0 + (int)r0;
        GemmTune generic;                                  // the tail always runs on the 128x128 kernel
        generic.nt_variant = 0;
        rc = uvit_gemm_nt_launch(mode, a + r0 * lda, W, m_tail, N, K, lda, ldw, &t, s, &generic, nullptr);
        if (tail_rows_out) *tail_rows_out = m_tail;
    }
    return rc;
}

int uvit_gemm_tn_launch(const void* Y, const void* X, int M, int Nn, int Kk, int ldy, int ldx, float* C,
                        int ldc, int allow_split, hipStream_t s, const GemmTune* tune) {
    const GemmTune& tu = tune ? *tune : g_default_tune;
    const int tn_variant = tu.tn_variant, tn_target = tu.tn_target > 0 ? tu.tn_target : 512;
    if (M <= 0 || (M % BK) || (Nn % 8) || (Kk % 8) || (ldy % 8) || (ldx % 8) || (ldc % 4)) return UVIT_ERR_SHAPE;
    gemm_init_once();
    const int nm = M / BK;
    if (tn_variant == 1 && (Nn % T_BM) == 0 && (Kk % T_BN) == 0 && nm >= 8) {
        // 256x256 tiles, one workgroup per CU: split the token reduction until ~256 workgroups exist
        const int tiles256 = (Nn / T_BM) * (Kk / T_BN);
        int split = allow_split ? (256 + tiles256 / 2) / tiles256 : 1;
        if (split > nm / 4) split = nm / 4;
        if (split < 1) split = 1;
        const int steps256 = (nm + split - 1) / split;
        split = (nm + steps256 - 1) / steps256;
        hipLaunchKernelGGL(gemm_tn256_kernel, dim3(tiles256, split), dim3(T_THREADS), T_LDS_BYTES, s, (const bf16*)Y,
                           (const bf16*)X, M, Nn, Kk, ldy, ldx, C, ldc, steps256);
        return uvit_check_launch();
    }
    const int tiles = ((Nn + BM - 1) / BM) * ((Kk + BN - 1) / BN);
    // split the token reduction until ~1024 workgroups are in flight (2 resident per CU x 256 CUs, two rounds)
    int split = 1;
    if (allow_split) {
        split = tn_target / tiles;
        if (split > nm / 8) split = nm / 8;
        if (split < 1) split = 1;
    }
    const int steps = (nm + split - 1) / split;
    split = (nm + steps - 1) / steps;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, split), dim3(GEMM_THREADS), TN_LDS_BYTES, s, (const bf16*)Y,
                       (const bf16*)X, M, Nn, Kk, ldy, ldx, C, ldc, steps);
    return uvit_check_launch();
}

// Grouped wgrad: every problem needs Nn, Kk multiples of 256 (bias segments too) and a token count that is a
// multiple of 64.  Returns UVIT_ERR_SHAPE without launching when the group does not qualify (callers fall back to
// per-problem launches).
bool uvit_gemm_tn_group_ok(const TnProb* probs, int n, const GemmTune* tune) {
    if (n < 1 || n > UVIT_TN_GROUP_MAX || (tune ? tune : &g_default_tune)->tn_variant == 0) return false;
    for (int i = 0; i < n; ++i) {
        const TnProb& q = probs[i];
        if (q.M <= 0 || (q.M % BK) || q.M / BK < 8 || (q.Nn % T_BM) || (q.Kk % T_BN) || (q.ldy % 8) || (q.ldx % 8) || (q.ldc % 4)) return false;
        if (q.bias && ((q.bias_end % T_BM) || (q.bias2 && (q.bias2_begin % T_BM)))) return false;
        // (the planner needs a 2-chunk plan to exist: at least 16 K-tiles, the boundary at or just above half of them)
        if (q.s1_row && (!q.bias || !q.bias_s1 || (q.bias2 && !q.bias2_s1) || (q.s1_row % BK) || q.s1_row >= q.M || q.M / BK < 16 ||
                         (q.M / BK + 1) / 2 != q.s1_row / BK)) return false;
    }
    return true;
}

int uvit_gemm_tn_group_launch(const TnProb* probs, int n, hipStream_t s, const GemmTune* tune) {
    if (!uvit_gemm_tn_group_ok(probs, n, tune)) return UVIT_ERR_SHAPE;
    const int group_chunks = (tune ? tune : &g_default_tune)->group_chunks;
    gemm_init_once();
    TnGroup G;
    G.nprob = n;
    int nm_max = 0;
    for (int i = 0; i < n; ++i) {
        G.p[i] = probs[i];
        G.p[i].nm = probs[i].M / BK;
        G.p[i].tiles_n = probs[i].Nn / T_BM;
        G.p[i].tiles_k = probs[i].Kk / T_BN;
        nm_max = nm_max > G.p[i].nm ? nm_max : G.p[i].nm;
    }
    // token chunks per tile: minimise rounds x chunk main loop + the atomic traffic of combining the chunks, in units
    // of one 64-token K-tile (~1.7 us on MI355X; fp32 atomics drain at ~1.5 TB/s = 2.6 MB per unit) -- fitted to
    // tools/bench_tn_sweep.py (profiles/round1_wgrad_group.txt)
    auto plan = [&](int sp, int& items) {
        const int L = (nm_max + sp - 1) / sp;
        items = 0;
        for (int i = 0; i < n; ++i) items += G.p[i].tiles_n * G.p[i].tiles_k * ((G.p[i].nm + L - 1) / L);
        return L;
    };
    int best_sp = 1; double best_cost = 1e30;
    const int sp_max = nm_max / 8 < 16 ? nm_max / 8 : 16;
    auto stream_ok = [&](int L) {                     // no chunk [c L, (c + 1) L) of a two-stream problem straddles its stream boundary
        for (int i = 0; i < n; ++i) if (probs[i].s1_row && ((probs[i].s1_row / BK) % L)) return false;
        return true;
    };
    for (int sp = 1; sp <= (sp_max < 1 ? 1 : sp_max); ++sp) {
        int items; const int L = plan(sp, items);
        if (!stream_ok(L)) continue;
        double atomic_bytes = 0.0;
        for (int i = 0; i < n; ++i) {
            const int ch = (G.p[i].nm + L - 1) / L;
            if (ch > 1) atomic_bytes += 4.0 * probs[i].Nn * probs[i].Kk * ch;
        }
        const double cost = (double)((items + g_num_cu - 1) / g_num_cu) * L + atomic_bytes / 2.6e6 + 6.0;
        if (cost < best_cost) { best_cost = cost; best_sp = sp; }
    }
    if (group_chunks > 0) best_sp = group_chunks < (nm_max / 4 > 1 ? nm_max / 4 : 1) ? group_chunks : (nm_max / 4 > 1 ? nm_max / 4 : 1);
    int items; int L = plan(best_sp, items);
    if (!stream_ok(L)) {                               // a forced chunk count that would straddle the stream boundary: the 2-chunk plan always
        best_sp = 2; L = plan(best_sp, items);         // qualifies (uvit_gemm_tn_group_ok checked that the boundary is half the K-tiles)
        if (!stream_ok(L)) return UVIT_ERR_SHAPE;
    }
    G.max_chunks = 0;
    for (int i = 0; i < n; ++i) {
        G.p[i].chunk_steps = L;
        G.p[i].chunks = (G.p[i].nm + L - 1) / L;
        G.max_chunks = G.max_chunks > G.p[i].chunks ? G.max_chunks : G.p[i].chunks;
    }
    hipLaunchKernelGGL(gemm_tn256_group_kernel, dim3(items), dim3(T_THREADS), T_LDS_BYTES, s, G);
    return uvit_check_launch();
}
