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
//   bwd     : ONE fused kernel (round 3): S^T, dP^T = V.dO^T -> P, dS -> dQ^T = K^T.dS^T from the accumulators; P, dS cross LDS once
//             for dV^T = dO^T.P, dK^T = Q^T.dS; dS also leaves as bf16 and attn_dbias_reduce_kernel sums it over the batch
#include <mutex>
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
#define NT_MAX 13            // 13 * 16 = 208 >= 197 tokens
#define ROWS_PAD 224         // 14 * 16: k-steps pair two 16-row tiles
#define IMG_BYTES (ROWS_PAD * 128)

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

// One 1-KiB LDS-DMA piece (global_load_lds, 16 B per lane, no VGPR round trip): rows 8 rb .. 8 rb + 7 of a [rows][64] bf16
// source land in the swizzled image -- lane l writes LDS byte rb * 1024 + 16 l = img_off(row, chunk) for row = 8 rb + (l >> 3),
// chunk = (l & 7) ^ (row & 7), so the swizzle is applied on the SOURCE side.  Rows >= n_valid re-read row n_valid - 1 (the DMA
// cannot zero-fill): the user must mask padded rows arithmetically (the forward's -1e30 bias columns give p = 0 for padded
// keys, and 0 x finite = 0 in P.V), so they only need to be finite.  `img` and `rb` must be wave-uniform.
__device__ __forceinline__ void dma_rows8(char* img, int rb, const bf16* src, size_t stride, int n_valid, int lane) {
    const int row = 8 * rb + (lane >> 3);
    const int chunk = (lane & 7) ^ (lane >> 3);
    const int r = row < n_valid ? row : n_valid - 1;
    __builtin_amdgcn_global_load_lds(GLB_PTR(void, src + (size_t)r * stride + chunk * 8), LDS_PTR(void, img + rb * 1024), 16, 0, 0);
}
#define IMG_PIECES (ROWS_PAD / 8)          // 28 DMA pieces per image

// After an explicit `s_waitcnt vmcnt(0)`: tell the compiler's wait-count tracking that a prefetched register HAS landed (it inserts
// its own, by then free, wait in front of this use).  Without it the first real use -- on the far side of a loop back-edge and
// behind newly issued stores, which the in-order vmcnt cannot skip -- waits for those as well.
__device__ __forceinline__ void landed(bf16x8& v) { asm volatile("" : "+v"(v)); }

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

#ifdef ATTN_DEBUG
int uvit_attn_fwd_launch(const void*, const float*, void*, float*, int, int, int, int, float, float, uint32_t, uint32_t, hipStream_t, const int*);
extern "C" int uvit_debug_attn_fwd(const void* qkv, const float* biasP, void* out, float* lse, int B, int H, int N, int NP, float scale,
                                   float p_drop, void* stream) {
    return uvit_attn_fwd_launch(qkv, biasP, out, lse, B, H, N, NP, scale, p_drop, 1u, 0u, (hipStream_t)stream, nullptr);
}
#endif

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NW, bool HAS_BIAS>
__global__ __launch_bounds__(NW * 64, 4)      // two 7-wave workgroups per CU need <= 128 VGPRs
void attn_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ biasP, bf16* __restrict__ out,
                     float* __restrict__ lse, int H, int N, int NP, float scale, uint32_t drop_thr,
                     float inv_keep, uint32_t drop_key, int ncu, const int* __restrict__ bmap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* kimg = smem;
    char* vimg = smem + IMG_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const bf16* base = qkv + (size_t)b * N * ld + h * HD;
    const uint32_t bh_rng = (uint32_t)__builtin_amdgcn_readfirstlane(bmap ? bmap[b] * H + h : bh);
    // Two workgroups share a CU and start together: left alone they load their images together and compute together.  The
    // second workgroup of every CU (first round only: ids ncu .. 2 ncu - 1) starts 8k cycles late, so that from then on one
    // workgroup's image load runs under the other's tile loop (59 -> 55.5 us without dropout, 66 -> 61.6 us with).
    if (blockIdx.x >= (unsigned)ncu && blockIdx.x < 2u * (unsigned)ncu) __builtin_amdgcn_s_sleep(127);
    // The Q fragments of a wave's first tile are requested first (the tile loop needs them before anything else), then the K / V
    // images by LDS-DMA, 56 pieces of 8 rows over the NW waves (round 2: in-kernel stamps showed 8.5-12k of a workgroup's ~32k
    // cycles in the register-staged image load and another ~4k waiting for Q behind it; tools/stamp_attn.py).
    bf16x8 qn[2];
    {
        const int q0 = wave * 16 + li, qr0 = q0 < N ? q0 : N - 1;
        qn[0] = *(const bf16x8*)(base + (size_t)qr0 * ld + g * 8);
        qn[1] = *(const bf16x8*)(base + (size_t)qr0 * ld + 32 + g * 8);
    }
    for (int p = wave; p < 2 * IMG_PIECES; p += NW) {
        const int img = p / IMG_PIECES, rb = p - img * IMG_PIECES;
        dma_rows8(smem + img * IMG_BYTES, rb, base + (size_t)(1 + img) * C, ld, N, lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed(qn[0]); landed(qn[1]);
    __syncthreads();
    const int nt = (N + 15) >> 4, nt2 = (nt + 1) >> 1;

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
        // bh_rng (drop-path sample lists): sample slot b of a COMPACT batch is sample bmap[b] of the step's batch -- the dropout draws are
        // indexed by the sample, so a compacted launch draws what the dense one does
        const uint32_t rowpair = (bh_rng * N + q) * (uint32_t)(NP >> 1);
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
        // the next tile's Q fragments were requested a whole tile ago: mark them landed before this tile's stores are issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        landed(qn[0]); landed(qn[1]);
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
// backward, FUSED (round 3): dQ, dK, dV and dS from ONE recomputation of P.
//
// One 13-wave workgroup per (batch, head).  Wave w owns the 16 queries 16 w .. 16 w + 15 for the whole sample ("query on the
// lane": S^T = K.Q^T as in the forward) and the workgroup walks the keys in steps of 32.  Per step i:
//   A_i (every wave): S^T, dP^T of its queries against the step's 2 key tiles -> P, dS in registers (fp32) ->
//        dQ^T += K^T.dS^T straight from the accumulators (the contraction index -- the key -- is the accumulator's row index);
//        P (dropout applied) and dS are written ONCE, as bf16, into the step buffers [208 queries][32 keys];
//   barrier;
//   B_i (waves 0..7, in program order beside A_{i+1}): dV^T / dK^T of the step's 32 keys = dO^T.P / Q^T.dS, contracted over ALL
//        queries by transposed reads of the Q / dO images and of the step buffers; the 16 x 16 result tiles are written (bf16) into
//        the K / V image rows of step i, which nobody reads any more, and leave for HBM as full 128-B rows one barrier later;
//        waves 8..12 meanwhile stream those rows and the dS step buffer (for the bias gradient) to HBM.
// The step buffers are double-buffered, so there is one barrier per step.  Five MFMA products and one exp per score instead of
// the seven products and two exps of the two-kernel form, Q / K / V / dO are read once (310 MB per layer at ViT-B bs = 128 instead
// of 660 MB), and 3.25 waves per SIMD at <= 128 VGPRs instead of 1.75 at 256: the two older kernels kept the wave's bias rows and
// bias-gradient accumulators (2 x 52 VGPRs) in registers.  Here the bias tile comes from L2 per (query tile, key tile) and the
// bias gradient leaves the kernel as dS (bf16, 143 MB per layer): attn_dbias_reduce_kernel sums it over the batch.  LDS float
// atomics are no alternative: ds_add_f32 retires one lane every 3 cycles (profiles/round3_micro_lds_atomic.txt).
//
// LDS: K, V, Q, dO images [208 rows][128 B] (the forward's swizzle) = 104 KiB + 2 x 2 step buffers of 13 KiB = 156 KiB.
// Step buffer: 64-B rows of eight 8-B slots (4 keys each); slot s of row q sits at s ^ f(q), f(q) = bits (q2, q3, q1): the
// 8-B writes of a 16-lane group (16 consecutive rows, one slot) fall in 16 different bank pairs, and a transposed read's 32-lane
// half (8 consecutive rows x 4 slots) covers all 64 banks once.
// ------------------------------------------------------------------------------------------
#define FB_WAVES 13
#define FB_ROWS (NT_MAX * 16)                 // 208
// Diagnostic build only (-DATTN_STAMP, tools/stamp_attn.py): s_memtime stamps of waves 0 and 8 of every workgroup; never in libuvit.so
#ifdef ATTN_STAMP
__device__ unsigned long long g_attn_stamps[2048 * 2 * 16 * 2];
#define ASTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (lane == 0 && (wave == 0 || wave == 8) && blockIdx.x < 2048) g_attn_stamps[(blockIdx.x * 2 + (wave >> 3)) * 16 + (k)] = t_; } while (0)
// sub-stage stamps of step 3 (waves 0 and 8): second half of the buffer
#define BSTAMP(k) do { if (i == 3) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (lane == 0 && (wave == 0 || wave == 8) && blockIdx.x < 2048) g_attn_stamps[2048 * 32 + (blockIdx.x * 2 + (wave >> 3)) * 16 + (k)] = t_; } } while (0)
#define ASTAMP_ID() do { if (lane == 0 && wave == 0 && blockIdx.x < 2048) \
        g_attn_stamps[blockIdx.x * 32 + 15] = __builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32); } while (0)
extern "C" int uvit_debug_attn_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)) == hipSuccess ? 0 : -3;
}
int uvit_attn_bwd_fused_launch(const void*, const void*, const void*, const float*, const float*, float*, void*, void*, int, int, int, int,
                               int, float, float, uint32_t, uint32_t, hipStream_t, const int*);
extern "C" int uvit_debug_attn_bwd_fused(const void* qkv, const void* o, const void* d_o, const float* biasP, const float* lse, float* delta,
                                         void* dqkv, void* ds_ws, float* slab, int B, int H, int N, float p_drop, void* stream) {
    (void)slab;
    return uvit_attn_bwd_fused_launch(qkv, o, d_o, biasP, lse, delta, dqkv, ds_ws, ds_ws != nullptr, B, H, N, 208, 0.125f, p_drop, 1u, 0u, (hipStream_t)stream, nullptr);
}
#else
#define ASTAMP(k)
#define BSTAMP(k)
#define ASTAMP_ID()
#endif
#define FB_IMG (FB_ROWS * 128)                // 26,624 B
#define FB_SB (FB_ROWS * 64)                  // 13,312 B
#define FB_LDS (4 * FB_IMG + 4 * FB_SB)       // 159,744 B
#define FB_BWAVES 8                           // waves that run the B phase (one (product, d-tile) each, both key tiles of the step)

__device__ __forceinline__ int sb_off(int q, int slot) {
    const int f = (((q >> 2) & 1) << 2) | (((q >> 3) & 1) << 1) | ((q >> 1) & 1);
    return q * 64 + ((slot ^ f) << 3);
}
// B operand of the B phase: element j of lane (g, i) = buf[query (j<4 ? r_lo : r_hi) + 4g + (j&3)][key 16 tt + i]
__device__ __forceinline__ bf16x8 sb_col_frag(const char* sb, int r_lo, int r_hi, int tt, bool has_hi, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sb + sb_off(r_lo + 4 * g + q, 4 * tt + p)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sb + sb_off(r_hi + 4 * g + q, 4 * tt + p)));
    if (!has_hi) hi = s16x4{0, 0, 0, 0};
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NT_C = 13: the token count is known at compile time to need all 13 query tiles (192 < N <= 208: ViT-B/16 and ViT-L/16 at 224),
// every loop is unrolled and every wave is active; NT_C = 0: any N <= 208 (runtime tile counts).
template <bool HAS_BIAS, int NT_C>
__global__ __launch_bounds__(FB_WAVES * 64)
void attn_bwd_fused_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o_fwd, const bf16* __restrict__ d_o,
                           const float* __restrict__ biasP, const float* __restrict__ lse, float* __restrict__ delta,
                           bf16* __restrict__ dqkv, bf16* __restrict__ ds_out, int H, int N, int NP, float scale,
                           uint32_t drop_thr, float inv_keep, uint32_t drop_key, const int* __restrict__ bmap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kimg = smem;
    char* const vimg = smem + FB_IMG;
    char* const qimg = smem + 2 * FB_IMG;
    char* const doimg = smem + 3 * FB_IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    const int C = H * HD;
    const size_t ld = 3 * (size_t)C;
    const int nt = NT_C ? NT_C : (N + 15) >> 4;
    const int nsteps = (nt + 1) >> 1;
    const bf16* base = qkv + (size_t)b * N * ld + h * HD;
    const bf16* dobase = d_o + (size_t)b * N * C + h * HD;
    const bool active = NT_C ? true : wave < nt;
    const int q = wave * 16 + li, qr = q < N ? q : N - 1;
    ASTAMP(0); ASTAMP_ID();

    // ---- images by LDS-DMA: row blocks of 8; K and V first (rb-interleaved), then Q and dO; blocks beyond the last tile are never read
    for (int p = wave; p < 4 * (FB_ROWS / 8); p += FB_WAVES) {
        int img, rb;
        if (p < 2 * (FB_ROWS / 8)) { img = p & 1; rb = p >> 1; } else { img = 2 + (p >= 3 * (FB_ROWS / 8)); rb = p - img * (FB_ROWS / 8); }
        if (rb * 8 >= nt * 16) continue;
        if (img == 3) dma_rows8(doimg, rb, dobase, (size_t)C, N, lane);
        else dma_rows8(smem + img * FB_IMG, rb, base + (img == 2 ? 0 : (size_t)(1 + img) * C), ld, N, lane);
    }
    // ---- this wave's queries: delta = rowsum(dO o O), LSE  (their Q / dO fragments -- the B operands of S^T and dP^T -- are re-read
    //      from the images every step: 16 registers the 128-VGPR budget does not have)
    float dl = 0.f, lse_q = 1e30f;                    // padded query lanes: p = exp2(.. - 1e30) = 0, so they add nothing to dK / dV
    if (active) {
        const size_t orow = ((size_t)b * N + qr) * C + h * HD;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 dof = *(const bf16x8*)(d_o + orow + kk * 32 + g * 8);
            const bf16x8 of = *(const bf16x8*)(o_fwd + orow + kk * 32 + g * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) dl += bf2f(dof[j]) * bf2f(of[j]);
        }
        dl = group_sum4(dl);
        if (q < N) {
            lse_q = lse[(size_t)bh * N + q];
            if (g == 0) delta[(size_t)bh * N + q] = dl;
        }
    }
    const uint32_t bh_rng = (uint32_t)__builtin_amdgcn_readfirstlane(bmap ? bmap[b] * H + h : bh);     // (compact batch: see the forward)
    const uint32_t rowpair = (bh_rng * N + q) * (uint32_t)(NP >> 1);
    const float cs = scale * LOG2E;
    const float* brow = HAS_BIAS ? biasP + ((size_t)h * NP + (q < NP ? q : NP - 1)) * NP + 4 * g : nullptr;
    // the bias tiles of a step are requested one step ahead
    float4 bnext[2];
    auto bias_fetch = [&](int i) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = 2 * i + tt;
            if constexpr (HAS_BIAS) {
                bnext[tt] = (t < nt && active) ? *(const float4*)(brow + t * 16) : make_float4(NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG);
            } else {
                const int k0 = t * 16 + 4 * g;
                bnext[tt] = make_float4(k0 < N ? 0.f : NEG_BIG, k0 + 1 < N ? 0.f : NEG_BIG, k0 + 2 < N ? 0.f : NEG_BIG, k0 + 3 < N ? 0.f : NEG_BIG);
            }
        }
    };
    bias_fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(1);
    __syncthreads();
    ASTAMP(2);

    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B-phase job of this wave: product pj (0: dV from P and dO, 1: dK from dS and Q), d-tile dtj
    const int pj = wave >> 2, dtj = wave & 3;

    // ================= A_i: this wave's 16 queries against the 2 key tiles of step i
    auto A_step = [&](int i) {
        char* pb = smem + 4 * FB_IMG + (i & 1) * FB_SB;
        char* db = pb + 2 * FB_SB;
        const int t0 = 2 * i;
        const bool has1 = t0 + 1 < nt;
        const float4 bcur[2] = {bnext[0], bnext[1]};
        if (i + 1 < nsteps) bias_fetch(i + 1);
        // NO wave-uniform branch may sit between an MFMA and the first VALU read of its result: hipcc's hazard recogniser pads the
        // MFMA -> VALU wait states along the layout (fall-through) path only, and a taken branch that skips a block lands on the
        // consumer too early (seen here: accumulator elements 1 and 2 stale after a skipped `if`).  So the dropout draw -- the one
        // conditional block of the step -- comes BEFORE the MFMAs.
        BSTAMP(4);
        bool k4[2][4] = {{true, true, true, true}, {true, true, true, true}};
#ifndef ATTN_HAZARD_DEMO
        if (drop_thr) {
            keep4(drop_key, rowpair, t0 * 16 + 4 * g, drop_thr, k4[0]);
            keep4(drop_key, rowpair, t0 * 16 + 16 + 4 * g, drop_thr, k4[1]);
        }
#endif
        BSTAMP(5);
        // Staged so that at most 16 operand registers are live: K rows -> S^T, V rows -> dP^T, softmax backward, K^T columns -> dQ^T.
        // (With every read of the step hoisted to the top the kernel spills ~55 registers at the 128-VGPR budget of 13 waves.)
        const int row0 = t0 * 16 + li, row1 = (has1 ? t0 + 1 : t0) * 16 + li;   // no second tile: re-read the first (its p is 0)
        f32x4 sacc[2], dp[2];
        {
            const bf16x8 k00 = row_frag(kimg, row0, g), k01 = row_frag(kimg, row0, 4 + g);
            const bf16x8 k10 = row_frag(kimg, row1, g), k11 = row_frag(kimg, row1, 4 + g);
            const bf16x8 qf0 = row_frag(qimg, q, g), qf1 = row_frag(qimg, q, 4 + g);      // B operand: this lane's query row
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            sacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k00, qf0, z, 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k10, qf0, z, 0, 0, 0);
            sacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k01, qf1, sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k11, qf1, sacc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        BSTAMP(6);
        {
            const bf16x8 v00 = row_frag(vimg, row0, g), v01 = row_frag(vimg, row0, 4 + g);
            const bf16x8 v10 = row_frag(vimg, row1, g), v11 = row_frag(vimg, row1, 4 + g);
            const bf16x8 do0 = row_frag(doimg, q, g), do1 = row_frag(doimg, q, 4 + g);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            dp[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v00, do0, z, 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v10, do0, z, 0, 0, 0);
            dp[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v01, do1, dp[0], 0, 0, 0);
            dp[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v11, do1, dp[1], 0, 0, 0);
        }
        BSTAMP(7);
#ifdef ATTN_HAZARD_DEMO      // the round-3 bug, kept as the build-time guard's test case (tools/check_mfma_hazard.py): a skipped `if` right after the MFMAs
        if (drop_thr) {
            keep4(drop_key, rowpair, t0 * 16 + 4 * g, drop_thr, k4[0]);
            keep4(drop_key, rowpair, t0 * 16 + 16 + 4 * g, drop_thr, k4[1]);
        }
#endif
        float pdv[2][4], dsv[2][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const float bb[4] = {bcur[tt].x, bcur[tt].y, bcur[tt].z, bcur[tt].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // no second tile: its bias is -1e30, so p = 0 and the tile adds nothing
                const float p = __builtin_amdgcn_exp2f(sacc[tt][r] * cs + bb[r] - lse_q);
                const float pd = k4[tt][r] ? p * inv_keep : 0.f;
                pdv[tt][r] = pd;
                dsv[tt][r] = pd * dp[tt][r] - p * dl;
            }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (tt == 0 || has1) {
                const bf16x4 pv = {f2bf(pdv[tt][0]), f2bf(pdv[tt][1]), f2bf(pdv[tt][2]), f2bf(pdv[tt][3])};
                const bf16x4 dv = {f2bf(dsv[tt][0]), f2bf(dsv[tt][1]), f2bf(dsv[tt][2]), f2bf(dsv[tt][3])};
                *(bf16x4*)(pb + sb_off(q, 4 * tt + g)) = pv;
                *(bf16x4*)(db + sb_off(q, 4 * tt + g)) = dv;
            }
        }
        const bf16x8 dsf = pack8(dsv[0], dsv[1]);
        __builtin_amdgcn_sched_barrier(0);
        BSTAMP(8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 kt = col_frag(kimg, t0 * 16, has1 ? t0 * 16 + 16 : t0 * 16, dt * 16, lane);
            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[dt], 0, 0, 0);
        }
        BSTAMP(9);
    };

    // ================= B_i (waves 0..7): d{V,K}^T of the step's 32 keys, contracted over every query
    auto B_step = [&](int i) {
        const char* pb = smem + 4 * FB_IMG + (i & 1) * FB_SB;
        const char* sb = pj ? pb + 2 * FB_SB : pb;
        const char* img = pj ? qimg : doimg;
        const bool has1 = 2 * i + 1 < nt;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // both key tiles always (a step without a second tile multiplies stale but finite slots; the result is dropped): no
        // branch between the MFMAs and the conversion of their results (see A_step)
        auto kstep = [&](int ks) {
            const bool hk = 2 * ks + 1 < nt;
            const int r_lo = 32 * ks, r_hi = hk ? r_lo + 16 : r_lo;
            const bf16x8 a = col_frag(img, r_lo, r_hi, dtj * 16, lane);
            const bf16x8 b0 = sb_col_frag(sb, r_lo, r_hi, 0, hk, lane);
            const bf16x8 b1 = sb_col_frag(sb, r_lo, r_hi, 1, hk, lane);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[1], 0, 0, 0);
        };
        if constexpr (NT_C != 0) {
#pragma unroll
            for (int ks = 0; ks < (NT_C + 1) / 2; ++ks) kstep(ks);
        } else {
            for (int ks = 0; ks < nsteps; ++ks) kstep(ks);
        }
        HAZARD_PAD();     // the loop exit is a branch: pad the MFMA -> VALU wait states by hand
        // acc[tt][r] = d{V,K}[key 32 i + 16 tt + li][d = 16 dtj + 4 g + r]  ->  the dead rows of the V / K image
        char* dst = pj ? kimg : vimg;
        const float sc = pj ? scale : 1.0f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (tt == 0 || has1) {
                const int row = 32 * i + 16 * tt + li;
                const bf16x4 v = {f2bf(acc[tt][0] * sc), f2bf(acc[tt][1] * sc), f2bf(acc[tt][2] * sc), f2bf(acc[tt][3] * sc)};
                *(bf16x4*)(dst + img_off(row, 2 * dtj + (g >> 1)) + ((g & 1) << 3)) = v;
            }
        }
    };
    // rows 32 i .. 32 i + 31 of the K / V images hold dK / dV of step i once B_i is done: 8 pieces of 8 rows, as full 128-B rows
    auto store_rows = [&](int i, int sw) {
        for (int c = sw; c < 8; c += FB_WAVES - FB_BWAVES) {
            const int img = c >> 2, row = 32 * i + 8 * (c & 3) + (lane >> 3), ch = lane & 7;
            if (row < N) {
                const uint4 v = *(const uint4*)(smem + img * FB_IMG + img_off(row, ch));
                *(uint4*)(dqkv + ((size_t)b * N + row) * ld + (size_t)(1 + img) * C + h * HD + ch * 8) = v;
            }
        }
    };
    // the dS step buffer as it stands (attn_dbias_reduce_kernel undoes the slot swizzle): 1-KiB pieces of 16 rows
    auto stream_ds = [&](int i, int sw) {
        if (!ds_out) return;
        const char* db = smem + 4 * FB_IMG + (i & 1) * FB_SB + 2 * FB_SB;
        char* dst = (char*)(ds_out + ((size_t)bh * nsteps + i) * (FB_SB / 2));
        for (int j = sw; j < nt; j += FB_WAVES - FB_BWAVES) {
            const uint4 v = *(const uint4*)(db + j * 1024 + lane * 16);
            *(uint4*)(dst + j * 1024 + lane * 16) = v;
        }
    };

    // iteration i: B_{i-1} (or the streaming of step i-1 / i-2) and A_i in ONE basic block per wave role, then the step's barrier
#pragma unroll 1
    for (int i = 0; i <= nsteps; ++i) {
        {
            if (wave < FB_BWAVES) {
                BSTAMP(1);
                if (i > 0) B_step(i - 1);
                __builtin_amdgcn_sched_barrier(0);
                BSTAMP(2);
                if (i < nsteps && active) A_step(i);
            } else {
                const int sw = wave - FB_BWAVES;
                BSTAMP(1);
                if (i > 1) store_rows(i - 2, sw);
                if (i > 0) stream_ds(i - 1, sw);
                BSTAMP(2);
                if (i < nsteps && active) A_step(i);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            BSTAMP(10);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            BSTAMP(11);
            ASTAMP(3 + i);
        }
    }
    if (wave >= FB_BWAVES) store_rows(nsteps - 1, wave - FB_BWAVES);
    if (active) {
        // dq[dt][r] = dQ[q = li][d = 16 dt + 4 g + r]: through this wave's own 16 rows of the Q image, then 16 B per lane
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x4 v = {f2bf(dq[dt][0] * scale), f2bf(dq[dt][1] * scale), f2bf(dq[dt][2] * scale), f2bf(dq[dt][3] * scale)};
            *(bf16x4*)(qimg + img_off(q, 2 * dt + (g >> 1)) + ((g & 1) << 3)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = wave * 16 + (lane >> 3) + 8 * k, ch = lane & 7;
            const uint4 v = *(const uint4*)(qimg + img_off(row, ch));
            if (row < N) *(uint4*)(dqkv + ((size_t)b * N + row) * ld + h * HD + ch * 8) = v;
        }
    }
    ASTAMP(13);
}

// Bias gradient from the dS the fused kernel streamed out: slab[h][key][q] += sum_b dS_b[h][q][key]  (slab zeroed by the launcher
// unless it accumulates).  ds = [B * H][nsteps][208 rows x 64 B] bf16 step-buffer images (slot swizzle of sb_off).
// One 256-thread workgroup per (head, step, 64 queries, batch part): thread (q, 16-B chunk) streams its chunk of every sample of the
// part (1-KiB rows of 64 B are contiguous: 4 KiB per sample and workgroup, fully coalesced), the [32 keys][64 q] partial tile is
// turned through LDS and added with one fp32 atomic per lane, 256 contiguous bytes per wave-instruction.
#define DBR_PARTS 4
__global__ __launch_bounds__(256)
void attn_dbias_reduce_kernel(const bf16* __restrict__ ds, float* __restrict__ slab, int B, int H, int N, int NP) {
    __shared__ float tile[32][65];
    const int nt = (N + 15) >> 4, nsteps = (nt + 1) >> 1;
    const int qb = blockIdx.x & 3, hi = blockIdx.x >> 2, i = hi % nsteps, h = hi / nsteps;
    const int tid = threadIdx.x, ql = tid >> 2, c = tid & 3, q = qb * 64 + ql;
    const int per = (B + DBR_PARTS - 1) / DBR_PARTS, b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (q < N) {
        const char* p = (const char*)ds + ((size_t)h * nsteps + i) * FB_SB + q * 64 + c * 16;
        const size_t bstride = (size_t)H * nsteps * FB_SB;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) {
            const bf16x8 v = *(const bf16x8*)(p + b * bstride);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f(v[j]);
        }
    }
    const int f = (((q >> 2) & 1) << 2) | (((q >> 3) & 1) << 1) | ((q >> 1) & 1);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int slot = (2 * c + e) ^ f;
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[4 * slot + j][ql] = acc[4 * e + j];
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6, qo = qb * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int key = 32 * i + w * 8 + k;
        if (key < N && qo < N) atomicAdd(slab + ((size_t)h * NP + key) * NP + qo, tile[w * 8 + k][lane]);
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
#define FWD_WAVES 7
static std::once_flag g_attn_once;
static int g_attn_ncu = 256;
static void attn_init_impl() {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        g_attn_ncu = prop.multiProcessorCount;
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<FWD_WAVES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<FWD_WAVES, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG_BYTES);
    (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<true, NT_MAX>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    (void)hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<false, NT_MAX>, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
}
static void attn_init_once() { std::call_once(g_attn_once, attn_init_impl); }

static int attn_check(int B, int H, int N, int head_dim) {
    if (head_dim != HD || B <= 0 || H <= 0 || N <= 0 || N > NT_MAX * 16) return UVIT_ERR_SHAPE;
    return UVIT_OK;
}

int uvit_attn_fwd_launch(const void* qkv, const float* biasP, void* out, float* lse, int B, int H, int N, int NP,
                         float scale, float p_drop, uint32_t seed, uint32_t layer, hipStream_t s, const int* bmap) {
    int rc = attn_check(B, H, N, HD); if (rc) return rc;
    attn_init_once();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    if (biasP) hipLaunchKernelGGL((attn_fwd_kernel<FWD_WAVES, true>), dim3(B * H), dim3(FWD_WAVES * 64), 2 * IMG_BYTES, s, (const bf16*)qkv,
                                  biasP, (bf16*)out, lse, H, N, NP, scale, thr, inv_keep, uvit_layer_key(seed, layer), g_attn_ncu, bmap);
    else hipLaunchKernelGGL((attn_fwd_kernel<FWD_WAVES, false>), dim3(B * H), dim3(FWD_WAVES * 64), 2 * IMG_BYTES, s, (const bf16*)qkv,
                            biasP, (bf16*)out, lse, H, N, NP, scale, thr, inv_keep, uvit_layer_key(seed, layer), g_attn_ncu, bmap);
    return uvit_check_launch();
}

size_t uvit_attn_bwd_fused_ws_bytes(int B, int H, int N) {
    const int nt = (N + 15) / 16, nsteps = (nt + 1) / 2;
    return (size_t)B * H * nsteps * FB_SB;
}

// ds_ws: bf16 workspace of uvit_attn_bwd_fused_ws_bytes(B, H, N) bytes, written when want_ds != 0 (the bias gradient needs it)
int uvit_attn_bwd_fused_launch(const void* qkv, const void* o_fwd, const void* d_o, const float* biasP, const float* lse,
                               float* delta, void* dqkv, void* ds_ws, int want_ds, int B, int H, int N, int NP, float scale,
                               float p_drop, uint32_t seed, uint32_t layer, hipStream_t s, const int* bmap) {
    int rc = attn_check(B, H, N, HD); if (rc) return rc;
    if (NP < NT_MAX * 16 || (want_ds && !ds_ws)) return UVIT_ERR_ARG;
    attn_init_once();
    const uint32_t thr = p_drop > 0.f ? uvit_drop_threshold16(p_drop) : 0u;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint32_t key = uvit_layer_key(seed, layer);
    bf16* dsw = want_ds ? (bf16*)ds_ws : nullptr;
#define FB_ARGS dim3(B * H), dim3(FB_WAVES * 64), FB_LDS, s, (const bf16*)qkv, (const bf16*)o_fwd, (const bf16*)d_o, biasP, lse, delta, \
        (bf16*)dqkv, dsw, H, N, NP, scale, thr, inv_keep, key, bmap
    if ((N + 15) / 16 == NT_MAX) {
        if (biasP) hipLaunchKernelGGL((attn_bwd_fused_kernel<true, NT_MAX>), FB_ARGS); else hipLaunchKernelGGL((attn_bwd_fused_kernel<false, NT_MAX>), FB_ARGS);
    } else {
        if (biasP) hipLaunchKernelGGL((attn_bwd_fused_kernel<true, 0>), FB_ARGS); else hipLaunchKernelGGL((attn_bwd_fused_kernel<false, 0>), FB_ARGS);
    }
#undef FB_ARGS
    return uvit_check_launch();
}

// dbias_slab = ONE [H][NP][NP] slab laid out [h][key][q]; accumulate = 0 overwrites it (zero fill first), 1 adds
int uvit_attn_dbias_reduce_launch(const void* ds_ws, float* dbias_slab, int accumulate, int B, int H, int N, int NP, hipStream_t s) {
    int rc = attn_check(B, H, N, HD); if (rc) return rc;
    if (!ds_ws || !dbias_slab || NP < NT_MAX * 16) return UVIT_ERR_ARG;
    if (!accumulate) { rc = uvit_zero_launch(dbias_slab, (size_t)H * NP * NP * sizeof(float), s); if (rc) return rc; }
    const int nt = (N + 15) / 16, nsteps = (nt + 1) / 2;
    hipLaunchKernelGGL(attn_dbias_reduce_kernel, dim3(H * nsteps * 4, DBR_PARTS), dim3(256), 0, s, (const bf16*)ds_ws, dbias_slab, B, H, N, NP);
    return uvit_check_launch();
}

