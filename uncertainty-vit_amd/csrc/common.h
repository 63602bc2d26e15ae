// Shared device helpers for the gfx950 (CDNA4) kernels of libuvit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UVIT_OK 0
#define UVIT_ERR_ARG (-1)
#define UVIT_ERR_SHAPE (-2)
#define UVIT_ERR_LAUNCH (-3)
#define UVIT_ERR_WORKSPACE (-4)

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

static inline int uvit_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? UVIT_OK : UVIT_ERR_LAUNCH;
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32 (RNE, NaN kept)

// ---- counter-based RNG shared with oracle/vit_oracle.py (_mix32 / attn_keep_mask) ----
__host__ __device__ __forceinline__ uint32_t uvit_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t uvit_layer_key(uint32_t seed, uint32_t layer) {
    return uvit_hash32(seed ^ ((layer + 1u) * 0x9E3779B9u));
}
__host__ __device__ __forceinline__ uint32_t uvit_drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
__host__ __device__ __forceinline__ uint32_t uvit_drop_threshold16(float p) {
    const double t = (double)p * 65536.0;
    return t >= 65535.0 ? 65535u : (uint32_t)t;
}
// element index idx = ((b*H+h)*N+i)*N+j (mod 2^32); kept when hash >= threshold
__device__ __forceinline__ bool uvit_keep(uint32_t key, uint32_t idx, uint32_t thr) {
    return uvit_hash32(idx ^ key) >= thr;
}

// ---- wave (64 lanes) reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_exact(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// XCD-aware bijective remap of a linear block id (guide T1): blocks that share an XCD get a
// contiguous chunk of the tile grid so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}
