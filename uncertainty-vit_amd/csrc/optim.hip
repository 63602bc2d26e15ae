// Flat-arena optimiser kernels (gfx950), pure HBM streams with float4 accesses:
//   EMA teacher update        engine_for_cyclical.py:182-185  (e <- d*e + (1-d)*m), + bf16 shadow
//   global grad-norm + clip   utils.py:375-376 (clip_grad_norm_)
//   AdamW                     optim_factory.py:133-134 (torch.optim.AdamW), + bf16 shadow of the weights
// The parameter arena is laid out [decay tensors | no-decay tensors] so weight decay is a single
// index comparison (optim_factory.py:58-97).
#include "common.h"
#include "uvit_internal.h"

// A step whose loss or gradient norm is not finite must leave the weights as they were: the reference stops BEFORE
// backward / optimizer.step / EMA (engine_for_cyclical.py:166-168).  Both kernels test the device-side values, so the
// decision needs no host round trip and is identical on every rank (the norm is taken after the all-reduce).
__device__ __forceinline__ bool not_finite(float x) { return (__float_as_uint(x) & 0x7F800000u) == 0x7F800000u; }
__device__ __forceinline__ bool step_poisoned(const float* loss, const double* sumsq) {
    return (loss && not_finite(*loss)) || (sumsq && not_finite((float)*sumsq));
}

__global__ __launch_bounds__(256)
void ema_kernel(float* __restrict__ e, const float* __restrict__ p, bf16* __restrict__ eb, size_t n4, float d,
                const float* __restrict__ guard_loss, const double* __restrict__ guard_sumsq) {
    if (step_poisoned(guard_loss, guard_sumsq)) return;
    const float om = 1.0f - d;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = ((const float4*)e)[i];
        const float4 b = ((const float4*)p)[i];
        a.x = d * a.x + om * b.x; a.y = d * a.y + om * b.y; a.z = d * a.z + om * b.z; a.w = d * a.w + om * b.w;
        ((float4*)e)[i] = a;
        if (eb) { bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)}; ((bf16x4*)eb)[i] = o; }
    }
}

__global__ __launch_bounds__(256)
void sumsq_kernel(const float* __restrict__ g, size_t n4, double* __restrict__ out) {
    __shared__ double red[4];
    float part = 0.f;
    double acc = 0.0;
    int cnt = 0;
    // four independent 16-B loads in flight per lane (end of round 4: one load per iteration read 345 MB in 77 us = 4.5 TB/s)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = ((const float4*)g)[i], b = ((const float4*)g)[i + stride], c = ((const float4*)g)[i + 2 * stride], d = ((const float4*)g)[i + 3 * stride];
        part += (a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w) +
                (c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) + (d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
        if (++cnt == 16) { acc += part; part = 0.f; cnt = 0; }   // bounded fp32 run lengths
    }
    for (; i < n4; i += stride) {
        const float4 a = ((const float4*)g)[i];
        part += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    acc += part;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256)
void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                  bf16* __restrict__ pb, size_t n4, size_t n4_decay, float lr, float wd, float b1, float b2, float eps,
                  float bc1, float bc2_sqrt, const double* __restrict__ sumsq, float max_norm, float grad_scale,
                  float* __restrict__ gnorm_out, const float* __restrict__ guard_loss,
                  float* __restrict__ ema, bf16* __restrict__ ema_b, float ema_d, int* __restrict__ sticky,
                  const float* __restrict__ sched, int sched_len, int sched_idx) {
    if (sched) {           // device-resident schedules: {lr, weight_decay, ema_decay (< 0: skip)} of this iteration
        lr = sched[sched_idx]; wd = sched[sched_len + sched_idx]; ema_d = sched[2 * sched_len + sched_idx];
        if (ema_d < 0.f) ema = nullptr;
    }
    // ema != nullptr: the EMA teacher update e <- d e + (1 - d) p_new (engine_for_cyclical.py:182-185) rides in the same pass:
    // the new weights are in registers, so the separate EMA kernel's second read of the 345 MB parameter arena disappears
    float coef = grad_scale;
    if (sumsq) {
        const float norm = (float)sqrt(*sumsq) * grad_scale;
        if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) *gnorm_out = norm;
        if (max_norm > 0.f) coef *= fminf(max_norm / (norm + 1e-6f), 1.0f);
    }
    // sticky: the host learns of a poisoned step one step late (asynchronous metrics) and has the next step enqueued by
    // then -- that step, and any later one, must not move the weights either
    if (step_poisoned(guard_loss, sumsq) || (sticky && *sticky)) {
        if (sticky && blockIdx.x == 0 && threadIdx.x == 0) *sticky = 1;
        return;
    }
    const float step_size = lr / bc1;
    const float decay = 1.0f - lr * wd;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = ((const float4*)p)[i];
        const float4 gg = ((const float4*)g)[i];
        float4 mm = ((const float4*)m)[i], vv = ((const float4*)v)[i];
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ema) a = ((const float4*)ema)[i];         // requested with the other four operands, not behind the update's arithmetic
        if (i < n4_decay) { pp.x *= decay; pp.y *= decay; pp.z *= decay; pp.w *= decay; }
        const float g0 = gg.x * coef, g1 = gg.y * coef, g2 = gg.z * coef, g3 = gg.w * coef;
        mm.x = b1 * mm.x + (1.f - b1) * g0; mm.y = b1 * mm.y + (1.f - b1) * g1;
        mm.z = b1 * mm.z + (1.f - b1) * g2; mm.w = b1 * mm.w + (1.f - b1) * g3;
        vv.x = b2 * vv.x + (1.f - b2) * g0 * g0; vv.y = b2 * vv.y + (1.f - b2) * g1 * g1;
        vv.z = b2 * vv.z + (1.f - b2) * g2 * g2; vv.w = b2 * vv.w + (1.f - b2) * g3 * g3;
        pp.x -= step_size * mm.x / (sqrtf(vv.x) / bc2_sqrt + eps);
        pp.y -= step_size * mm.y / (sqrtf(vv.y) / bc2_sqrt + eps);
        pp.z -= step_size * mm.z / (sqrtf(vv.z) / bc2_sqrt + eps);
        pp.w -= step_size * mm.w / (sqrtf(vv.w) / bc2_sqrt + eps);
        ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv;
        if (pb) { bf16x4 o = {f2bf(pp.x), f2bf(pp.y), f2bf(pp.z), f2bf(pp.w)}; ((bf16x4*)pb)[i] = o; }
        if (ema) {
            const float om = 1.0f - ema_d;
            a.x = ema_d * a.x + om * pp.x; a.y = ema_d * a.y + om * pp.y; a.z = ema_d * a.z + om * pp.z; a.w = ema_d * a.w + om * pp.w;
            ((float4*)ema)[i] = a;
            if (ema_b) { bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)}; ((bf16x4*)ema_b)[i] = o; }
        }
    }
}

// zero fill with 16-B stores: hipMemsetAsync's fill kernel moved the step's 440 MB at 0.8 TB/s (0.53 ms per step)
__global__ __launch_bounds__(256)
void zero_kernel(float4* __restrict__ dst, size_t n4) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = z;
}

__global__ __launch_bounds__(256)
void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = ((const float4*)src)[i];
        bf16x4 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
        ((bf16x4*)dst)[i] = o;
    }
}

static inline int stream_grid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int uvit_ema_launch(float* ema, const float* p, void* ema_bf16, size_t n, float decay, hipStream_t s, const float* guard_loss,
                    const double* guard_sumsq) {
    if (n % 4) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(ema_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, s, ema, p, (bf16*)ema_bf16, n / 4, decay, guard_loss, guard_sumsq);
    return uvit_check_launch();
}
int uvit_sumsq_launch(const float* g, size_t n, double* out, hipStream_t s) {
    if (n % 4) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(sumsq_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, s, g, n / 4, out);
    return uvit_check_launch();
}
int uvit_adamw_launch(float* p, const float* g, float* m, float* v, void* p_bf16, size_t n, size_t n_decay, float lr,
                      float wd, float b1, float b2, float eps, int step, const double* sumsq, float max_norm,
                      float grad_scale, float* gnorm_out, hipStream_t s, const float* guard_loss, float* ema, void* ema_bf16,
                      float ema_decay, int* sticky, const float* sched, int sched_len, int sched_idx) {
    if (n % 4 || n_decay % 4 || step < 1) return UVIT_ERR_SHAPE;
    const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, s, p, g, m, v, (bf16*)p_bf16, n / 4, n_decay / 4,
                       lr, wd, b1, b2, eps, bc1, bc2_sqrt, sumsq, max_norm, grad_scale, gnorm_out, guard_loss, ema, (bf16*)ema_bf16, ema_decay, sticky, sched, sched_len, sched_idx);
    return uvit_check_launch();
}
int uvit_cast_bf16_launch(const float* src, void* dst, size_t n, hipStream_t s) {
    if (n % 4) return UVIT_ERR_SHAPE;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, s, src, (bf16*)dst, n / 4);
    return uvit_check_launch();
}
int uvit_zero_launch(void* dst, size_t bytes, hipStream_t s) {
    if (bytes == 0) return UVIT_OK;
    if (((uintptr_t)dst | bytes) & 15) return hipMemsetAsync(dst, 0, bytes, s) == hipSuccess ? UVIT_OK : UVIT_ERR_LAUNCH;
    hipLaunchKernelGGL(zero_kernel, dim3(stream_grid(bytes / 16)), dim3(256), 0, s, (float4*)dst, bytes / 16);
    return uvit_check_launch();
}
