// Diagnostic micro-benchmark (not part of libuvit.so): rate of LDS float atomics (ds_add_f32, no return) for the address
// patterns the fused attention backward would use to accumulate the relative-position-bias gradient into per-head bins
// (bin = u(q) - u(k) + c: lanes (g, li) of an MFMA accumulator register hit bin li - 4 g - r - 16 t).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic.hip -o tools/micro/lds_atomic.out
// pattern 0: natural (4 lanes per address along the diagonals), 1: one address per lane (conflict-free), 2: all lanes one address,
//         3: natural with the query tile strided by 13 tokens (u differs by 26 per lane), 4: natural, 32x32-MFMA register layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int PATTERN>
__global__ __launch_bounds__(832) void atomic_kernel(float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
    __shared__ float bins[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2048; i += blockDim.x) bins[i] = 0.f;
    __syncthreads();
    const int g = lane >> 4, li = lane & 15;
    const int uq = 16 * wave + li + 400;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 13; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int idx;
                if (PATTERN == 0) idx = uq - (16 * t + 4 * g + r);
                else if (PATTERN == 1) idx = lane + 64 * r + 256 * (t & 3);
                else if (PATTERN == 2) idx = 7 + t;
                else if (PATTERN == 3) idx = 26 * li + wave + 800 - (16 * t + 4 * g + r);
                else idx = 400 + (lane & 31) + 32 * wave - (16 * t + 4 * (lane >> 5) + r);
                idx &= 2047;
                __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&bins[idx], 1.0f + it, 0, 0, false);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    __syncthreads();
    if (tid < 2048 / 64) { float s = 0.f; for (int i = 0; i < 64; ++i) s += bins[tid * 64 + i]; out[blockIdx.x * 32 + tid] = s; }
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int P>
static int run(const char* name, float* out, unsigned long long* cyc, int waves) {
    const int iters = 50, grid = 256;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(atomic_kernel<P>, dim3(grid), dim3(64 * waves), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(atomic_kernel<P>, dim3(grid), dim3(64 * waves), 0, 0, out, cyc, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[256 * 16];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
    const double n_instr = (double)iters * 52;
    printf("%-58s %2d waves/CU: %8.1f us  slowest wave %9.0f cycles = %6.1f cycles per ds_add_f32 per wave, %5.1f LDS cycles per instruction per CU\n",
           name, waves, ms * 1e3, mx, mx / n_instr, mx / (n_instr * waves));
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 256 * 32 * 4)); CHECK(hipMalloc(&cyc, 256 * 16 * 8));
    for (int waves : {1, 4, 13}) {
        if (run<1>("one address per lane (conflict-free)", out, cyc, waves)) return 1;
        if (run<0>("natural: bin = li - 4g - r - 16t (4 lanes per address)", out, cyc, waves)) return 1;
        if (run<3>("query tile strided by 13 tokens", out, cyc, waves)) return 1;
        if (run<4>("natural, 32x32 accumulator layout (2 lanes per address)", out, cyc, waves)) return 1;
        if (run<2>("all 64 lanes one address", out, cyc, waves)) return 1;
    }
    return 0;
}
