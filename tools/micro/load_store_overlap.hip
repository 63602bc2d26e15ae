// Diagnostic micro-benchmark (not part of libuvit.so): do a CU's LDS-DMA operand loads (L2-resident panels, as in a GEMM K loop)
// make progress while the same CU streams a finished tile out?  One 512-thread workgroup per CU; waves 0-3 only load, waves 4-7
// only store (separate in-order vmcnt queues).  Also: the store rate of a CU when only a fraction of the CUs store.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/load_store_overlap.hip -o tools/micro/load_store_overlap.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// loads: each loader wave pulls `load_kb` KiB from a 2 MiB window of `src` (L2 hits after the first pass) into LDS, 8 KiB in flight
// stores: each storer wave writes `store_kb` KiB of the CU's private output range
__global__ __launch_bounds__(512) void overlap_kernel(const uint4* __restrict__ src, uint4* __restrict__ out, int load_kb, int store_kb,
                                                      int trickle_every) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        const uint4* base = src + ((size_t)(blockIdx.x % 64) * 2048 + wave * 512) * 64;      // 32 KiB per wave inside a 128-KiB slice
        char* lds = smem + wave * 8192;
        for (int i = 0; i < load_kb; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(i & 31) * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(lds + (i & 7) * 1024), 16, 0, 0);
            if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        uint4* base = out + ((size_t)blockIdx.x * 4 + (wave - 4)) * (size_t)store_kb * 64;
        const uint4 v = make_uint4(lane, wave, 1, 2);
        for (int i = 0; i < store_kb; ++i) {
            base[(size_t)i * 64 + lane] = v;
            for (int k = 0; k < trickle_every; ++k) __builtin_amdgcn_s_sleep(8);
        }
    }
}

static int run(const char* name, const uint4* src, uint4* out, int grid, int load_kb, int store_kb, int trickle) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipFuncSetAttribute((const void*)overlap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(overlap_kernel, dim3(grid), dim3(512), 100 * 1024, 0, src, out, load_kb, store_kb, trickle);
    CHECK(hipEventRecord(a));
    const int iters = 10;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(overlap_kernel, dim3(grid), dim3(512), 100 * 1024, 0, src, out, load_kb, store_kb, trickle);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    const double lb = (double)grid * 4 * load_kb * 1024, sb = (double)grid * 4 * store_kb * 1024;
    printf("%-52s grid %3d: %8.1f us   loads %6.2f TB/s (%5.1f B/clk/CU @2GHz)   stores %6.2f TB/s (%5.1f B/clk/CU)\n", name, grid, us,
           lb / us / 1e6, lb / grid / (us * 2000.0), sb / us / 1e6, sb / grid / (us * 2000.0));
    return 0;
}

int main() {
    uint4 *src, *out;
    CHECK(hipMalloc(&src, (size_t)64 * 2048 * 64 * 16 + (1 << 20)));
    CHECK(hipMemset(src, 1, (size_t)64 * 2048 * 64 * 16));
    const int SKB = 512;                                    // 512 KiB per storer wave = 2 MiB per CU and launch
    CHECK(hipMalloc(&out, (size_t)256 * 4 * SKB * 1024));
    for (int grid : {256, 128, 64, 32}) {
        if (run("stores only", src, out, grid, 0, SKB, 0)) return 1;
    }
    if (run("loads only (L2-resident panels)", src, out, 256, 2048, 0, 0)) return 1;
    if (run("loads only, 4x the bytes", src, out, 256, 8192, 0, 0)) return 1;
    if (run("loads + stores together", src, out, 256, 2048, SKB, 0)) return 1;
    if (run("loads 4x + stores together", src, out, 256, 8192, SKB, 0)) return 1;
    if (run("loads 4x + stores trickled (1 x s_sleep 8 per KiB)", src, out, 256, 8192, SKB, 1)) return 1;
    if (run("stores trickled alone (1 x s_sleep 8 per KiB)", src, out, 256, 0, SKB, 1)) return 1;
    if (run("loads 4x + stores trickled (4 x s_sleep 8 per KiB)", src, out, 256, 8192, SKB, 4)) return 1;
    if (run("stores trickled alone (4 x s_sleep 8 per KiB)", src, out, 256, 0, SKB, 4)) return 1;
    return 0;
}
