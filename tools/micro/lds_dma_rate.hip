// Diagnostic micro-benchmark (not part of libuvit.so): how fast can ONE CU take in LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per
// wave-instruction) when nothing else runs -- the ceiling the GEMM K loop's operand stream lives under.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_rate.hip -o tools/micro/lds_dma_rate.out
// Every workgroup streams `steps` x 64 KiB "K-tiles" (the 256x256x64 GEMM's operand bytes per K-tile: 64 pieces of 8 rows x 128 B) from a
// source of `footprint` bytes (small: L2-resident; large: HBM) into its LDS, 8 pieces in flight per wave (counted vmcnt), no compute.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const char* __restrict__ src, size_t footprint, int steps, size_t row_stride,
                                                         unsigned long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 128 KiB ring: two 64 KiB K-tiles
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECES = 64 / WAVES;                                // pieces per wave per K-tile
    // piece p of K-tile t: 8 rows x 128 B; rows of one operand panel are row_stride apart (a GEMM's lda * 2 bytes)
    const size_t lane_off = (size_t)(lane >> 3) * row_stride + (size_t)((lane & 7) ^ (lane >> 3)) * 16;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = 0; t < steps; ++t) {
        char* dst = smem + (t & 1) * 65536;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int p = j * WAVES + wave;
            // K-tile t of workgroup b: a 128-B column block of 512 rows; successive K-tiles walk along the rows (+128 B)
            size_t off = ((size_t)blockIdx.x * 512 + (size_t)p * 8) * row_stride + (size_t)t * 128 + lane_off;
            off %= footprint;
            off &= ~(size_t)15;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                             (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
        }
        if (PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same stream by BUFFER addressing: buffer_load_dwordx4 ... lds (one 32-bit offset per lane against a wave-uniform descriptor)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bdma_kernel(const char* __restrict__ src, size_t footprint, int steps, size_t row_stride,
                                                          unsigned long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECES = 64 / WAVES;
    const uint32_t lane_off = (uint32_t)((lane >> 3) * row_stride + ((lane & 7) ^ (lane >> 3)) * 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(footprint > 0x7fffffff ? 0x7fffffff : footprint), 0x00020000);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = 0; t < steps; ++t) {
        char* dst = smem + (t & 1) * 65536;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int p = j * WAVES + wave;
            size_t off = ((size_t)blockIdx.x * 512 + (size_t)p * 8) * row_stride + (size_t)t * 128;
            off %= footprint - 65536;
            off &= ~(size_t)15;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, (int)lane_off,
                                                 (int)__builtin_amdgcn_readfirstlane((uint32_t)off), 0, 0);
        }
        if (PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WAVES>
static int run_b(const char* name, const char* src, size_t footprint, size_t row_stride, unsigned long long* cyc, int grid) {
    const int steps = 400;
    CHECK(hipFuncSetAttribute((const void*)bdma_kernel<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(bdma_kernel<WAVES>, dim3(grid), dim3(WAVES * 64), 131072, 0, src, footprint, steps, row_stride, cyc);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(bdma_kernel<WAVES>, dim3(grid), dim3(WAVES * 64), 131072, 0, src, footprint, steps, row_stride, cyc);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[grid / 2];
    const double bytes = (double)grid * steps * 65536.0;
    printf("%-54s %d waves: %8.1f us  %6.2f TB/s chip | median CU: %7.0f cycles per 64-KiB K-tile = %5.1f cycles per 1-KiB piece = %5.1f B/clk\n",
           name, WAVES, ms * 1e3, bytes / (ms * 1e-3) / 1e12, med / steps, med / steps / 64.0, 65536.0 * steps / med);
    return 0;
}

// the same stream through registers: global_load_dwordx4 (16 B per lane), optionally written on to LDS with ds_write_b128
template <int WAVES, bool TO_LDS>
__global__ __launch_bounds__(WAVES * 64) void reg_kernel(const char* __restrict__ src, size_t footprint, int steps, size_t row_stride,
                                                         unsigned long long* __restrict__ cyc, uint4* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECES = 64 / WAVES;
    const size_t lane_off = (size_t)(lane >> 3) * row_stride + (size_t)((lane & 7) ^ (lane >> 3)) * 16;
    unsigned long long t0, t1;
    uint4 acc = make_uint4(0, 0, 0, 0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int t = 0; t < steps; ++t) {
        char* dst = smem + (t & 1) * 65536;
        uint4 v[PIECES];
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int p = j * WAVES + wave;
            size_t off = ((size_t)blockIdx.x * 512 + (size_t)p * 8) * row_stride + (size_t)t * 128 + lane_off;
            off %= footprint;
            off &= ~(size_t)15;
            v[j] = *(const uint4*)(src + off);
        }
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            if (TO_LDS) *(uint4*)(dst + (j * WAVES + wave) * 1024 + lane * 16) = v[j];
            else { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        }
        if (TO_LDS) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (lane == 0 && wave == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345 && sink) sink[threadIdx.x] = acc;
}

template <int WAVES, bool TO_LDS>
static int run_reg(const char* name, const char* src, size_t footprint, size_t row_stride, unsigned long long* cyc, int grid) {
    const int steps = 400;
    CHECK(hipFuncSetAttribute((const void*)reg_kernel<WAVES, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((reg_kernel<WAVES, TO_LDS>), dim3(grid), dim3(WAVES * 64), 131072, 0, src, footprint, steps, row_stride, cyc, (uint4*)nullptr);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((reg_kernel<WAVES, TO_LDS>), dim3(grid), dim3(WAVES * 64), 131072, 0, src, footprint, steps, row_stride, cyc, (uint4*)nullptr);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[grid / 2];
    const double bytes = (double)grid * steps * 65536.0;
    printf("%-54s %d waves: %8.1f us  %6.2f TB/s chip | median CU: %7.0f cycles per 64-KiB K-tile = %5.1f cycles per 1-KiB piece = %5.1f B/clk\n",
           name, WAVES, ms * 1e3, bytes / (ms * 1e-3) / 1e12, med / steps, med / steps / 64.0, 65536.0 * steps / med);
    return 0;
}

template <int WAVES>
static int run(const char* name, const char* src, size_t footprint, size_t row_stride, unsigned long long* cyc, int grid) {
    const int steps = 400;
    CHECK(hipFuncSetAttribute((const void*)dma_kernel<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(dma_kernel<WAVES>, dim3(grid), dim3(WAVES * 64), 131072, 0, src, footprint, steps, row_stride, cyc);
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(dma_kernel<WAVES>, dim3(grid), dim3(WAVES * 64), 131072, 0, src, footprint, steps, row_stride, cyc);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(grid);
    CHECK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[grid / 2];
    const double bytes = (double)grid * steps * 65536.0;
    printf("%-54s %d waves: %8.1f us  %6.2f TB/s chip | median CU: %7.0f cycles per 64-KiB K-tile = %5.1f cycles per 1-KiB piece = %5.1f B/clk\n",
           name, WAVES, ms * 1e3, bytes / (ms * 1e-3) / 1e12, med / steps, med / steps / 64.0, 65536.0 * steps / med);
    return 0;
}

int main() {
    const size_t big = (size_t)4 << 30;
    char* src; unsigned long long* cyc;
    CHECK(hipMalloc(&src, big)); CHECK(hipMemset(src, 1, big));
    CHECK(hipMalloc(&cyc, 1024 * 8));
    // row_stride 1536 = a K = 768 bf16 operand; footprints: 48 MiB (one fc1 A panel set: L2 + Infinity Cache), 4 GiB (HBM)
    if (run<8>("K=768 rows, 48 MiB footprint (L2 / Infinity Cache)", src, (size_t)48 << 20, 1536, cyc, 256)) return 1;
    if (run<4>("K=768 rows, 48 MiB footprint (L2 / Infinity Cache)", src, (size_t)48 << 20, 1536, cyc, 256)) return 1;
    if (run<8>("K=768 rows, 4 MiB footprint (L2 only)", src, (size_t)4 << 20, 1536, cyc, 256)) return 1;
    if (run<8>("K=3072 rows, 192 MiB footprint", src, (size_t)192 << 20, 6144, cyc, 256)) return 1;
    if (run<8>("K=768 rows, 4 GiB footprint (HBM)", src, big, 1536, cyc, 256)) return 1;
    if (run<8>("one workgroup alone, 4 MiB footprint", src, (size_t)4 << 20, 1536, cyc, 1)) return 1;
    if (run_b<8>("BUFFER_load ... lds, 48 MiB footprint", src, (size_t)48 << 20, 1536, cyc, 256)) return 1;
    if (run_b<4>("BUFFER_load ... lds, 48 MiB footprint", src, (size_t)48 << 20, 1536, cyc, 256)) return 1;
    if (run_b<8>("BUFFER_load ... lds, one workgroup alone, 4 MiB", src, (size_t)4 << 20, 1536, cyc, 1)) return 1;
    if (run_reg<8, false>("REGISTER loads only, 48 MiB footprint", src, (size_t)48 << 20, 1536, cyc, 256)) return 1;
    if (run_reg<8, true>("REGISTER loads + ds_write_b128, 48 MiB footprint", src, (size_t)48 << 20, 1536, cyc, 256)) return 1;
    if (run_reg<8, false>("REGISTER loads only, one workgroup alone, 4 MiB", src, (size_t)4 << 20, 1536, cyc, 1)) return 1;
    return 0;
}
