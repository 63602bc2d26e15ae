// Diagnostic micro-benchmark (not part of libuvit.so): how fast can ONE workgroup per CU stream a 25216 x 3072 bf16 matrix out,
// tile by tile, for different shapes of a store instruction's footprint?  Mirrors the residency of gemm_nt256_kernel's epilogue
// (8 waves per CU, 256 x 256 tiles) without any arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o gpurun_out/store_pattern && gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// PAT 0: linear (wave-instruction = 1 KiB contiguous).  PAT 1: 8 rows x 128 B.  PAT 2: 4 rows x 256 B.  PAT 3: 2 rows x 512 B.
template <int PAT, int NW>
__global__ __launch_bounds__(NW * 64) void store_kernel(uint4* __restrict__ out, int M, int N, int tiles_n, int ntiles, int persistent) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (int tile = blockIdx.x; tile < ntiles; tile += persistent ? gridDim.x : ntiles) {
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * 256, n0 = tn * 256;          // element coordinates; a row of the matrix is N * 2 bytes
        constexpr int INSTR = 256 * 256 * 2 / 1024 / NW;  // 1-KiB store instructions per wave and tile
        if (PAT == 0) {
            // the tile's 128 KiB as one contiguous range (what a fill kernel does)
            uint4* base = out + ((size_t)tile * 256 * 256 * 2) / 16;
#pragma unroll 4
            for (int i = 0; i < INSTR; ++i) base[(size_t)(wave * INSTR + i) * 64 + lane] = v;
        } else {
            constexpr int ROW_B = PAT == 1 ? 128 : PAT == 2 ? 256 : 512;     // bytes of one row segment per instruction
            constexpr int ROWS_I = 1024 / ROW_B;                             // rows per instruction
            constexpr int WN = 512 / ROW_B;                                  // waves side by side
            const int wn = wave % WN, wm = wave / WN;
            constexpr int ROWS_W = 256 / (NW / WN);                          // rows per wave
            const int r_in = lane / (ROW_B / 16), c16 = lane % (ROW_B / 16);
            // a wave covers its ROWS_W x (ROW_B bytes) sub-tile in column halves when the tile is wider than WN * ROW_B
            constexpr int CH = 512 / (WN * ROW_B);
#pragma unroll
            for (int ch = 0; ch < CH; ++ch)
#pragma unroll 4
                for (int i = 0; i < ROWS_W / ROWS_I; ++i) {
                    const int m = m0 + wm * ROWS_W + i * ROWS_I + r_in;
                    const size_t byte = (size_t)m * N * 2 + (size_t)n0 * 2 + (size_t)(ch * WN + wn) * ROW_B + c16 * 16;
                    if (m < M) out[byte / 16] = v;
                }
        }
    }
}

template <int PAT, int NW>
static int run(const char* name, uint4* out, int M, int N, size_t lds, int persistent) {
    const int tiles_n = N / 256, tiles_m = (M + 255) / 256, ntiles = tiles_m * tiles_n;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipFuncSetAttribute((const void*)store_kernel<PAT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = persistent ? 256 : ntiles;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_kernel<PAT, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, M, N, tiles_n, ntiles, persistent);
    CHECK(hipEventRecord(a));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((store_kernel<PAT, NW>), dim3(grid), dim3(NW * 64), lds, 0, out, M, N, tiles_n, ntiles, persistent);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters, mb = (double)M * N * 2 / 1e6;
    printf("%-44s lds %6zu %s: %7.1f us  %5.2f TB/s\n", name, lds, persistent ? "persistent" : "1 tile/WG ", us, mb / us);
    return 0;
}

int main() {
    const int M = 25216, N = 3072;
    uint4* out;
    CHECK(hipMalloc(&out, (size_t)(M + 256) * N * 2));
    for (int persistent = 0; persistent < 2; ++persistent)
        for (size_t lds : {(size_t)0, (size_t)100 * 1024}) {
            if (run<0, 8>("linear 1 KiB, 8 waves", out, M, N, lds, persistent)) return 1;
            if (run<1, 8>("8 rows x 128 B per instr, 8 waves", out, M, N, lds, persistent)) return 1;
            if (run<2, 8>("4 rows x 256 B per instr, 8 waves", out, M, N, lds, persistent)) return 1;
            if (run<3, 8>("2 rows x 512 B per instr, 8 waves", out, M, N, lds, persistent)) return 1;
            if (run<1, 4>("8 rows x 128 B per instr, 4 waves", out, M, N, lds, persistent)) return 1;
            if (run<3, 4>("2 rows x 512 B per instr, 4 waves", out, M, N, lds, persistent)) return 1;
        }
    return 0;
}
