// CU hog for tools/bench_gemm_cu_loss.py: `blocks` workgroups that each take a whole CU (160 KiB of LDS) and spin for `micros` microseconds --
// a stand-in for RCCL's channel workgroups, which hold CUs for the length of a collective while the step's kernels are dispatched.
// Not part of libuvit.so.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/cu_hog.hip -o uncertainty-vit_amd/libcuhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void cu_hog_kernel(int micros, unsigned* sink) {
    extern __shared__ char lds[];
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
    uint64_t t = t0;
    unsigned acc = 0;
    while (t - t0 < (uint64_t)micros * 100ull) {
        __builtin_amdgcn_s_sleep(64);
        acc += lds[(acc * 64 + threadIdx.x) & 1023];
        t = __builtin_amdgcn_s_memrealtime();
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
extern "C" int cu_hog_launch(int blocks, int micros, unsigned* sink, void* stream) {
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(64), 160 * 1024, (hipStream_t)stream, micros, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
