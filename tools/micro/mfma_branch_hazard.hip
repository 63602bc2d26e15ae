// Test case of tools/check_mfma_hazard.py: an MFMA, a short wave-uniform `if`, then a VALU read of the MFMA's result.
// hipcc pads the MFMA -> VALU wait states along the fall-through path (through the `if` body); the guard reports the taken path
// when the padding it finds there is too short.  Not part of libuvit.so.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void mfma_branch_hazard(const bf16x8* a, const bf16x8* b, float* out, const float* aux, int flag) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bf16x8 av = a[threadIdx.x], bv = b[threadIdx.x];
    float extra = aux[threadIdx.x];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
    if (flag) {                      // wave-uniform (kernel argument): s_cbranch_scc
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %0, 0.5" : "+v"(extra));     // opaque: not folded into a select
    }
    out[threadIdx.x] = acc[0] * extra + acc[1] + acc[2] + acc[3];
}
