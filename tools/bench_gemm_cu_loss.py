"""The persistent NT GEMM (fc1 shape) with some CUs held by another kernel, fixed-stride tile assignment against the dynamic one (round 4).
A hog kernel (tools/micro/cu_hog.hip: one 160-KiB-LDS workgroup per CU, spinning) is started on a second stream, then the GEMM is timed with
HIP events on its own stream.  Build the hog first:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/cu_hog.hip -o uncertainty-vit_amd/libcuhog.so"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import P, S, L  # noqa: E402
from uncertainty_vit_amd.native import GemmEpilogue, Tuning  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOG = C.CDLL(os.path.join(ROOT, "uncertainty-vit_amd", "libcuhog.so"))
M, N, K = 25216, 3072, 768
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
cnt = torch.zeros(16, dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
ep = GemmEpilogue(); ep.out = out.data_ptr(); ep.bias = bias.data_ptr(); ep.ldo = N; ep.tokens = 1; ep.patches = 1
tu = Tuning.default(nt_variant=1, nt_persist=1)
side = torch.cuda.Stream()


def run(dynamic):
    return L.uvit_op_gemm_nt_sched(2, P(a), P(w), M, N, K, K, K, C.byref(ep), C.byref(tu), P(cnt) if dynamic else None, S())


def timed_pair(hog_blocks, iters=12):
    """median us of the fixed-stride and of the dynamic launch, ALTERNATING (the chip's clock drifts with what ran before)"""
    ts = {False: [], True: []}
    for it in range(2 * iters):
        dynamic = bool(it & 1)
        torch.cuda.synchronize()
        if hog_blocks:
            HOG.cu_hog_launch(hog_blocks, 2000, P(sink), C.c_void_p(side.cuda_stream))     # 2 ms: longer than the GEMM
            torch.cuda._sleep(200000)                                                        # let the hog get onto its CUs first
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); assert run(dynamic) == 0; e1.record()
        torch.cuda.synchronize()
        ts[dynamic].append(e0.elapsed_time(e1) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    return med(ts[False]), med(ts[True])


if __name__ == "__main__":
    for _ in range(20):            # warm the clocks: the first timed launches after an idle GPU read 10-20 % slow
        run(False); run(True)
    print(f"fc1-shaped GEMM (M={M} N={N} K={K}, bias+GELU, persistent 256x256 kernel, 1182 tiles on 256 workgroups), median us of 12, fixed / dynamic alternating:")
    for hog in (0, 8, 16, 32, 0):
        f, d = timed_pair(hog)
        print(f"  CUs held by the hog: {hog:3d}   fixed stride {f:7.1f} us   dynamic {d:7.1f} us   ({d / f:.2f}x)")
