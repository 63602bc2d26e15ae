"""Diagnostic A/B of stand-alone builds of csrc/gemm.hip (run on the GPU box): python tools/bench_gemm_variants.py lib1.so[:variant] ...
(variant = uvit_tuning.nt_variant, + 100 to switch the persistent form off; default 3 = auto)
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -DUVIT_SRC_HASH='"dbg"' -DGEMM_DEBUG [...] -shared \
       uncertainty-vit_amd/csrc/gemm.hip -o uncertainty-vit_amd/libgemm_dbg_X.so"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
SHAPES = [(8, 3072, 768, "fc1 GELU+GELU'"), (2, 3072, 768, "fc1 GELU"), (0, 2304, 768, "qkv bf16"), (0, 768, 3072, "fc1 dgrad bf16"),
          (0, 768, 2304, "qkv dgrad bf16"), (0, 768, 768, "proj dgrad bf16"), (3, 768, 768, "proj resid"), (3, 768, 3072, "fc2 resid")]


def timeit(fn, iters=20):
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


if __name__ == "__main__":
    M = 25216
    torch.manual_seed(0)
    ref = {}
    for arg in sys.argv[1:]:
        name, _, var = arg.partition(":")
        variant = int(var) if var else 3
        L = C.CDLL(os.path.join(ROOT, "uncertainty-vit_amd", name))
        line = []
        for mode, N, K, label in SHAPES:
            torch.manual_seed(1000 * mode + N + K)
            A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda")
            out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if mode == 3 else torch.bfloat16)
            out2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            resid = torch.randn(M, N, device="cuda") if mode == 3 else None
            gamma = torch.ones(N, device="cuda") if mode == 3 else None
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            run = lambda: L.uvit_debug_gemm_nt(mode, variant, P(A), P(W), M, N, K, P(out), P(out2), P(bias), P(resid), P(gamma), st)  # noqa: E731
            us = timeit(run)
            key = (mode, N, K)
            torch.cuda.synchronize()
            ok = ""
            if key in ref:
                d1, d2 = (out.float() - ref[key][0].float()).abs().max().item(), (out2.float() - ref[key][1].float()).abs().max().item()
                ok = "" if d1 == 0 and d2 == 0 else f" MISMATCH max|diff| {d1:.3g} {d2:.3g}"
            else:
                ref[key] = (out.clone(), out2.clone())
            line.append(f"{label} {us:6.1f}{ok}")
        print(f"{arg:28s} " + " | ".join(line), flush=True)
