"""Epilogue-only timing of the NT GEMM kernels: K = 64 (one K-tile) makes the launch almost pure epilogue, so the bytes
each fused epilogue moves can be priced against HBM directly (run on the GPU box)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import L, P, S, TUNE, _timed  # noqa: E402
from uncertainty_vit_amd.native import GemmEpilogue  # noqa: E402

M = 25216
names = {0: "bf16", 1: "qkv", 2: "gelu", 3: "resid", 4: "f32", 8: "gelu_dg", 9: "mulaux"}


def run(mode, N, K, variant, out2=True):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias, gamma = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda") if mode == 3 else None
    aux = torch.randn(M, N, device="cuda").to(torch.bfloat16) if mode == 9 else None
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if mode in (3, 4) else torch.bfloat16)
    o2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if (mode in (2, 3, 8) and out2) else None
    e = GemmEpilogue()
    e.out, e.out2 = out.data_ptr(), (o2.data_ptr() if o2 is not None else 0)
    e.bias = e.bias2 = bias.data_ptr(); e.gamma = gamma.data_ptr()
    e.resid = resid.data_ptr() if resid is not None else 0
    e.aux = aux.data_ptr() if aux is not None else 0
    e.ldo, e.tokens, e.patches = N, 197, 196
    TUNE.nt_variant = variant
    fn = lambda: L.uvit_op_gemm_nt_tuned(mode, P(a), P(w), M, N, K, K, K, C.byref(e), C.byref(TUNE), None, S())  # noqa: E731
    for _ in range(3):
        assert fn() == 0
    us = _timed(fn, 20)
    rd = M * K * 2 + N * K * 2 + (M * N * 4 if mode == 3 else 0) + (M * N * 2 if mode == 9 else 0)
    wr = M * N * (4 if mode in (3, 4) else 2) + (M * N * 2 if o2 is not None else 0)
    return us, (rd + wr) / us / 1e6


for mode, N, o2 in ((0, 768, True), (0, 3072, True), (4, 768, True), (3, 768, True), (3, 768, False), (8, 3072, True), (2, 3072, False), (9, 3072, True), (1, 2304, True)):
    row = []
    for K in (64, 768):
        for v in (1, 5, 7):
            us, tbs = run(mode, N, K, v, o2)
            row.append(f"K{K} v{v}: {us:6.1f}us {tbs:4.2f}TB/s")
    print(f"{names[mode]:7s} N={N:4d} out2={int(o2)}: " + " | ".join(row))
