"""wgrad (TN) GEMM: token-count sweep -> slope (main loop) and intercept (fill + atomics epilogue)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import time_tn, L, TUNE  # noqa: E402

for N, K in ((3072, 768), (768, 768), (2304, 768), (768, 3072)):
    for v in (0, 1):
        TUNE.tn_variant = v
        row = []
        for M in (6400, 12608, 25216, 50432):
            us, tf = time_tn(M, N, K)
            row.append(f"M={M}: {us:6.1f}us {tf:5.0f}TF")
        print(f"N={N:5d} K={K:5d} variant {v}: " + " | ".join(row))
TUNE.tn_variant = 3

# the four wgrads of one ViT-B layer: separate launches (best variant each) vs the grouped launch
import ctypes as C
import torch
from uncertainty_vit_amd.native import WgradProblem
from bench_gemm import P, S
M, Cd, Hd = 25216, 768, 3072
specs = [(M, Cd, Hd), (M, Hd, Cd), (M, Cd, Cd), (M, 3 * Cd, Cd)]
bufs = []
probs = (WgradProblem * 4)()
for i, (m, n, k) in enumerate(specs):
    y = (torch.randn(m, n, device="cuda") * 0.1).to(torch.bfloat16); x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    out = torch.zeros(n, k, device="cuda"); b = torch.zeros(n, device="cuda")
    bufs.append((y, x, out, b))
    q = probs[i]; q.Y, q.X, q.C = y.data_ptr(), x.data_ptr(), out.data_ptr()
    q.bias = b.data_ptr() if i in (1, 3) else None; q.bias2 = None; q.bias_end = n if i == 1 else Cd; q.bias2_begin = 2 * Cd
    q.M, q.N, q.K, q.ldy, q.ldx, q.ldc = m, n, k, n, k, k


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record(); torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


def separate():
    for (y, x, out, b), (m, n, k) in zip(bufs, specs):
        L.uvit_op_gemm_tn(P(y), P(x), m, n, k, n, k, P(out), k, C.byref(TUNE), S())


flops = sum(2.0 * m * n * k for m, n, k in specs)
for v in (0, 1):
    TUNE.tn_variant = v
    us = timed(separate)
    print(f"layer wgrads, 4 launches, variant {v}: {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s")
TUNE.tn_variant = 3
for ch in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14):
    TUNE.wgrad_group_chunks = ch
    us = timed(lambda: L.uvit_op_wgrad_group(probs, 4, C.byref(TUNE), S()))
    print(f"layer wgrads, grouped, chunks {ch:2d}: {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s")
TUNE.wgrad_group_chunks = 0
