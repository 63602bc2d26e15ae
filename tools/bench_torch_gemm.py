"""Vendor-library reference for the step's GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS): what is achievable, not a product path."""
import torch
M = 25216
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304)):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    us = t(lambda: torch.matmul(x, w.t()))
    print(f"NT  M={M} N={N:5d} K={K:5d}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
for N, K in ((3072, 768), (768, 768), (2304, 768), (768, 3072)):
    y = (torch.randn(M, N, device="cuda") * 0.1).bfloat16(); x = torch.randn(M, K, device="cuda").bfloat16()
    us = t(lambda: torch.matmul(y.t(), x))
    print(f"TN  M={M} N={N:5d} K={K:5d}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s")
for n in (4096, 8192):
    x = torch.randn(n, n, device="cuda").bfloat16(); w = torch.randn(n, n, device="cuda").bfloat16()
    us = t(lambda: torch.matmul(x, w.t()), 5)
    print(f"NT  {n}^3: {us:8.1f} us {2.0*n**3/us/1e6:7.1f} TF/s")
