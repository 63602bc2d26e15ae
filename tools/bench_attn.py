"""Micro-benchmark of the attention kernels through the C ABI (run on the GPU box)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncertainty_vit_amd import native  # noqa: E402

L = native.lib()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


if __name__ == "__main__":
    B, H, N, NP = 128, 12, 197, 208
    Cd = H * 64
    qkv = torch.randn(B * N, 3 * Cd, device="cuda").to(torch.bfloat16)
    biasP = torch.zeros(H, NP, NP, device="cuda"); biasP[:, :, N:] = -1e30
    out = torch.zeros(B * N, Cd, device="cuda", dtype=torch.bfloat16)
    d_o = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16)
    lse = torch.zeros(B, H, N, device="cuda"); delta = torch.zeros_like(lse)
    dqkv = torch.zeros_like(qkv)
    flops_fwd = 4.0 * B * H * N * N * 64
    for p in (0.0, 0.05):
        f = lambda: L.uvit_op_attn_fwd(P(qkv), P(biasP), P(out), P(lse), B, H, N, NP, 0.125, p, 1, 0, S())
        us = timeit(f)
        print(f"fwd  p={p}: {us:7.1f} us  {flops_fwd / us / 1e6:6.1f} TF/s")
        ws = torch.empty(L.uvit_op_attn_bwd_ws_bytes(B, H, N), dtype=torch.uint8, device="cuda")
        slab1 = torch.zeros(H, NP, NP, device="cuda")
        for with_dbias in (True, False):
            g = lambda: L.uvit_op_attn_bwd(P(qkv), P(out), P(d_o), P(biasP), P(lse), P(delta), P(dqkv), P(slab1 if with_dbias else None), 1,
                                                 P(ws), B, H, N, NP, 0.125, p, 1, 0, S())
            us = timeit(g)
            print(f"bwd  p={p} fused{' + dbias reduce' if with_dbias else '               '}: {us:7.1f} us  {2.5 * flops_fwd / us / 1e6:6.1f} TF/s (5 products)")

    # ---- two-stream (Wasserstein) attention: forward, fused backward (+ bias-gradient reduction)
    qkv_m = torch.randn(B * N, 3 * Cd, device="cuda").to(torch.bfloat16)
    qkv_c = (torch.nn.functional.elu(torch.randn(B * N, 3 * Cd, device="cuda")) + 1).to(torch.bfloat16)
    out_m = torch.zeros(B * N, Cd, device="cuda", dtype=torch.bfloat16); out_c = torch.zeros_like(out_m)
    d_m = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16); d_c = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16)
    dq_m = torch.zeros_like(qkv_m); dq_c = torch.zeros_like(qkv_m)
    ws2 = torch.empty(L.uvit_op_attn2_bwd_ws_bytes(B, H, N), dtype=torch.uint8, device="cuda")
    slab2 = torch.zeros(H, NP, NP, device="cuda")
    for p in (0.0, 0.05):
        f = lambda: L.uvit_op_attn2_fwd(P(qkv_m), P(qkv_c), P(biasP), P(out_m), P(out_c), P(lse), B, H, N, NP, 0.125, p, 1, 0, S())
        us = timeit(f)
        print(f"two-stream fwd  p={p}: {us:7.1f} us  {2.0 * flops_fwd / us / 1e6:6.1f} TF/s (4 products)")
        for with_dbias in (True, False):
            g = lambda: L.uvit_op_attn2_bwd(P(qkv_m), P(qkv_c), P(out_m), P(out_c), P(d_m), P(d_c), P(biasP), P(lse), P(delta), P(dq_m), P(dq_c),
                                            P(slab2 if with_dbias else None), 1, P(ws2), B, H, N, NP, 0.125, p, 1, 0, S())
            us = timeit(g)
            print(f"two-stream bwd  p={p} fused{' + dbias reduce' if with_dbias else '               '}: {us:7.1f} us  {5.0 * flops_fwd / us / 1e6:6.1f} TF/s (10 products)")
