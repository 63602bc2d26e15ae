"""Diagnostic A/B of stand-alone -DATTN_DEBUG builds of csrc/attention.hip (forward only): python tools/bench_attn_fwd_variants.py a.so b.so"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731

if __name__ == "__main__":
    B, H, N, NP = 128, 12, 197, 208
    Cd = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B * N, 3 * Cd, device="cuda").to(torch.bfloat16)
    biasP = torch.randn(H, NP, NP, device="cuda"); biasP[:, :, N:] = -1e30
    ref = {}
    for name in sys.argv[1:]:
        L = C.CDLL(os.path.join(ROOT, "uncertainty-vit_amd", name))
        for p in (0.0, 0.05):
            out = torch.zeros(B * N, Cd, device="cuda", dtype=torch.bfloat16)
            lse = torch.zeros(B, H, N, device="cuda")
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            fn = lambda: L.uvit_debug_attn_fwd(P(qkv), P(biasP), P(out), P(lse), B, H, N, NP, C.c_float(0.125), C.c_float(p), st)  # noqa: E731
            for _ in range(3):
                assert fn() == 0
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b.record(); torch.cuda.synchronize()
            ok = ""
            if p in ref:
                ok = "  identical" if torch.equal(out, ref[p]) else "  MISMATCH"
            else:
                ref[p] = out.clone()
            print(f"{name:28s} p={p}: fwd {a.elapsed_time(b) / 20 * 1e3:7.1f} us{ok}", flush=True)
