"""Diagnostic: time line of the fused two-stream attention backward from in-kernel s_memtime stamps (segment shares, not run time).

Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -DUVIT_SRC_HASH='"dbg"' -DATTN2_STAMP \
            -shared uncertainty-vit_amd/csrc/attention2.hip uncertainty-vit_amd/csrc/optim.hip -o uncertainty-vit_amd/libattn2_stamp.so
Run on the GPU box:  python tools/stamp_attn2.py [p_drop] [with_dbias]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "uncertainty-vit_amd", "libattn2_stamp.so"))
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731

if __name__ == "__main__":
    p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
    with_dbias = len(sys.argv) > 2 and sys.argv[2] == "1"
    B, H, N, NP = 128, 12, 197, 208
    Cd = H * 64
    torch.manual_seed(0)
    qkv_m = torch.randn(B * N, 3 * Cd, device="cuda").to(torch.bfloat16)
    qkv_c = (torch.nn.functional.elu(torch.randn(B * N, 3 * Cd, device="cuda")) + 1).to(torch.bfloat16)
    biasP = torch.zeros(H, NP, NP, device="cuda"); biasP[:, :, N:] = -1e30
    o_m = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16); o_c = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16)
    d_m = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16); d_c = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16)
    lse = torch.full((B, H, N), 9.0, device="cuda"); delta = torch.zeros_like(lse)
    dq_m = torch.zeros_like(qkv_m); dq_c = torch.zeros_like(qkv_m)
    ws = torch.empty(B * H * 13 * 13 * 512, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.uvit_debug_attn2_bwd.argtypes = [C.c_void_p] * 12 + [C.c_int] * 3 + [C.c_float, C.c_void_p]
    run = lambda: L.uvit_debug_attn2_bwd(P(qkv_m), P(qkv_c), P(o_m), P(o_c), P(d_m), P(d_c), P(biasP), P(lse), P(delta), P(dq_m), P(dq_c),  # noqa: E731
                                         P(ws if with_dbias else None), B, H, N, p_drop, st)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); assert run() == 0; e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2048 * 3 * 32))()
    assert L.uvit_debug_attn2_stamps(buf) == 0
    t = np.array(buf, dtype=np.uint64).astype(np.int64).reshape(2048, 3, 32)[:B * H]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), p_drop {p_drop}, dS stream {with_dbias}")
    med = lambda v: f"{np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}"  # noqa: E731
    for wv, nm in ((0, "wave 0 (B role 0 + A)"), (1, "wave 4 (B role 1 + A)"), (2, "wave 8 (key loader + A)")):
        w = t[:, wv, :]
        print(f"== {nm}: cycles (median / p10 / p90 over workgroups)")
        print(f"  preamble issue        {med(w[:, 1] - w[:, 0])}")
        print(f"  wait vmcnt/lgkm       {med(w[:, 2] - w[:, 1])}")
        print(f"  image barrier         {med(w[:, 3] - w[:, 2])}")
        prev = w[:, 3]
        for i in range(14):
            cur = w[:, 4 + i]
            print(f"  iteration {i:2d}          {med(cur - prev)}")
            prev = cur
        print(f"  tail (dq epilogue)    {med(w[:, 19] - w[:, 18 if False else 17])}")
        print(f"  workgroup life        {med(w[:, 19] - w[:, 0])}")
        print("  iteration 6 sub-stages (cycles since iteration start):")
        names = {22: "B_5 MFMA loop done", 23: "B_5 / loader done", 24: "A: bias, dropout, c_j", 25: "A: S MFMAs issued", 26: "A: dP MFMAs issued",
                 27: "A: VALU + step-buffer writes + dS store", 28: "A: dA MFMAs issued", 29: "waitcnt before barrier", 30: "barrier passed"}
        for k in sorted(names):
            print(f"     {names[k]:42s} {np.median(w[:, k] - w[:, 21]):8.0f}")
