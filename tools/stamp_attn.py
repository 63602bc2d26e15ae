"""Diagnostic: time line of the fused attention backward from in-kernel s_memtime stamps (segment shares, not run time).

Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -DUVIT_SRC_HASH='"dbg"' -DATTN_STAMP \
            -shared uncertainty-vit_amd/csrc/attention.hip -o uncertainty-vit_amd/libattn_stamp.so
Run on the GPU box:  python tools/stamp_attn.py [p_drop] [with_dbias]"""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "uncertainty-vit_amd", "libattn_stamp.so"))
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731

if __name__ == "__main__":
    p_drop = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
    with_dbias = len(sys.argv) > 2 and sys.argv[2] == "1"
    B, H, N, NP = 128, 12, 197, 208
    Cd = H * 64
    torch.manual_seed(0)
    qkv = torch.randn(B * N, 3 * Cd, device="cuda").to(torch.bfloat16)
    biasP = torch.zeros(H, NP, NP, device="cuda"); biasP[:, :, N:] = -1e30
    out = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16)
    d_o = torch.randn(B * N, Cd, device="cuda").to(torch.bfloat16)
    lse = torch.full((B, H, N), 9.0, device="cuda"); delta = torch.zeros_like(lse)
    dqkv = torch.zeros_like(qkv)
    ws = torch.empty(B * H * 7 * 13312, dtype=torch.uint8, device="cuda")
    slab = torch.zeros(H, NP, NP, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.uvit_debug_attn_bwd_fused.argtypes = [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_float, C.c_void_p]
    run = lambda: L.uvit_debug_attn_bwd_fused(P(qkv), P(out), P(d_o), P(biasP), P(lse), P(delta), P(dqkv), P(ws if with_dbias else None),  # noqa: E731
                                              P(slab if with_dbias else None), B, H, N, p_drop, st)
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); assert run() == 0; e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2048 * 2 * 16 * 2))()
    assert L.uvit_debug_attn_stamps(buf) == 0
    allst = np.array(buf, dtype=np.uint64).astype(np.int64)
    t = allst[:2048 * 32].reshape(2048, 2, 16)[:B * H]
    sub = allst[2048 * 32:].reshape(2048, 2, 16)[:B * H]
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), p_drop {p_drop}, dbias stream {with_dbias}")
    w0 = t[:, 0, :]
    sub_names = {1: "iteration start", 2: "B_2 / streaming done", 4: "A: bias copies + prefetch issued", 5: "A: dropout draw", 6: "A: K rows + S^T MFMAs issued",
                 7: "A: V rows + dP^T MFMAs issued", 8: "A: softmax bwd + P/dS written", 9: "A: K^T cols + dQ^T MFMAs issued", 10: "lgkmcnt(0) before barrier", 11: "barrier passed"}
    for wv, nm in ((0, "wave 0 (B + A)"), (1, "wave 8 (streaming + A)")):
        print(f"step 3 sub-stages of {nm}: cycles since iteration start (median over workgroups)")
        for k in sorted(sub_names):
            print(f"   {sub_names[k]:36s} {np.median(sub[:, wv, k] - sub[:, wv, 1]):8.0f}")
    names = ["start", "own loads + DMA landed (vmcnt 0)", "image barrier"] + [f"barrier {i}" for i in range(7)] + ["", "last B done", "final barrier", "end (dQ stored)"]
    seg = np.diff(w0[:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13]], axis=1)
    labels = ["load wait", "barrier", "A0", "B0+A1", "B1+A2", "B2+A3", "B3+A4", "B4+A5", "B5+A6", "B6", "rows + dQ out"]
    print("wave 0 segments, cycles (median / p10 / p90 over workgroups):")
    for k, lb in enumerate(labels):
        v = seg[:, k]
        print(f"  {lb:14s} {np.median(v):8.0f} {np.percentile(v, 10):8.0f} {np.percentile(v, 90):8.0f}")
    life = w0[:, 13] - w0[:, 0]
    print(f"  workgroup life {np.median(life):8.0f} {np.percentile(life, 10):8.0f} {np.percentile(life, 90):8.0f}")
    w8 = t[:, 1, :]
    life8 = w8[:, 13] - w8[:, 0]
    print(f"  wave 8 life    {np.median(life8):8.0f}")
    # per-CU time line: workgroups by (xcc, cu id)
    ids = w0[:, 15]
    by = defaultdict(list)
    for i in range(len(w0)):
        by[int(ids[i])].append((w0[i, 0], w0[i, 13]))
    gaps, spans = [], []
    for k, v in by.items():
        v.sort()
        spans.append(v[-1][1] - v[0][0])
        gaps += [v[j + 1][0] - v[j][1] for j in range(len(v) - 1)]
    print(f"CUs seen {len(by)}, workgroups per CU {np.mean([len(v) for v in by.values()]):.2f}, CU span median {np.median(spans):.0f} cycles, "
          f"turnover gap median {np.median(gaps):.0f} (p90 {np.percentile(gaps, 90):.0f})")
