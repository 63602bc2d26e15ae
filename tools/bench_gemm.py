"""Micro-benchmark of the GEMM kernels through the C ABI (run on the GPU box)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uncertainty_vit_amd import native  # noqa: E402
from uncertainty_vit_amd.native import GemmEpilogue, Tuning  # noqa: E402

L = native.lib()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


TUNE = Tuning.default()          # launch tuning is an argument of every call (no process-wide state)
COLD = "--cold" in sys.argv      # evict the Infinity Cache between launches: operands come from HBM, as inside the step
_flush = None


def _timed(fn, iters):
    """Mean duration of fn() in us; with --cold a 640 MB buffer is rewritten before every launch and only fn is timed."""
    global _flush
    if not COLD:
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        torch.cuda.synchronize()
        return st.elapsed_time(en) / iters * 1e3
    if _flush is None:
        _flush = torch.empty(160 * 1024 * 1024, device="cuda")
    tot = 0.0
    for i in range(iters):
        _flush.fill_(float(i))
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); fn(); en.record()
        torch.cuda.synchronize()
        tot += st.elapsed_time(en)
    return tot / iters * 1e3


def time_nt(mode, M, N, K, iters=20):
    a = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    gamma = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda") if mode == 3 else None
    aux = torch.randn(M, N, device="cuda").to(torch.bfloat16) if mode in (6, 9) else None
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if mode in (3, 4) else torch.bfloat16)
    out2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if mode in (2, 3, 8) else None
    e = GemmEpilogue()
    e.out = out.data_ptr(); e.out2 = out2.data_ptr() if out2 is not None else 0
    e.bias = bias.data_ptr(); e.bias2 = bias.data_ptr(); e.gamma = gamma.data_ptr()
    e.resid = resid.data_ptr() if resid is not None else 0
    e.aux = aux.data_ptr() if aux is not None else 0
    e.ldo, e.tokens, e.patches = N, 197, 196
    run = lambda: L.uvit_op_gemm_nt_tuned(mode, P(a), P(w), M, N, K, K, K, C.byref(e), C.byref(TUNE), None, S())  # noqa: E731
    for _ in range(3):
        assert run() == 0
    us = _timed(run, iters)
    return us, 2.0 * M * N * K / us / 1e6


def time_tn(M, N, K, iters=20):
    y = (torch.randn(M, N, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    out = torch.zeros(N, K, device="cuda")
    run = lambda: L.uvit_op_gemm_tn(P(y), P(x), M, N, K, N, K, P(out), K, C.byref(TUNE), S())  # noqa: E731
    for _ in range(3):
        assert run() == 0
    us = _timed(run, iters)
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__" and "bench_gemm" in sys.argv[0]:
    M = 25216
    names = {0: "bf16", 1: "qkv", 2: "gelu", 3: "resid", 4: "f32", 6: "dgelu", 8: "gelu_dg", 9: "mulaux"}
    print("== NT: step shapes, variants 0 (128x128, 2 WG/CU) / 1 (256x256 staggered, 1 WG/CU) ==")
    for mode, N, K in ((1, 2304, 768), (3, 768, 768), (2, 3072, 768), (8, 3072, 768), (3, 768, 3072), (9, 3072, 768), (0, 768, 3072), (0, 768, 768), (0, 768, 2304)):
        row = []
        for v in (0, 1, 5, 6, 7, 3):
            TUNE.nt_variant = v
            us, tf = time_nt(mode, M, N, K)
            row.append(f"{'auto' if v == 3 else 'v%d' % v}: {us:6.1f} us {tf:5.0f} TF")
        print(f"{names[mode]:6s} N={N:5d} K={K:5d}: " + " | ".join(row))
    TUNE.nt_variant = 3
    if "--persist" in sys.argv:
        print("== NT: persistent workgroups (nt_persist) on / off, 256-row kernel ==")
        TUNE.nt_variant = 1
        for mode, N, K in ((8, 3072, 768), (2, 3072, 768), (9, 3072, 768), (6, 3072, 768), (1, 2304, 768), (0, 3072, 768), (4, 3072, 768)):
            row = []
            for pz in (1, 0, 1, 0):
                TUNE.nt_persist = pz
                us, tf = time_nt(mode, M, N, K)
                row.append(f"persist {pz}: {us:6.1f} us {tf:5.0f} TF")
            print(f"{names[mode]:7s} N={N:5d} K={K:5d}: " + " | ".join(row))
        TUNE.nt_persist = 1
        TUNE.nt_variant = 3
        sys.exit(0)
    if "--group" in sys.argv:
        print("== NT: column tiles per row-tile group of the tile order (nt_group), 256-row kernel ==")
        TUNE.nt_variant = 1
        for mode, N, K in ((8, 3072, 768), (2, 3072, 768), (9, 3072, 768), (1, 2304, 768), (0, 3072, 3072)):
            row = []
            for gsz in (1, 2, 3, 4, 6, 12):
                TUNE.nt_group = gsz
                us, tf = time_nt(mode, M, N, K)
                row.append(f"g{gsz}: {us:6.1f} us")
            print(f"{names[mode]:7s} N={N:5d} K={K:5d}: " + " | ".join(row))
        TUNE.nt_group = 0
        TUNE.nt_variant = 3
        sys.exit(0)
    if "--quick" in sys.argv:
        sys.exit(0)
    print("== NT: square references ==")
    for n in (4096, 8192):
        row = []
        for v in (0, 1, 6):
            TUNE.nt_variant = v
            us, tf = time_nt(0, n, n, n, iters=5)
            row.append(f"v{v}: {us:8.1f} us {tf:7.1f} TF/s")
        print(f"{n}^3: " + " | ".join(row))
    print("== NT: K sweep at M=25216 N=3072 ==")
    for K in (128, 256, 768, 1536, 3072):
        row = []
        for v in (0, 1, 6):
            TUNE.nt_variant = v
            us, tf = time_nt(0, M, 3072, K)
            row.append(f"v{v}: {us:7.1f} us {tf:6.1f} TF/s")
        print(f"K={K:5d}: " + " | ".join(row))
    TUNE.nt_variant = 3
    if "--tn" not in sys.argv:
        sys.exit(0)
    print("== TN (wgrad), split target sweep ==")
    for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        row = []
        for tgt in (256, 384, 512, 768, 1024, 1536):
            TUNE.tn_split_target = tgt
            us, tf = time_tn(M, N, K)
            row.append(f"{tgt}: {us:6.1f}us {tf:5.0f}TF")
        print(f"wgrad N={N:5d} K={K:5d}: " + " | ".join(row))
    TUNE.tn_split_target = 512
