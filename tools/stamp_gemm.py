"""Diagnostic: per-CU time line of one gemm_nt256 launch from in-kernel s_memtime stamps (segment shares, not run time).

Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -DUVIT_SRC_HASH='"dbg"' -DGEMM_STAMP \
            -shared uncertainty-vit_amd/csrc/gemm.hip -o uncertainty-vit_amd/libgemm_stamp.so
Run on the GPU box:  python tools/stamp_gemm.py [mode] [N] [K]      (mode 0 bf16, 8 GELU+GELU', 3 residual)"""
import ctypes as C
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "uncertainty-vit_amd", "libgemm_stamp.so"))
P = lambda t: C.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731

if __name__ == "__main__":
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    M = 25216
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    f32 = mode in (3, 4)
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    out2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    resid = torch.randn(M, N, device="cuda") if mode == 3 else None
    gamma = torch.ones(N, device="cuda") if mode == 3 else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    variant = int(sys.argv[4]) if len(sys.argv) > 4 else 101
    run = lambda: L.uvit_debug_gemm_nt(mode, variant, P(A), P(W), M, N, K, P(out), P(out2), P(bias), P(resid), P(gamma), st)  # noqa: E731
    for _ in range(3):
        assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); assert run() == 0; e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4096 * 8))()
    assert L.uvit_debug_gemm_stamps(buf) == 0
    t = np.array(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    ph = t[3500:3500 + 32].reshape(-1)[:4 * 2 * 32].reshape(4, 2, 32)
    if ph[0, 0, 0]:
        names = {0: "p0 start", 1: "p0 reads+DMA issued", 2: "p0 vmcnt wait done", 3: "p0 barrier 1", 4: "p0 lds wait", 5: "p0 MMA issued",
                 6: "p1 start (barrier 2)", 7: "p1 reads+DMA issued", 8: "p1 vmcnt wait done", 9: "p1 barrier 1", 10: "p1 lds wait", 11: "p1 MMA issued",
                 12: "p2 start (barrier 2)", 13: "p2 reads+DMA issued", 15: "p2 barrier 1", 16: "p2 lds wait", 17: "p2 MMA issued",
                 18: "p3 start (barrier 2)", 19: "p3 DMA issued", 20: "p3 vmcnt wait done", 21: "p3 barrier 1", 23: "p3 MMA issued", 24: "p3 barrier 2"}
        print("K-tile 5 of the first tile, cycles since the K-tile's start; columns: workgroup 0..3 x wave group 0 / 1")
        for k in sorted(names):
            print(f"  {names[k]:24s} " + " ".join(f"{ph[w, gq, k] - ph[w, 0, 0]:6d}" for w in range(4) for gq in range(2)))
    if variant < 100:
        # persistent form: per workgroup, tiles 0..7: [1] K loop start, [2] K loop end, [4] next tile's operands landed, [3] epilogue end
        tp = t[2048:].reshape(256, 8, 8)
        print(f"mode {mode}  M {M} N {N} K {K} persistent: launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build)")
        for wg in (0, 1, 100, 255):
            rows = []
            for sq in range(8):
                r = tp[wg, sq]
                if r[1] == 0 or (sq and r[1] < tp[wg, sq - 1][1]):
                    break
                prev_end = tp[wg, sq - 1][3] if sq else r[1]
                rows.append(f"tile {sq}: top-of-loop gap {r[1] - prev_end:6d}  K loop {r[2] - r[1]:6d}  landed-wait {(r[4] - r[2]) if r[4] > r[2] else -1:6d}  epilogue {r[3] - max(r[4], r[2]):6d}")
            print(f"  workgroup {wg}: " + " | ".join(rows))
        sys.exit(0)
    bm = 320 if variant % 100 == 5 else 256
    ntiles = ((M + bm - 1) // bm) * (N // 256)
    t = t[:ntiles]
    hw, xcc = t[:, 6], t[:, 7] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
    print(f"mode {mode}  M {M} N {N} K {K}: {ntiles} tiles on {len(set(cu.tolist()))} CUs, launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build)")
    start = t[:, 0].min()
    pro, main, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"per tile (shader cycles): prologue (entry -> first operands landed) mean {pro.mean():.0f}  p10 {np.percentile(pro, 10):.0f}  p90 {np.percentile(pro, 90):.0f}")
    print(f"                          K loop   mean {main.mean():.0f}  p10 {np.percentile(main, 10):.0f}  p90 {np.percentile(main, 90):.0f}")
    print(f"                          epilogue mean {epi.mean():.0f}  p10 {np.percentile(epi, 10):.0f}  p90 {np.percentile(epi, 90):.0f}")
    by = defaultdict(list)
    for i in range(ntiles):
        by[int(cu[i])].append((t[i, 0] - start, t[i, 3] - start))
    gaps, busy, last = [], [], []
    for k, v in by.items():
        v.sort()
        for (a0, a1), (b0, b1) in zip(v[:-1], v[1:]):
            gaps.append(b0 - a1)
        busy.append(sum(b - a for a, b in v))
        last.append(v[-1][1])
    gaps = np.array(gaps)
    if len(gaps):
        print(f"gap between a tile's end and the next tile's entry on the same CU: mean {gaps.mean():.0f}  p10 {np.percentile(gaps, 10):.0f}  p90 {np.percentile(gaps, 90):.0f} cycles")
    print(f"launch span (first entry -> last end) {max(last)} cycles; per-CU busy mean {np.mean(busy):.0f}; tiles per CU min {min(len(v) for v in by.values())} max {max(len(v) for v in by.values())}")
    print(f"first entries: p50 {np.percentile([v[0][0] for v in by.values()], 50):.0f}  max {max(v[0][0] for v in by.values())};  last ends: min {min(last)}  p50 {np.percentile(last, 50):.0f}")
