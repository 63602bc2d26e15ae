"""Grouped wgrad of one ViT-B layer (M = 25216): the one-round plan (216 items on 256 CUs) against the balanced plan (round 4: main pieces
shortened, the tails of every tile on the 40 idle CUs).  Run twice: UVIT_TN_BALANCE=0 python tools/bench_wgrad_balance.py ; python tools/bench_wgrad_balance.py
Also checks the result against the fp32 torch product."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gemm import P, S, L, TUNE  # noqa: E402
from uncertainty_vit_amd.native import WgradProblem  # noqa: E402

M, Cd, Hd = 25216, 768, 3072
Mpad = (M + 127) // 128 * 128
specs = [(Cd, Hd), (Hd, Cd), (Cd, Cd), (3 * Cd, Cd)]
bufs = []
probs = (WgradProblem * 4)()
torch.manual_seed(0)
for i, (n, k) in enumerate(specs):
    y = torch.zeros(Mpad, n, device="cuda", dtype=torch.bfloat16); y[:M] = (torch.randn(M, n, device="cuda") * 0.1).to(torch.bfloat16)
    x = torch.zeros(Mpad, k, device="cuda", dtype=torch.bfloat16); x[:M] = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    out = torch.zeros(n, k, device="cuda"); b = torch.zeros(n, device="cuda")
    bufs.append((y, x, out, b))
    q = probs[i]; q.Y, q.X, q.C = y.data_ptr(), x.data_ptr(), out.data_ptr()
    q.bias = b.data_ptr() if i in (1, 3) else None; q.bias2 = None; q.bias_end = n if i == 1 else Cd; q.bias2_begin = 2 * Cd
    q.M, q.N, q.K, q.ldy, q.ldx, q.ldc = Mpad, n, k, n, k, k
TUNE.tn_variant = 3
TUNE.wgrad_group_chunks = 0
assert L.uvit_op_wgrad_group(probs, 4, C.byref(TUNE), S()) == 0
torch.cuda.synchronize()
for (y, x, out, b), (n, k) in zip(bufs, specs):
    ref = y.float().t() @ x.float()
    err = (out - ref).norm() / ref.norm()
    assert err < 2e-3, err
    out.zero_()
y, x, out, b = bufs[1]
assert ((b - y.float().sum(0)).norm() / y.float().sum(0).norm()) < 2e-3
for _ in range(3):
    L.uvit_op_wgrad_group(probs, 4, C.byref(TUNE), S())
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(20):
    L.uvit_op_wgrad_group(probs, 4, C.byref(TUNE), S())
en.record(); torch.cuda.synchronize()
us = st.elapsed_time(en) / 20 * 1e3
flops = sum(2.0 * M * n * k for n, k in specs)
print(f"UVIT_TN_BALANCE={os.environ.get('UVIT_TN_BALANCE', '1 (default)')}: layer wgrads, grouped: {us:7.1f} us  {flops / us / 1e6:6.0f} TF/s  (results checked)")
