"""GPU parity of each HIP operator, called through the C ABI, against a plain PyTorch fp32
restatement of the same reference arithmetic on the same seeded inputs."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from uncertainty_vit_amd import native
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return native.lib()


def P(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ok(rc):
    assert rc == 0, f"libuvit returned {rc}"


def bf(t):
    return t.to(torch.bfloat16).contiguous()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


class tuned:
    """Launch tuning for the GEMM calls inside the block.  It is an ARGUMENT of every call (uvit_tuning): the library
    keeps no process-wide launcher state."""
    cur = None

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        from uncertainty_vit_amd.native import Tuning
        self.prev, tuned.cur = tuned.cur, Tuning.default(**self.kw)
        return tuned.cur

    def __exit__(self, *a):
        tuned.cur = self.prev


def _tune_ref():
    return None if tuned.cur is None else C.byref(tuned.cur)


def nt(L, mode, a, w, M, N, K, lda, ldw, ep, stream, tail=None):
    return L.uvit_op_gemm_nt_tuned(mode, a, w, M, N, K, lda, ldw, ep, _tune_ref(), None if tail is None else C.byref(tail), stream)


def tn(L, y, x, M, N, K, ldy, ldx, out, ldc, stream):
    return L.uvit_op_gemm_tn(y, x, M, N, K, ldy, ldx, out, ldc, _tune_ref(), stream)


def epi(**kw):
    from uncertainty_vit_amd.native import GemmEpilogue
    e = GemmEpilogue()
    for k, v in kw.items():
        setattr(e, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return e


# bf16 inputs, fp32 accumulate: error budget for a K-long dot of O(1) terms
def close(a, b, rtol=2e-2, atol=2e-2, what=""):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).sum().item()
    assert bad == 0, f"{what}: {bad}/{a.numel()} off, max err {err.max().item():.4g} (ref max {b.abs().max().item():.4g})"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 192, 128), (1000, 768, 768), (394, 2304, 192), (64, 3072, 768)])
def test_gemm_nt_bias_bf16_and_f32(L, M, N, K):
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.05, seed=2)), rnd(N, seed=3)
    ref = a.float() @ w.float().t() + b
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(nt(L, 0, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out, bias=b, ldo=N)), S()))
    close(out, ref, what="bf16 out")
    out32 = torch.zeros(M, N, device="cuda")
    ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), S()))
    close(out32, ref, rtol=2e-3, atol=2e-3, what="f32 out")


@pytest.mark.parametrize("variant", [0, 1, 5, 6, 7])
@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (25216, 3072, 768), (2048, 768, 3072), (1100, 2304, 768), (1024, 256, 128), (1024, 256, 64)])
def test_gemm_nt_large_tile_kernel(L, M, N, K, variant):
    """Shapes that dispatch to the 256x256 deep-prefetch kernel (N % 256 == 0, M >= 1024): parity, a ragged
    last row tile, and a race screen (the counted-vmcnt pipeline must give bit-identical results every launch)."""
    with tuned(nt_variant=variant):
        a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.05, seed=2)), rnd(N, seed=3)
        ref = a.float() @ w.float().t() + b
        out32 = torch.zeros(M, N, device="cuda")
        ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), S()))
        close(out32, ref, rtol=2e-3, atol=2e-3, what="f32 out")
        first = out32.clone()
        for _ in range(10):
            out32.zero_()
            ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), S()))
            assert torch.equal(out32, first), "non-deterministic result: LDS pipeline race"


@pytest.mark.parametrize("variant", [0, 1, 5, 6, 7])
def test_gemm_nt_large_identity(L, variant):
    """A = [I; I; ...] against an asymmetric W on the large-tile kernels: exact, catches any fragment / quadrant mix-up."""
    K, N, M = 256, 512, 2048
    a = bf(torch.eye(K).repeat(M // K, 1).cuda())
    w = bf(((torch.arange(N * K).reshape(N, K) * 7) % 509 - 254).float().cuda() / 128)
    out = torch.zeros(M, N, device="cuda")
    with tuned(nt_variant=variant):
        ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out, ldo=N)), S()))
    torch.testing.assert_close(out, w.float().t().repeat(M // K, 1).contiguous(), rtol=0, atol=0)


def test_gemm_nt_asymmetric_identity(L):
    """A = I with an asymmetric W catches a transposed C write (guide section 3)."""
    K = 128
    a = bf(torch.eye(K).cuda())
    w = bf((torch.arange(192 * K).reshape(192, K) % 251 - 125).float().cuda() / 64)
    out = torch.zeros(K, 192, device="cuda")
    ok(nt(L, 4, P(a), P(w), K, 192, K, K, K, C.byref(epi(out=out, ldo=192)), S()))
    torch.testing.assert_close(out, w.float().t().contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("variant,nb,tokens,Cd,Pn", [(3, 3, 10, 128, 9), (0, 11, 100, 256, 140), (1, 11, 100, 256, 140), (5, 11, 100, 256, 140),
                                                     (6, 11, 100, 256, 140), (7, 11, 100, 256, 140)])
def test_gemm_nt_qkv_gelu_resid_dgelu_patch(L, variant, nb, tokens, Cd, Pn):
    """Every fused epilogue, on the 128x128 kernel (small shapes) and on each large-tile variant (ragged M = 1100)."""
    with tuned(nt_variant=variant):
        _epilogue_modes(L, nb, tokens, Cd, Pn)


def test_gemm_nt_persistent_workgroups(L):
    """gemm_nt256_kernel<MODE, 4, PERSIST>: one workgroup per CU walks several tiles with the operand pipeline running across
    them (the next tile's first K-tiles are fetched during the current tile's last ones, the epilogue stages through its own
    LDS region).  Every fused epilogue at a size where each GEMM has more tiles than CUs; then, against one workgroup per
    tile, bit-identical results for an odd K-tile count (the LDS buffer parity flips from tile to tile), a ragged last row
    tile and the minimum K, with a race screen."""
    with tuned(nt_variant=1):
        _epilogue_modes(L, 700, 100, 256, 140)
    for M, N, K in [(25216, 3072, 192), (25216, 2304, 768), (70000, 256, 128), (25216, 768, 320)]:
        a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.05, seed=2)), rnd(N, seed=3)
        ref = a.float() @ w.float().t() + b
        outs = []
        for persist in (1, 0):
            with tuned(nt_variant=1, nt_persist=persist):
                out32 = torch.zeros(M, N, device="cuda")
                ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), S()))
                outs.append(out32.clone())
                for _ in range(5):
                    out32.zero_()
                    ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), S()))
                    assert torch.equal(out32, outs[-1]), "non-deterministic result: LDS pipeline race"
        assert torch.equal(outs[0], outs[1]), (M, N, K)
        close(outs[0], ref, rtol=2e-3, atol=2e-3, what=f"persistent {M}x{N}x{K}")


def test_gemm_nt_persistent_dynamic_tile_assignment(L):
    """uvit_op_gemm_nt_sched: the persistent workgroups take their tiles from per-XCD counters instead of by a fixed stride (round 4).
    Bit-identical to the fixed-stride launch (a tile's arithmetic does not depend on who computes it), also when another kernel keeps part
    of the GPU busy so that workgroups start late and the shares differ from launch to launch; the counters are back at zero every time."""
    from uncertainty_vit_amd.native import Tuning
    cnt = torch.zeros(16, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    junk = torch.randn(8192, 8192, device="cuda")
    for M, N, K in [(25216, 3072, 192), (25216, 2304, 768), (9000, 768, 320), (300, 256, 128)]:
        a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.05, seed=2)), rnd(N, seed=3)
        out32 = torch.zeros(M, N, device="cuda")
        with tuned(nt_variant=1, nt_persist=1) as tu:
            ok(nt(L, 4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), S()))
            ref = out32.clone()
            for rep in range(8):
                if rep % 2:
                    with torch.cuda.stream(side):
                        junk = (junk @ junk) * 1e-4          # a long kernel that holds CUs while the persistent workgroups are dispatched
                out32.zero_()
                ok(L.uvit_op_gemm_nt_sched(4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out32, bias=b, ldo=N)), C.byref(tu), P(cnt), S()))
                torch.cuda.synchronize()
                assert torch.equal(out32, ref), (M, N, K, rep)
                assert int(cnt.abs().sum()) == 0, "the tile counters were not put back to zero"
        close(ref, a.float() @ w.float().t() + b, rtol=2e-3, atol=2e-3, what=f"dynamic {M}x{N}x{K}")


def test_gemm_nt_row_split_of_a_nearly_empty_last_round(L):
    """Auto dispatch sends the row tiles that overflow whole rounds of 256x256 tiles (fc2: 297 tiles on 256 CUs) to the
    128x128 kernel: the residual epilogue's per-sample drop-path scale must keep indexing by the global row."""
    tokens, B, N, K = 197, 128, 768, 2048
    M = B * tokens
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.05, seed=2))
    b2, gam, res = rnd(N, seed=3), rnd(N, scale=0.1, seed=4), rnd(M, N, seed=5)
    dp = (torch.arange(B, device="cuda") % 3).float() * 0.625
    xo = torch.zeros(M, N, device="cuda"); branch = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ok(nt(L, 3, P(a), P(w), M, N, K, K, K,
                         C.byref(epi(out=xo, out2=branch, bias=b2, gamma=gam, resid=res, rowscale=dp, ldo=N, tokens=tokens)), S()))
    y = a.float() @ w.float().t() + b2
    close(branch, y, what="resid branch")
    close(xo, res + dp.repeat_interleave(tokens)[:, None] * gam * y, rtol=5e-3, atol=5e-3, what="resid out")


def _epilogue_modes(L, nb, tokens, Cd, Pn):
    M, Hd = nb * tokens, 512
    x = bf(rnd(M, Cd, seed=4))
    # QKV: bias = cat(q_bias, 0, v_bias)
    w = bf(rnd(3 * Cd, Cd, scale=0.05, seed=5)); qb, vb = rnd(Cd, seed=6), rnd(Cd, seed=7)
    out = torch.zeros(M, 3 * Cd, dtype=torch.bfloat16, device="cuda")
    ok(nt(L, 1, P(x), P(w), M, 3 * Cd, Cd, Cd, Cd, C.byref(epi(out=out, bias=qb, bias2=vb, ldo=3 * Cd)), S()))
    close(out, x.float() @ w.float().t() + torch.cat([qb, torch.zeros_like(vb), vb]), what="qkv")
    # GELU: a = gelu(h), h kept
    w1 = bf(rnd(Hd, Cd, scale=0.1, seed=8)); b1 = rnd(Hd, seed=9)
    a_out = torch.zeros(M, Hd, dtype=torch.bfloat16, device="cuda"); h_out = torch.zeros_like(a_out)
    ok(nt(L, 2, P(x), P(w1), M, Hd, Cd, Cd, Cd, C.byref(epi(out=a_out, out2=h_out, bias=b1, ldo=Hd)), S()))
    h_ref = x.float() @ w1.float().t() + b1
    close(h_out, h_ref, what="gelu h")
    close(a_out, F.gelu(h_out.float()), rtol=1e-2, atol=1e-2, what="gelu a")
    # GELU_DG / MULAUX: the training pair -- gelu(h) with gelu'(h) stored, then dH = (dY @ W2) * gelu'(h)
    a2 = torch.zeros_like(a_out); dg = torch.zeros_like(a_out)
    ok(nt(L, 8, P(x), P(w1), M, Hd, Cd, Cd, Cd, C.byref(epi(out=a2, out2=dg, bias=b1, ldo=Hd)), S()))
    assert torch.equal(a2, a_out)
    hq = h_out.float().requires_grad_(True)
    F.gelu(hq).sum().backward()
    close(dg, hq.grad, rtol=1e-2, atol=1e-2, what="gelu'")
    # RESID: x + dp * gamma * (a @ W2^T + b2)
    w2 = bf(rnd(Cd, Hd, scale=0.05, seed=10)); b2, gam = rnd(Cd, seed=11), rnd(Cd, scale=0.1, seed=12)
    res = rnd(M, Cd, seed=13); dp = torch.tensor([0.0] + [1.25] * (nb - 1), device="cuda")
    xo = torch.zeros(M, Cd, device="cuda"); branch = torch.zeros(M, Cd, dtype=torch.bfloat16, device="cuda")
    ok(nt(L, 3, P(a_out), P(w2), M, Cd, Hd, Hd, Hd,
                         C.byref(epi(out=xo, out2=branch, bias=b2, gamma=gam, resid=res, rowscale=dp, ldo=Cd, tokens=tokens)), S()))
    y = a_out.float() @ w2.float().t() + b2
    close(branch, y, what="resid branch")
    close(xo, res + dp.repeat_interleave(tokens)[:, None] * gam * y, rtol=5e-3, atol=5e-3, what="resid out")
    # DGELU: dH = (dY @ W2) * gelu'(h)
    dy = bf(rnd(M, Cd, scale=0.1, seed=14)); w2t = bf(w2.float().t())
    dh = torch.zeros(M, Hd, dtype=torch.bfloat16, device="cuda")
    ok(nt(L, 6, P(dy), P(w2t), M, Hd, Cd, Cd, Cd, C.byref(epi(out=dh, aux=h_out, ldo=Hd)), S()))
    hh = h_out.float().requires_grad_(True)
    F.gelu(hh).backward(dy.float() @ w2.float())
    close(dh, hh.grad, rtol=2e-2, atol=5e-3, what="dgelu")
    dh2 = torch.zeros_like(dh)
    ok(nt(L, 9, P(dy), P(w2t), M, Hd, Cd, Cd, Cd, C.byref(epi(out=dh2, aux=dg, ldo=Hd)), S()))
    close(dh2, hh.grad, rtol=2e-2, atol=8e-3, what="mul-aux dgelu")
    # PATCH: rows b*P+p -> token rows b*(P+1)+1+p, masked rows take the mask token
    B, Kpe = nb - 3 if nb > 3 else 3, 768
    cols = bf(rnd(B * Pn, Kpe, seed=15)); wpe = bf(rnd(Cd, Kpe, scale=0.03, seed=16)); bpe = rnd(Cd, seed=17)
    mt = rnd(Cd, seed=18); mask = (torch.arange(B * Pn, device="cuda") % 3 == 0).long()
    x0 = torch.full((B * (Pn + 1), Cd), 7.0, device="cuda")
    ok(nt(L, 5, P(cols), P(wpe), B * Pn, Cd, Kpe, Kpe, Kpe,
                         C.byref(epi(out=x0, bias=bpe, mask=mask, mask_token=mt, ldo=Cd, patches=Pn)), S()))
    ref = cols.float() @ wpe.float().t() + bpe
    ref = torch.where(mask[:, None].bool(), mt[None, :].expand_as(ref), ref).view(B, Pn, Cd)
    got = x0.view(B, Pn + 1, Cd)
    close(got[:, 1:], ref, rtol=5e-3, atol=5e-3, what="patch")
    assert torch.all(got[:, 0] == 7.0)


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (128, 192, 576), (25216, 768, 768), (1024, 3072, 768)])
def test_gemm_tn_wgrad(L, M, N, K):
    y, x = bf(rnd(M, N, scale=0.1, seed=20)), bf(rnd(M, K, seed=21))
    out = torch.zeros(N, K, device="cuda")
    ok(tn(L, P(y), P(x), M, N, K, N, K, P(out), K, S()))
    ref = y.float().t() @ x.float()
    close(out, ref, rtol=2e-3, atol=2e-3 * math.sqrt(M / 64), what="wgrad")


@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (25216, 3072, 768), (1024, 768, 3072), (512, 256, 256), (576, 2304, 768)])
def test_gemm_tn_wgrad_256_tile_kernel(L, M, N, K):
    """The 256x256 staggered wgrad kernel: split reductions (fp32 atomics), the no-split store path (M = 512),
    odd K-tile counts, and a race screen on a shape without atomics (bit-identical every launch)."""
    with tuned(tn_variant=1):
        y, x = bf(rnd(M, N, scale=0.1, seed=20)), bf(rnd(M, K, seed=21))
        ref = y.float().t() @ x.float()
        out = torch.zeros(N, K, device="cuda")
        ok(tn(L, P(y), P(x), M, N, K, N, K, P(out), K, S()))
        close(out, ref, rtol=2e-3, atol=2e-3 * math.sqrt(M / 64), what="wgrad 256")
        if M == 512:
            first = out.clone()
            for _ in range(10):
                out.fill_(7.0)     # the no-split path overwrites
                ok(tn(L, P(y), P(x), M, N, K, N, K, P(out), K, S()))
                assert torch.equal(out, first), "non-deterministic result: LDS pipeline race"


@pytest.mark.parametrize("chunks", [0, 1, 3])
def test_wgrad_group_with_fused_bias_sums(L, chunks):
    """One launch for four Linears (qkv with split q / v bias sums, proj, fc1 with bias, fc2) at different token counts."""
    from uncertainty_vit_amd.native import WgradProblem
    Cd, Hd, M = 256, 512, 1600
    specs = [(M, 3 * Cd, Cd, "qkv"), (1024, Cd, Cd, None), (M, Hd, Cd, "full"), (M, Cd, Hd, None)]
    probs = (WgradProblem * len(specs))()
    keep, refs = [], []
    for i, (m, n, k, bias) in enumerate(specs):
        y, x = bf(rnd(m, n, scale=0.1, seed=30 + i)), bf(rnd(m, k, seed=40 + i))
        out = torch.full((n, k), 0.5, device="cuda")              # accumulates into the existing value
        b1 = torch.zeros(n if bias == "full" else Cd, device="cuda") if bias else None
        b2 = torch.zeros(Cd, device="cuda") if bias == "qkv" else None
        keep.append((y, x, out, b1, b2))
        q = probs[i]
        q.Y, q.X, q.C = y.data_ptr(), x.data_ptr(), out.data_ptr()
        q.bias, q.bias2 = (b1.data_ptr() if b1 is not None else None), (b2.data_ptr() if b2 is not None else None)
        q.bias_end, q.bias2_begin = (n if bias == "full" else Cd), 2 * Cd
        q.M, q.N, q.K, q.ldy, q.ldx, q.ldc = m, n, k, n, k, k
        refs.append((y.float().t() @ x.float() + 0.5, y.float().sum(0)))
    with tuned(wgrad_group_chunks=chunks) as tu:
        ok(L.uvit_op_wgrad_group(probs, len(specs), C.byref(tu), S()))
    for (y, x, out, b1, b2), (ref, colsum), (m, n, k, bias) in zip(keep, refs, specs):
        close(out, ref, rtol=2e-3, atol=2e-3 * math.sqrt(m / 64), what="grouped wgrad")
        if bias == "full":
            close(b1, colsum, rtol=2e-3, atol=2e-2, what="bias sums")
        if bias == "qkv":
            close(b1, colsum[:Cd], rtol=2e-3, atol=2e-2, what="q bias sums")
            close(b2, colsum[2 * Cd:], rtol=2e-3, atol=2e-2, what="v bias sums")
    # a problem that does not qualify is refused, not approximated
    probs[0].N = 3 * Cd - 8
    assert L.uvit_op_wgrad_group(probs, len(specs), None, S()) == -2


def test_gemm_tn_256_tile_identity(L):
    """Y = [I; 0] picks rows of an asymmetric X: exact, catches fragment / quadrant / column-map mix-ups."""
    M, N, K = 512, 512, 256
    y = torch.zeros(M, N, device="cuda"); y[torch.arange(M), torch.arange(M)] = 1.0
    x = ((torch.arange(M * K).reshape(M, K) * 5 % 251) - 125).float().cuda() / 64
    out = torch.zeros(N, K, device="cuda")
    yb, xb = bf(y), bf(x)
    with tuned(tn_variant=1):
        ok(tn(L, P(yb), P(xb), M, N, K, N, K, P(out), K, S()))
    torch.testing.assert_close(out, xb.float(), rtol=0, atol=0)


def test_gemm_tn_asymmetric(L):
    M, N, K = 64, 128, 192
    y = torch.zeros(M, N, device="cuda"); y[torch.arange(M), torch.arange(M)] = 1.0     # Y^T picks rows of X
    x = ((torch.arange(M * K).reshape(M, K) % 127) - 63).float().cuda() / 32
    out = torch.zeros(N, K, device="cuda")
    yb, xb = bf(y), bf(x)          # keep the operands alive across the asynchronous launch
    ok(tn(L, P(yb), P(xb), M, N, K, N, K, P(out), K, S()))
    ref = torch.zeros(N, K, device="cuda"); ref[:M] = xb.float()
    torch.testing.assert_close(out, ref, rtol=0, atol=0)


def attn_ref(qkv, bias, B, H, N, keep=None):
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-2, -1)
    if bias is not None:
        s = s + bias
    a = s.softmax(-1)
    lse = torch.logsumexp(s, -1)
    if keep is not None:
        a = a * keep
    return (a @ v).transpose(1, 2).reshape(B, N, H * 64), lse


LOG2E = 1.4426950408889634


def padded_bias(bias, NP=208):
    """The kernels' private bias layout: bias * log2(e), padded key columns -1e30 (see uvit_op_relpos_gather)."""
    H, N, _ = bias.shape
    bp = torch.zeros(H, NP, NP, device="cuda")
    bp[:, :, N:] = -1e30
    bp[:, :N, :N] = bias * LOG2E
    return bp


@pytest.mark.parametrize("B,H,N", [(2, 2, 10), (1, 3, 5), (2, 12, 197), (3, 2, 64)])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_fwd(L, B, H, N, p_drop):
    from oracle.vit_oracle import attn_keep_mask
    Cd = H * 64
    qkv = bf(rnd(B * N, 3 * Cd, seed=30)).requires_grad_(False)
    bias = rnd(H, N, N, scale=0.5, seed=31)
    biasP = padded_bias(bias)
    seed, layer = 1234, 3
    keep = attn_keep_mask(seed, layer, B, H, N, p_drop).cuda() if p_drop > 0 else None
    out = torch.zeros(B * N, Cd, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, N, device="cuda")
    ok(L.uvit_op_attn_fwd(P(qkv), P(biasP), P(out), P(lse), B, H, N, 208, C.c_float(0.125), C.c_float(p_drop), seed, layer, S()))
    qf = qkv.float().requires_grad_(True)
    bq = bias.clone().requires_grad_(True)
    ref, lse_ref = attn_ref(qf, bq, B, H, N, keep)
    close(out.view(B, N, Cd), ref, rtol=2e-2, atol=1e-2, what="attn out")
    close(lse, lse_ref * LOG2E, rtol=1e-3, atol=3e-3, what="lse (log2 units)")


@pytest.mark.parametrize("B,H,N", [(2, 2, 10), (1, 3, 5), (2, 12, 197), (3, 2, 64), (2, 1, 33), (1, 2, 208), (5, 3, 177)])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_bwd(L, B, H, N, p_drop):
    """The fused backward (one recomputation of P, dS through LDS once, bias gradient from the streamed-out dS) against
    autograd of the fp32 reference (modeling_finetune.py:152-185), with the forward kernel's LSE and the replayed dropout mask."""
    from oracle.vit_oracle import attn_keep_mask
    Cd = H * 64
    qkv = bf(rnd(B * N, 3 * Cd, seed=30))
    bias = rnd(H, N, N, scale=0.5, seed=31)
    biasP = padded_bias(bias)
    seed, layer = 4321, 5
    keep = attn_keep_mask(seed, layer, B, H, N, p_drop).cuda() if p_drop > 0 else None
    out = torch.zeros(B * N, Cd, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, N, device="cuda")
    ok(L.uvit_op_attn_fwd(P(qkv), P(biasP), P(out), P(lse), B, H, N, 208, C.c_float(0.125), C.c_float(p_drop), seed, layer, S()))
    qf = qkv.float().requires_grad_(True)
    bq = bias.clone().requires_grad_(True)
    ref, _ = attn_ref(qf, bq, B, H, N, keep)
    d_o = bf(rnd(B * N, Cd, scale=0.5, seed=32))
    ref.backward(d_o.float().view(B, N, Cd))
    delta = torch.zeros(B, H, N, device="cuda")
    dqkv = torch.full((B * N, 3 * Cd), 7.0, dtype=torch.bfloat16, device="cuda")        # every element must be overwritten
    ws = torch.empty(L.uvit_op_attn_bwd_ws_bytes(B, H, N), dtype=torch.uint8, device="cuda")
    slab = torch.zeros(H, 208, 208, device="cuda")
    args = lambda acc: (P(qkv), P(out), P(d_o), P(biasP), P(lse), P(delta), P(dqkv), P(slab), acc, P(ws), B, H, N, 208,  # noqa: E731
                        C.c_float(0.125), C.c_float(p_drop), seed, layer, S())
    ok(L.uvit_op_attn_bwd(*args(0)))
    g = qf.grad
    scale = g.abs().max().item()
    close(dqkv, g, rtol=3e-2, atol=2e-2 * scale, what="dqkv")
    for part, name in enumerate("qkv"):                              # per-tensor relative L2 (a transposed tile would pass max-norm alone)
        a = dqkv.view(B * N, 3, Cd)[:, part].float(); r = g.view(B * N, 3, Cd)[:, part]
        assert ((a - r).norm() / r.norm()).item() < 1e-2, (name, ((a - r).norm() / r.norm()).item())
    close(delta, (d_o.float() * out.float()).view(B, N, H, 64).sum(-1).transpose(1, 2), rtol=1e-3, atol=1e-3, what="delta")
    dbias = slab[:, :N, :N].transpose(1, 2)             # slab is [h][key][q]
    close(dbias, bq.grad, rtol=3e-2, atol=2e-2 * bq.grad.abs().max().item(), what="dbias")
    assert ((dbias - bq.grad).norm() / bq.grad.norm()).item() < 1e-2
    assert slab[:, N:, :].abs().sum() == 0 and slab[:, :, N:].abs().sum() == 0
    ok(L.uvit_op_attn_bwd(*args(1)))            # accumulate flag adds on top
    close(slab[:, :N, :N].transpose(1, 2), 2 * bq.grad, rtol=3e-2, atol=4e-2 * bq.grad.abs().max().item(), what="dbias x2")
    # no bias, no bias gradient: same dqkv path without the dS stream
    ok(L.uvit_op_attn_fwd(P(qkv), P(None), P(out), P(lse), B, H, N, 208, C.c_float(0.125), C.c_float(p_drop), seed, layer, S()))
    qf2 = qkv.float().requires_grad_(True)
    ref2, _ = attn_ref(qf2, None, B, H, N, keep)
    ref2.backward(d_o.float().view(B, N, Cd))
    ok(L.uvit_op_attn_bwd(P(qkv), P(out), P(d_o), P(None), P(lse), P(delta), P(dqkv), P(None), 0, P(None), B, H, N, 208,
                                C.c_float(0.125), C.c_float(p_drop), seed, layer, S()))
    close(dqkv, qf2.grad, rtol=3e-2, atol=2e-2 * qf2.grad.abs().max().item(), what="dqkv (no bias)")


def test_attention_dropout_rate_and_determinism(L):
    B, H, N = 4, 12, 197
    qkv = bf(rnd(B * N, 3 * H * 64, seed=33))
    out1 = torch.zeros(B * N, H * 64, dtype=torch.bfloat16, device="cuda"); out2 = torch.zeros_like(out1)
    lse = torch.zeros(B, H, N, device="cuda")
    for o in (out1, out2):
        ok(L.uvit_op_attn_fwd(P(qkv), P(None), P(o), P(lse), B, H, N, 208, C.c_float(0.125), C.c_float(0.05), 7, 1, S()))
    assert torch.equal(out1, out2)
    from oracle.vit_oracle import attn_keep_mask
    keep = attn_keep_mask(7, 1, B, H, N, 0.05)
    assert abs((keep == 0).float().mean().item() - 0.05) < 2e-3


def test_relpos_gather_scatter(L):
    from uncertainty_vit_amd.modeling_cyclical import relative_position_index
    H, ws = 3, 3
    N = ws * ws + 1
    idx = relative_position_index(ws).cuda()
    table = rnd((2 * ws - 1) ** 2 + 3, H, seed=40)
    biasP = torch.full((H, 208, 208), 9.0, device="cuda")
    i32 = idx.to(torch.int32).contiguous()
    ok(L.uvit_op_relpos_gather(P(table), P(i32), P(biasP), H, N, 208, S()))
    ref = table[idx.view(-1)].view(N, N, H).permute(2, 0, 1) * LOG2E
    torch.testing.assert_close(biasP[:, :N, :N], ref)
    assert biasP[:, N:, :N].abs().sum() == 0 and torch.all(biasP[:, :, N:] == -1e30)
    slab = torch.zeros(2, H, 208, 208, device="cuda")
    dS = rnd(2, H, N, N, seed=41)                       # [slab][h][q][k]
    slab[:, :, :N, :N] = dS.transpose(2, 3)             # stored [key][q]
    dt = torch.zeros_like(table)
    ok(L.uvit_op_relpos_scatter(P(slab), 2, P(i32), P(dt), H, N, 208, S()))
    ref = torch.zeros_like(table)
    ref.index_add_(0, idx.view(-1), dS.sum(0).permute(1, 2, 0).reshape(N * N, H))
    torch.testing.assert_close(dt, ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,Cd", [(30, 128), (10, 192), (1000, 768), (77, 1024)])
def test_layernorm_fwd_bwd(L, M, Cd):
    x = rnd(M, Cd, seed=50) * 2 + 0.3
    w, b = rnd(Cd, seed=51) * 0.2 + 1, rnd(Cd, seed=52) * 0.1
    y = torch.zeros(M, Cd, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
    ok(L.uvit_op_ln_fwd(P(x), P(w), P(b), P(y), P(mean), P(rstd), M, Cd, C.c_float(1e-6), S()))
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (Cd,), wr, br, 1e-6)
    close(y, ref, rtol=1e-2, atol=1e-2, what="ln fwd")
    torch.testing.assert_close(mean, x.mean(-1), rtol=1e-5, atol=1e-5)
    dy = bf(rnd(M, Cd, seed=53)); dres = rnd(M, Cd, seed=54)
    ref.backward(dy.float())
    dx = torch.zeros(M, Cd, device="cuda"); dw = torch.zeros(Cd, device="cuda"); db = torch.zeros(Cd, device="cuda")
    ok(L.uvit_op_ln_bwd(P(dy), P(x), P(mean), P(rstd), P(w), P(dres), P(dx), P(dw), P(db), M, Cd, S()))
    torch.testing.assert_close(dx, xr.grad + dres, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dw, wr.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db, br.grad, rtol=1e-4, atol=1e-3)


def test_ema_adamw_sumsq_against_torch(L):
    n, n_decay = 64 * 1000, 64 * 600
    p = rnd(n, seed=60); g = rnd(n, seed=61) * 3
    e = rnd(n, seed=62)
    e0, p0 = e.clone(), p.clone()
    eb = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    ok(L.uvit_op_ema(P(e), P(p), P(eb), n, C.c_float(0.9998), S()))
    torch.testing.assert_close(e, 0.9998 * e0 + (1 - 0.9998) * p0, rtol=1e-6, atol=1e-7)   # engine_for_cyclical.py:183
    assert torch.equal(eb, e.to(torch.bfloat16))
    ss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ok(L.uvit_op_sumsq(P(g), n, P(ss), S()))
    assert abs(ss.item() - (g.double() ** 2).sum().item()) < 1e-6 * ss.item()
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [ref], "weight_decay": 0.0}], lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    pb = torch.zeros(n, dtype=torch.bfloat16, device="cuda"); gn = torch.zeros(1, device="cuda")
    refd = p0[:n_decay].clone().requires_grad_(True); refn = p0[n_decay:].clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [refd], "weight_decay": 0.05}, {"params": [refn], "weight_decay": 0.0}],
                            lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
    for step in range(1, 4):
        gg = g * step
        refd.grad, refn.grad = gg[:n_decay].clone(), gg[n_decay:].clone()
        norm = torch.nn.utils.clip_grad_norm_([refd, refn], 3.0)
        opt.step()
        ss.zero_()
        ok(L.uvit_op_sumsq(P(gg), n, P(ss), S()))
        ok(L.uvit_op_adamw(P(p), P(gg), P(m), P(v), P(pb), n, n_decay, C.c_float(2e-3), C.c_float(0.05), C.c_float(0.9),
                           C.c_float(0.999), C.c_float(1e-8), step, P(ss), C.c_float(3.0), C.c_float(1.0), P(gn), S()))
        assert abs(gn.item() - norm.item()) < 1e-4 * norm.item()
    torch.testing.assert_close(p, torch.cat([refd, refn]).detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(pb, p.to(torch.bfloat16))


@pytest.mark.parametrize("beta,l2", [(2.0, 0), (0.12, 0), (1.0, 1)])
def test_smooth_l1_fwd_bwd(L, beta, l2):
    Mmax, Cd, cnt = 64, 128, 50
    out = rnd(Mmax, Cd, seed=70) * 3; tgt = rnd(Mmax, Cd, seed=71)
    count = torch.tensor([cnt], dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda"); dout = torch.ones(Mmax, Cd, dtype=torch.bfloat16, device="cuda")
    ok(L.uvit_op_smooth_l1(P(out), P(tgt), P(count), C.c_float(beta), l2, C.c_float(1.0), P(loss), P(dout), Mmax, Cd, S()))
    o = out[:cnt].clone().requires_grad_(True)
    ref = F.mse_loss(o, tgt[:cnt]) if l2 else F.smooth_l1_loss(o, tgt[:cnt], beta=beta)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()) + 1e-7
    close(dout[:cnt], o.grad, rtol=1e-2, atol=1e-7, what="dout")
    assert dout[cnt:].abs().sum() == 0


def test_mask_compact_im2col_targets_droppath(L):
    from oracle import vit_oracle as vo
    B, g = 5, 14
    Pn = g * g
    mask = (torch.rand(B, Pn, generator=torch.Generator().manual_seed(80)) < 0.6).long().cuda()
    rowidx = torch.full((B * Pn,), -1, dtype=torch.int32, device="cuda"); count = torch.zeros(1, dtype=torch.int32, device="cuda")
    ok(L.uvit_op_mask_compact(P(mask), P(rowidx), P(count), B, Pn, S()))
    nz = mask.view(-1).nonzero().view(-1)
    ref_rows = (nz // Pn) * (Pn + 1) + 1 + nz % Pn
    assert count.item() == nz.numel()
    assert torch.equal(rowidx[: nz.numel()].long(), ref_rows)
    # empty and full masks
    for m in (torch.zeros_like(mask), torch.ones_like(mask)):
        ok(L.uvit_op_mask_compact(P(m), P(rowidx), P(count), B, Pn, S()))
        assert count.item() == int(m.sum())
    # im2col == conv patches
    img = rnd(2, 3, 48, 48, seed=81)
    cols = torch.zeros(2 * 9, 768, dtype=torch.bfloat16, device="cuda")
    ok(L.uvit_op_im2col(P(img), P(cols), 2, 3, 48, 16, S()))
    ref = img.reshape(2, 3, 3, 16, 3, 16).permute(0, 2, 4, 1, 3, 5).reshape(18, 768)
    assert torch.equal(cols, ref.to(torch.bfloat16))
    # target builder on masked rows: LN(no affine, 1e-5) per layer, mean, post LN
    ok(L.uvit_op_mask_compact(P(mask), P(rowidx), P(count), B, Pn, S()))
    Cd = 128
    layers = [rnd(B * (Pn + 1), Cd, seed=82 + i) * (i + 1) for i in range(3)]
    acc = torch.zeros(B * Pn, Cd, device="cuda")
    for i, x in enumerate(layers):
        ok(L.uvit_op_target_accum(P(x), P(rowidx), P(count), P(acc), 1 if i == 0 else 0, B * Pn, Cd, C.c_float(1e-5), S()))
    ok(L.uvit_op_target_finalize(P(acc), P(count), 3, 1, B * Pn, Cd, C.c_float(1e-5), S()))
    hp = vo.StepHParams(target_layers=(0, 1, 2))
    ref = vo.build_targets([x.view(B, Pn + 1, Cd)[:, 1:].cpu() for x in layers], mask.view(B, g, g).cpu(), hp)
    torch.testing.assert_close(acc[: ref.shape[0]].cpu(), ref, rtol=1e-4, atol=1e-4)
    # drop-path multipliers mirror the oracle's counter-based stream
    cfg = vo.VitConfig(depth=4, drop_path_rate=0.25)
    rates = torch.tensor(cfg.drop_path_rates(), device="cuda")
    sc = torch.zeros(4 * 2 * 7, device="cuda")
    ok(L.uvit_op_droppath(P(sc), P(rates), 4, 7, 99, 5, S()))
    p1, p2 = vo.drop_path_scales(99, 5, cfg, 7)
    sc = sc.view(4, 2, 7).cpu()
    for i in range(4):
        for br, ref in ((0, p1[i]), (1, p2[i])):
            torch.testing.assert_close(sc[i, br], torch.ones(7) if ref is None else ref, rtol=1e-6, atol=0)


def test_gemm_nt_row_split_tail_branch_two_stream_shapes(L):
    """The row-split tail of the auto dispatch at the shape that reaches it in the benchmarked two-stream bs=128 step
    (stacked mean + covariance rows, M = 2 x 25216; N = 3072, K = 768: 2364 tiles of 256x256 = 9 rounds of 256 CUs + 60, so
    the last 1280 rows go to the 128x128 kernel with out / out2 / aux / resid / row0 moved by hand): GELU_DG (out + out2),
    MULAUX (aux) and RESID (resid + per-sample drop-path through row0).  The launcher reports the tail rows; the test
    asserts the branch was really taken."""
    tokens, B = 197, 256
    M, N, K = B * tokens, 3072, 768
    x = bf(rnd(M, K, seed=1)); w = bf(rnd(N, K, scale=0.05, seed=2)); b = rnd(N, seed=3)
    a_out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); dg = torch.zeros_like(a_out)
    tail = C.c_int(0)
    ok(nt(L, 8, P(x), P(w), M, N, K, K, K, C.byref(epi(out=a_out, out2=dg, bias=b, ldo=N)), S(), tail=tail))
    assert tail.value > 0 and tail.value % 256 == 0, f"row-split tail not taken (tail rows {tail.value}); CU count != 256?"
    t0 = M - tail.value
    for lo, hi in ((0, 4096), (t0 - 2048, t0 + 2048), (M - 2048, M)):      # head, the seam, the end of the tail
        h = (x[lo:hi].float() @ w.float().t() + b).to(torch.bfloat16).float().requires_grad_(True)
        F.gelu(h).sum().backward()
        close(a_out[lo:hi], F.gelu(h.detach()), rtol=1e-2, atol=1e-2, what=f"gelu rows {lo}:{hi}")
        close(dg[lo:hi], h.grad, rtol=1e-2, atol=1e-2, what=f"gelu' rows {lo}:{hi}")
    # MULAUX: dH = (dY @ W2) * aux
    dy = bf(rnd(M, K, scale=0.1, seed=4)); dh = torch.zeros_like(a_out)
    tail2 = C.c_int(0)
    ok(nt(L, 9, P(dy), P(w), M, N, K, K, K, C.byref(epi(out=dh, aux=dg, ldo=N)), S(), tail=tail2))
    assert tail2.value == tail.value
    for lo, hi in ((t0 - 1024, t0 + 1024), (M - 1024, M)):
        close(dh[lo:hi], (dy[lo:hi].float() @ w.float().t()) * dg[lo:hi].float(), rtol=2e-2, atol=8e-3, what=f"mulaux rows {lo}:{hi}")
    del dh, dg
    # RESID: x + dp[sample] * gamma * (acc + bias), the sample index of the tail rows comes through row0
    gam, res = rnd(N, scale=0.1, seed=5), rnd(M, N, seed=6)
    dp = ((torch.arange(B, device="cuda") % 4).float() * 0.5)
    xo = torch.zeros(M, N, device="cuda"); branch = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    tail3 = C.c_int(0)
    ok(nt(L, 3, P(x), P(w), M, N, K, K, K,
          C.byref(epi(out=xo, out2=branch, bias=b, gamma=gam, resid=res, rowscale=dp, ldo=N, tokens=tokens)), S(), tail=tail3))
    assert tail3.value == tail.value
    for lo, hi in ((t0 - 1024, t0 + 1024), (M - 1024, M)):
        y = x[lo:hi].float() @ w.float().t() + b
        close(branch[lo:hi], y, what=f"resid branch rows {lo}:{hi}")
        scale = dp[torch.arange(lo, hi, device="cuda") // tokens][:, None]
        close(xo[lo:hi], res[lo:hi] + scale * gam * y, rtol=5e-3, atol=5e-3, what=f"resid out rows {lo}:{hi}")


def test_gemm_two_host_threads_two_streams_different_tuning(L):
    """include/uvit.h: no mutable process-wide launcher state.  Two host threads launch NT GEMMs on their own HIP streams
    with DIFFERENT tuning (one forces the 128x128 kernel, one the auto dispatch with its row-split tail) at the same time;
    every result must equal the single-threaded result of the same tuning bit for bit."""
    import threading
    from uncertainty_vit_amd.native import Tuning
    M, N, K = 197 * 128, 768, 2048                       # auto dispatch: 320-row tiles; variant 1: 256-row tiles + tail
    a, w, b = bf(rnd(M, K, seed=1)), bf(rnd(N, K, scale=0.05, seed=2)), rnd(N, seed=3)
    tunes = [Tuning.default(nt_variant=0), Tuning.default(nt_variant=1)]

    def run(tu, out, stream):
        return L.uvit_op_gemm_nt_tuned(4, P(a), P(w), M, N, K, K, K, C.byref(epi(out=out, bias=b, ldo=N)), C.byref(tu), None,
                                       C.c_void_p(stream.cuda_stream))
    refs = []
    for tu in tunes:
        o = torch.zeros(M, N, device="cuda")
        ok(run(tu, o, torch.cuda.current_stream()))
        refs.append(o)
    torch.cuda.synchronize()
    close(refs[0], a.float() @ w.float().t() + b, rtol=2e-3, atol=2e-3, what="variant 0")
    close(refs[1], refs[0], rtol=1e-4, atol=1e-4, what="variant 1 vs 0")
    errs = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            outs = [torch.zeros(M, N, device="cuda") for _ in range(4)]
            torch.cuda.synchronize()
            for it in range(40):
                rc = run(tunes[i], outs[it % 4], st)
                if rc != 0:
                    errs.append((i, it, rc)); return
            st.synchronize()
            for o in outs:
                if not torch.equal(o, refs[i]):
                    errs.append((i, "mismatch", float((o - refs[i]).abs().max())))
        except Exception as ex:      # noqa: BLE001
            errs.append((i, repr(ex)))
    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("Mrows,count,Cd", [(40, 40, 128), (300, 257, 768), (15360, 15360, 768)])
def test_wasserstein_loss_operator(L, Mrows, count, Cd):
    """WassersteinLoss (distloss.py:13-30 with wasserstein_distance :73-79) as an operator: value and the gradients with
    respect to the mean and covariance outputs, including the NON-LOCAL gradient of the arg-max row (`pos / pos.abs().max()`
    routes -sum_s g'(u_s) u_s / m to it), against an fp32 autograd restatement on the same inputs.  Rows >= count are padding."""
    lam, ls = 1e-2, 1.0
    om, oc = rnd(Mrows, Cd, seed=1), rnd(Mrows, Cd, seed=2)
    tm, tc = rnd(Mrows, Cd, seed=3), rnd(Mrows, Cd, seed=4)
    om[min(7, count - 1)] *= 3.0                         # a clear arg-max row that is not row 0
    cnt = torch.tensor([count], dtype=torch.int32, device="cuda")
    base = torch.zeros(Mrows, Cd, dtype=torch.bfloat16, device="cuda")   # the SmoothL1 gradient already in dout_m: the
    base[1, 5] = 1.0                                                         # operator ADDS to it (checked on one element)
    dm = base.clone(); dc = torch.full((Mrows, Cd), 9.0, dtype=torch.bfloat16, device="cuda")
    scratch = torch.zeros(16 + Mrows, device="cuda"); loss = torch.full((1,), 0.25, device="cuda")
    ok(L.uvit_op_wasserstein_loss(P(om), P(oc), P(tm), P(tc), P(cnt), lam, ls, P(scratch), P(loss), P(dm), P(dc), Mrows, Cd, S()))

    a, c = om[:count].clone().requires_grad_(True), oc[:count].clone().requires_grad_(True)
    mo, co, mt, ct = torch.sigmoid(a), torch.sigmoid(c), torch.sigmoid(tm[:count]), torch.sigmoid(tc[:count])
    pos = ((mo - mt) ** 2).sum(-1) + ((torch.sqrt(co.clamp(min=1e-24)) - torch.sqrt(ct.clamp(min=1e-24))) ** 2).sum(-1)
    pos = pos / pos.abs().max()
    lo = -torch.log(torch.sigmoid(-pos + 1e-24))
    lo = lo / lo.abs().max()
    ref = lo.sum() * lam
    ref.backward()
    assert loss.item() - 0.25 == pytest.approx(ref.item(), rel=2e-4)
    assert dm[1, 5].item() == pytest.approx(1.0, abs=8e-3)
    dm[1, 5] = dm[1, 4]; a.grad[1, 5] = a.grad[1, 4]     # exclude the marker from the precision check
    gm = dm[:count].float()
    amax = int(pos.argmax())
    gmax = a.grad.abs().max().item()
    close(gm, a.grad, rtol=1e-2, atol=1e-2 * gmax, what="d mean_out")
    close(dc[:count], c.grad, rtol=1e-2, atol=1e-2 * c.grad.abs().max().item(), what="d cov_out")
    rel = (gm[amax] - a.grad[amax]).norm() / a.grad[amax].norm()
    assert rel < 2e-2, f"arg-max row {amax}: relative L2 error {rel}"
    others = torch.arange(count, device="cuda") != amax
    rel_o = (gm[others] - a.grad[others]).norm() / a.grad[others].norm()
    assert rel_o < 2e-2, f"other rows: relative L2 error {rel_o}"
    if count < Mrows:
        assert torch.all(dc[count:] == 0) and torch.all(dm[count:] == 0)


def test_synth_batch_on_device(L):
    """On-device batch synthesis: masks with EXACTLY n ones per image, bit-exact against a numpy
    restatement of the kernel's counter-based ranking; images ~ N(0, 1); everything a pure function of (seed, it)."""
    B, Pn, n_mask, S_ = 16, 196, 120, 224
    img = torch.zeros(B, 3, S_, S_, device="cuda"); mask = torch.full((B, Pn), 7, dtype=torch.int64, device="cuda")
    ok(L.uvit_op_synth_batch(P(img), P(mask), B, 3, S_, Pn, n_mask, 1234, 5, S()))
    m = mask.cpu().numpy()
    assert ((m == 0) | (m == 1)).all() and (m.sum(1) == n_mask).all()
    from oracle.closed_form import _mix32
    with np.errstate(over="ignore"):
        key = _mix32(np.uint32(1234) ^ np.uint32((5 * 0x9E3779B9 + 0x51ED270B) & 0xFFFFFFFF))
        keys = _mix32(np.arange(B * Pn, dtype=np.uint32) ^ (key ^ np.uint32(0xA5A5A5A5))).reshape(B, Pn)
    order = np.argsort(keys, axis=1, kind="stable")
    ref = np.zeros((B, Pn), dtype=np.int64)
    np.put_along_axis(ref, order[:, :n_mask], 1, axis=1)
    assert np.array_equal(m, ref)
    x = img.float()
    assert abs(x.mean().item()) < 5e-3 and abs(x.std().item() - 1.0) < 5e-3
    assert abs((x ** 3).mean().item()) < 2e-2 and abs((x ** 4).mean().item() - 3.0) < 5e-2          # skewness 0, kurtosis 3
    assert abs(x.view(B, -1)[:, :-1].mul(x.view(B, -1)[:, 1:]).mean().item()) < 5e-3                  # neighbours uncorrelated
    img2 = torch.zeros_like(img); mask2 = torch.zeros_like(mask)
    ok(L.uvit_op_synth_batch(P(img2), P(mask2), B, 3, S_, Pn, n_mask, 1234, 5, S()))
    assert torch.equal(img, img2) and torch.equal(mask, mask2)
    ok(L.uvit_op_synth_batch(P(img2), P(mask2), B, 3, S_, Pn, n_mask, 1234, 6, S()))
    assert not torch.equal(img, img2) and not torch.equal(mask, mask2)
