"""Vendor reference for the attention core (torch SDPA on ROCm): B=128, H=12, N=197, d=64, additive bias, dropout 0.05."""
import torch, torch.nn.functional as F
B, H, N, D = 128, 12, 197, 64
q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
bias = torch.randn(1, H, N, N, device="cuda", dtype=torch.bfloat16)
do = torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for name, kw in (("bias+dropout", dict(attn_mask=bias, dropout_p=0.05)), ("bias", dict(attn_mask=bias)), ("plain", dict())):
    try:
        fwd = t(lambda: F.scaled_dot_product_attention(q, k, v, **kw))
        def fb():
            o = F.scaled_dot_product_attention(q, k, v, **kw)
            o.backward(do)
        tot = t(fb)
        print(f"SDPA {name:13s}: fwd {fwd:7.1f} us   fwd+bwd {tot:7.1f} us  (bwd ~ {tot - fwd:7.1f})")
    except Exception as e:
        print(name, "failed:", str(e)[:200])
