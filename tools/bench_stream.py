"""Diagnostic: chip-wide streaming rates (pure store, pure load, copy) at the sizes of the GEMM epilogues (run on the GPU box)."""
import torch


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
    for mb in (77, 155, 310, 620, 2480):
        n = mb * 1000 * 1000 // 2
        x = torch.empty(n, device="cuda", dtype=torch.bfloat16)
        y = torch.randn(n // 4, device="cuda").to(torch.bfloat16).repeat(4)[:n].contiguous()
        t_fill = timeit(lambda: x.fill_(1.0))
        t_copy = timeit(lambda: x.copy_(y))
        t_sum = timeit(lambda: y.view(torch.int16).sum())
        print(f"{mb:5d} MB: fill {t_fill:7.1f} us = {mb / t_fill:5.2f} TB/s   copy {t_copy:7.1f} us = {2 * mb / t_copy:5.2f} TB/s (r+w)   "
              f"reduce {t_sum:7.1f} us = {mb / t_sum:5.2f} TB/s", flush=True)
