"""Per-kernel budget table: measured time per launch against the floor max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s).

Reads a single-stream kernel-stats summary written by tools/prof_summary.py and prints a markdown table: calls per step, ms per step,
average us per launch, algorithmic GFLOP and MB per launch, the floor in us and time / floor.  The algorithmic figures are those of the
ViT-B/16 bs = 128 step (M = 25216 token rows, C = 768, hidden 3072, 12 heads, N = 197): operands read once, results written once.
Kernel names that cover several shapes (one template instantiation launched with different K) get the launch-weighted mean.

usage: python tools/kernel_budget.py profiles/round4_step_kernel_stats_singlestream.txt [--two-stream]
"""
import sys

PEAK_F, PEAK_B = 2.5e15, 8.0e12


def shapes(two_stream=False):
    B, N, C, Hd, H = 128, 197, 768, 3072, 12
    M = B * N
    S = 2 if two_stream else 1
    Ms = S * M                      # stacked rows of the ops whose weights the two streams share
    bf, f4 = 2, 4
    gemm = lambda m, n, k: 2.0 * m * n * k  # noqa: E731
    att = 4.0 * B * H * N * N * 64          # QK^T + PV of one forward
    k = {}
    # name prefix -> (GFLOP, MB) per launch
    k["void gemm_nt256_kernel<1, 4, true>"] = (gemm(M, 3 * C, C), bf * (M * C + 3 * C * C + M * 3 * C))                      # QKV (+ ELU twin: <7, 4>)
    k["void gemm_nt256_kernel<7, 4, true>"] = k["void gemm_nt256_kernel<1, 4, true>"]
    k["void gemm_nt256_kernel<2, 4, true>"] = (gemm(Ms, Hd, C), bf * (Ms * C + Hd * C + Ms * Hd))                               # fc1 teacher
    k["void gemm_nt256_kernel<8, 4, true>"] = (gemm(Ms, Hd, C), bf * (Ms * C + Hd * C + 2 * Ms * Hd))                           # fc1 student
    k["void gemm_nt256_kernel<9, 4, false>"] = (gemm(Ms, Hd, C), bf * (Ms * C + Hd * C + 2 * Ms * Hd))                          # dH = (dY W2) gelu'(h)
    proj = (gemm(M, C, C), bf * (M * C + C * C) + 2 * f4 * M * C)
    fc2 = (gemm(M, C, Hd), bf * (M * Hd + C * Hd) + 2 * f4 * M * C)
    k["void gemm_nt256_kernel<3, 5, false>"] = tuple((a + b) / 2 for a, b in zip(proj, fc2))                                    # proj + fc2 (residual epilogue)
    d_fc1 = (gemm(Ms, C, Hd), bf * (Ms * Hd + C * Hd + Ms * C))
    d_proj = (gemm(M, C, C), bf * (2 * M * C + C * C))
    d_qkv = (gemm(Ms, C, 3 * C), bf * (Ms * 3 * C + 3 * C * C + Ms * C))
    if two_stream:      # stacked launches: fc1, qkv dgrads once over 2 M rows, proj dgrad per stream
        k["void gemm_nt256_kernel<0, 5, false>"] = tuple((d_fc1[i] + 2 * d_proj[i] + d_qkv[i]) / 4 for i in range(2))
    else:
        k["void gemm_nt256_kernel<0, 5, false>"] = tuple((d_fc1[i] + d_proj[i] + d_qkv[i]) / 3 for i in range(2))
    wg_f = gemm(Ms, C, Hd) * 2 + S * gemm(M, C, C) + gemm(Ms, 3 * C, C)
    wg_b = bf * (Ms * (C + Hd) * 2 + S * M * 2 * C + Ms * 4 * C) + f4 * (2 * C * Hd + S * C * C + 3 * C * C)
    k["gemm_tn256_group_kernel"] = (wg_f, wg_b)
    k["void attn_fwd_kernel"] = (att, bf * (M * 3 * C + M * C) + f4 * H * 208 * 208)
    k["void attn_bwd_fused_kernel"] = (2.5 * att, bf * (M * 3 * C * 2 + 2 * M * C) + bf * B * H * 208 * 224)                   # + the bf16 dS stream
    k["attn_dbias_reduce_kernel"] = (0.0, bf * B * H * 208 * 224 + f4 * H * 208 * 208)
    k["void attn2_fwd_kernel"] = (2 * att, bf * (2 * M * 3 * C + 2 * M * C) + f4 * H * 208 * 208)
    k["void attn2_bwd_fused_kernel"] = (5 * att, bf * (2 * M * 3 * C * 2 + 4 * M * C) + bf * B * H * 208 * 208)
    k["_Z22attn2_bwd_fused_kernel"] = k["void attn2_bwd_fused_kernel"]            # (rocprofv3 prints some names mangled)
    k["attn2_dbias_reduce_kernel"] = (0.0, bf * B * H * 208 * 208 + f4 * H * 208 * 208)
    k["_Z13ln_fwd_kernel"] = (0.0, Ms * C * (f4 + bf))
    k["_Z13ln_bwd_kernelILi3ELb1E"] = (0.0, M * C * (bf + f4 + f4 + f4 + bf + bf))                                              # dy, x, dres -> dx; y_next -> dy_next
    k["_Z13ln_bwd_kernelILi3ELb0E"] = (0.0, Ms * C * (bf + f4 + f4 + f4))
    n_par = 115_778_640 if two_stream else 86_256_720
    k["adamw_kernel"] = (0.0, n_par * (5 * f4 * 2 - 2 * f4 + 2 * bf))                                                           # p g m v e read, p m v e written, 2 shadows
    return k


def main():
    path = sys.argv[1]
    two = "--two-stream" in sys.argv
    tab = shapes(two)
    rows = []
    for line in open(path):
        if line.startswith("#") or line.startswith("kernel"):
            continue
        parts = line.rstrip().split()
        if len(parts) < 5:
            continue
        try:
            calls, ms, avg = float(parts[-4]), float(parts[-3]), float(parts[-2])
        except ValueError:
            continue
        name = line[:72].strip()
        key = next((p for p in tab if name.startswith(p)), None)
        rows.append((name, calls, ms, avg, tab.get(key)))
    print("| kernel | calls/step | ms/step | avg us | GFLOP | MB | floor us | bound | time / floor |")
    print("|---|---:|---:|---:|---:|---:|---:|---|---:|")
    tot = sum(r[2] for r in rows)
    covered = 0.0
    for name, calls, ms, avg, fb in rows:
        if fb is None or ms / tot < 0.003:
            continue
        covered += ms
        tf, tb = fb[0] / PEAK_F * 1e6, fb[1] / PEAK_B * 1e6
        floor = max(tf, tb)
        print(f"| `{name[:58]}` | {calls:.1f} | {ms:.3f} | {avg:.1f} | {fb[0] / 1e9:.1f} | {fb[1] / 1e6:.0f} | {floor:.1f} | {'mfma' if tf >= tb else 'hbm'} | {avg / floor:.2f} |")
    print(f"\n{covered:.2f} of {tot:.2f} ms/step of kernel time in the table (kernels under 0.3 % of the step are left out)")


if __name__ == "__main__":
    main()
