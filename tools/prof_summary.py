"""Summarise a rocprofv3 --kernel-trace output directory: per-kernel time per STEP, from the kernel trace, over the step windows only.

A step window is the span between two consecutive ends of `adamw_kernel` (the last kernel of a step), so set-up work (workspace
memset, first-touch casts, synthetic batch) and anything before the first / after the last step end is reported separately and never
divided into "per step" (the old summary divided whole-process totals by the step count).

usage: python tools/prof_summary.py <dir> [first_window [n_windows]]
       windows are numbered from 0 (= between the 1st and 2nd adamw end); default: all of them.
"""
import collections
import csv
import glob
import sys


def short(name):
    return name if len(name) <= 72 else name[:72]


def main():
    d = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    files = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=lambda p: -len(open(p).read()))
    if not files:
        sys.exit("no *_kernel_trace.csv under " + d)
    f = files[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    ends = [e for s, e, n in rows if n.startswith("adamw_kernel")]
    if len(ends) < 2:
        sys.exit(f"{f}: fewer than two adamw_kernel dispatches: no step window")
    nwin = int(sys.argv[3]) if len(sys.argv) > 3 else len(ends) - 1 - first
    lo, hi = ends[first], ends[first + nwin]
    inside = [(s, e, n) for s, e, n in rows if s >= lo and e <= hi]
    outside = [(s, e, n) for s, e, n in rows if not (s >= lo and e <= hi)]
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n in inside:
        by[n][0] += 1
        by[n][1] += e - s
    tot = sum(v[1] for v in by.values())
    print(f"# {f}")
    print(f"# {nwin} step windows (adamw end -> adamw end), wall {(hi - lo) / 1e6 / nwin:.2f} ms/step, kernel time {tot / 1e6 / nwin:.2f} ms/step "
          f"({len(inside)} dispatches inside, {len(outside)} outside)")
    print(f"{'kernel':72s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:36]:
        print(f"{short(n):72s} {c / nwin:10.1f} {t / 1e6 / nwin:9.3f} {t / 1e3 / c:9.1f} {100.0 * t / tot:6.2f}")
    ob = collections.defaultdict(lambda: [0, 0])
    for s, e, n in outside:
        ob[n][0] += 1
        ob[n][1] += e - s
    big = sorted(ob.items(), key=lambda kv: -kv[1][1])[:6]
    print("# outside the step windows (set-up, warm-up, tail), largest: " +
          "; ".join(f"{short(n)[:40]} x{c} {t / 1e3:.0f} us" for n, (c, t) in big))


if __name__ == "__main__":
    main()
