"""Idle time in a rocprofv3 kernel trace: union of kernel intervals vs wall span, per step window (adamw marks a step end)."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=lambda p: -len(open(p).read()))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [e for s, e, n, q in rows if n.startswith("adamw_kernel")]
print(f"{f}: {len(rows)} dispatches, {len(ends)} steps")
for a, b in zip(ends[:-1], ends[1:]):
    win = [(s, e, n, q) for s, e, n, q in rows if s >= a and e <= b + 1]
    busy = 0; cur_s, cur_e = None, None
    for s, e, n, q in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = b - a
    ksum = sum(e - s for s, e, n, q in win)
    # gaps histogram
    gaps = []
    cur_e = None
    for s, e, n, q in win:
        if cur_e is not None and s > cur_e: gaps.append((s - cur_e, n))
        cur_e = e if cur_e is None else max(cur_e, e)
    gaps.sort(reverse=True)
    print(f"step span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms  idle {100*(span-busy)/span:.1f}%  kernel-sum {ksum/1e6:.2f} ms  n={len(win)}  queues={len(set(q for *_, q in win))}")
    print("   largest gaps (us, before kernel):", [(round(g/1e3, 1), n[:28]) for g, n in gaps[:6]], " total gap count", len(gaps), "median", (sorted(g for g, _ in gaps)[len(gaps)//2] / 1e3 if gaps else 0))
