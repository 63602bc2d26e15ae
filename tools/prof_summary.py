"""Summarise a rocprofv3 --kernel-trace --stats output directory (per-kernel time per step)."""
import csv
import glob
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # bench.py --steps 3 --warmup 1
f = glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step")
print(f"{'kernel':72s} {'calls/step':>10s} {'ms/step':>9s} {'avg us':>9s} {'%':>6s}")
for r in rows[:32]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']) / steps:10.1f} {float(r['TotalDurationNs']) / 1e6 / steps:9.3f} "
          f"{float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
