"""Average a rocprofv3 --pmc counter per dispatch of kernels whose name contains a pattern."""
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        if pat in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        print(f"{f}: {k}: {len(v)} dispatches of '{pat}', mean {sum(v) / len(v):.1f}, min {min(v):.1f}, max {max(v):.1f}")
