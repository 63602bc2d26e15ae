"""Average a rocprofv3 --pmc counter per dispatch of kernels whose name contains a pattern.

usage: python tools/pmc_summary.py <dir> <pattern> [--alternate K [labels,comma,separated]]
  --alternate K: the matching dispatches are dealt round-robin, in dispatch order, into K groups -- for ONE template instantiation
                 that the step launches with K shapes in a fixed rotation (gemm_nt256_kernel<3, 5>: proj, fc2, proj, fc2, ... in a
                 single-stream run), so that bytes are reported per shape and not as a mean over shapes.
"""
import csv
import glob
import sys


def main():
    d, pat = sys.argv[1], sys.argv[2]
    k, labels = 1, None
    if "--alternate" in sys.argv:
        i = sys.argv.index("--alternate")
        k = int(sys.argv[i + 1])
        labels = sys.argv[i + 2].split(",") if len(sys.argv) > i + 2 else None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        by = {}
        seen = {}
        for r in rows:
            c = r["Counter_Name"]
            j = seen.get(c, 0)
            seen[c] = j + 1
            by.setdefault((c, j % k), []).append(float(r["Counter_Value"]))
        for (c, g), v in sorted(by.items()):
            tag = f" [{labels[g] if labels and g < len(labels) else g}]" if k > 1 else ""
            print(f"{f}: {c}: {len(v)} dispatches of '{pat}'{tag}, mean {sum(v) / len(v):.1f}, min {min(v):.1f}, max {max(v):.1f}")


if __name__ == "__main__":
    main()
