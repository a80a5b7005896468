#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) from a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
by_grid = "--by-grid" in sys.argv          # split each kernel by its grid size (one GEMM instance serves several shapes)
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + root)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("kernel_name")
            if by_grid:
                wg = max(1, int(row.get("Workgroup_Size_X") or 1))
                name = f"[{int(row.get('Grid_Size_X') or 0) // wg:>6d} x{row.get('Grid_Size_Z') or 1} wg] " + name
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            a = agg[name]
            a[0] += 1
            a[1] += dur
total = sum(v[1] for v in agg.values())
print(f"# {sum(v[0] for v in agg.values())} dispatches, {total / 1e6:.3f} ms of kernel time; source: {files}")
print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'share':>7}  kernel")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:7d} {t / 1e6:10.3f} {t / n / 1e3:10.2f} {100 * t / total:6.2f}%  {name[:150]}")
