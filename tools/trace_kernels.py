#!/usr/bin/env python3
"""Per (kernel, grid size) statistics of a rocprofv3 --kernel-trace CSV: launches, median and total duration.
The grid size tells the launches of one kernel for the different inner products apart.
    python tools/trace_kernels.py <kernel_trace.csv> [min_total_us]"""
import collections
import csv
import re
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("zk::", "")
    g2 = "Fq2" in n
    return re.sub(r"<.*", "", n) + ("<G2>" if g2 else "")


rows = collections.defaultdict(list)
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        rows[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("%-30s %10s %6s %10s %12s" % ("kernel", "grid", "calls", "median_us", "total_ms"))
for (n, g), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    if sum(v) < (float(sys.argv[2]) if len(sys.argv) > 2 else 200.0):
        continue
    print("%-30s %10s %6d %10.1f %12.3f" % (n, g, len(v), v[len(v) // 2], sum(v) / 1e3))
