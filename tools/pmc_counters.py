#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc ... --kernel-trace --output-format csv run.

    python tools/pmc_counters.py <output dir> [kernel-name substrings ...]

For every kernel (optionally only those whose name contains one of the substrings) prints the number of dispatches and,
per counter, the median and the sum over dispatches, together with the median kernel duration from the kernel trace of
the same run (ns).  Used for profiles/r2_*pmc*.txt."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    subs = sys.argv[2:]
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    grid = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                if subs and not any(s in name for s in subs):
                    continue
                per[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                grid[name] = (row.get("Grid_Size"), row.get("Workgroup_Size"), row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("Scratch_Size"))
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                if subs and not any(s in name for s in subs):
                    continue
                dur[name].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    for name in sorted(per):
        c = per[name]
        n = max(len(v) for v in c.values())
        dd = sorted(dur.get(name, [0]))
        print("%s  dispatches %d  grid/wg/vgpr/agpr/sgpr/scratch %s  median_ns %.0f" % (name, n, "/".join(str(x) for x in grid[name]), dd[len(dd) // 2]))
        for k in sorted(c):
            v = sorted(c[k])
            print("    %-24s median %.6g  sum %.6g  min %.6g  max %.6g" % (k, v[len(v) // 2], sum(v), v[0], v[-1]))


if __name__ == "__main__":
    main()
