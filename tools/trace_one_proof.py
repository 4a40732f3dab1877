#!/usr/bin/env python3
"""Timeline of the last proof in a rocprofv3 --kernel-trace CSV: start (us from the proof's first kernel), duration, queue, workgroups x
lanes, kernel.  Usage: tools/trace_one_proof.py <kernel_trace.csv>"""
import csv
import re
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("zk::", "")
    g2 = "Fq2" in n
    return re.sub(r"<.*", "", n) + ("<Fq2>" if g2 else "")


rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        wg = int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)
        grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), grid // max(wg, 1), wg))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2] == "k_assemble"]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0 = seg[0][0]
for r in seg:
    print("%8.1f %7.1f q%-3s %6d x%-4d %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4], r[5], r[2]))
print("proof: %.1f us from first kernel to end of k_assemble; %d kernels" % ((seg[-1][1] - t0) / 1e3, len(seg)))
