#!/usr/bin/env python3
"""Timeline of the last proof in a rocprofv3 --kernel-trace database (rocpd sqlite): start (us from the proof's first kernel),
duration, stream, grid, kernel.  Usage: tools/trace_one_proof.py <results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select k.start, k.end, s.kernel_name, k.stream_id, k.grid_size_x, k.workgroup_size_x from %s k join %s s "
                   "on k.kernel_id = s.id order by k.start" % (kd, ks)).fetchall()
idx = [i for i, r in enumerate(rows) if "k_assembleEPK" in r[2]]
seg = rows[idx[-2] + 1: idx[-1] + 1]
t0 = seg[0][0]
for r in seg:
    m = re.search(r"_ZN2zk\d+([a-z_0-9]+?)(I|E)", r[2])
    name = m.group(1) if m else r[2][:40]
    if "Fq2" in r[2]:
        name += "<Fq2>"
    print("%8.1f %7.1f s%-3s %6d x%-4d %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4] // max(r[5], 1), r[5], name))
print("proof: %.1f us from first kernel to end of k_assemble; %d kernels" % ((seg[-1][1] - t0) / 1e3, len(seg)))
