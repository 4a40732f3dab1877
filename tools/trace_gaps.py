#!/usr/bin/env python3
"""Steady-state view of the bucket accumulations in a rocprofv3 --kernel-trace database of a pipelined bench run: per proof the
four k_msm_accumulate launches (start, duration), the idle gaps between consecutive accumulations and the busiest other kernels.
Usage: tools/trace_gaps.py <results.db> [first_proof last_proof]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select k.start, k.end, s.kernel_name, k.stream_id from %s k join %s s on k.kernel_id = s.id order by k.start" % (kd, ks)).fetchall()


def short(n):
    m = re.search(r"_ZN2zk\d+([a-z_0-9]+?)(I|E)", n)
    return (m.group(1) if m else n[:30]) + ("<Fq2>" if "Fq2" in n else "")


acc = [(s, e, short(n)) for s, e, n, _ in rows if "k_msm_accumulate" in n]
per = 4
nproofs = len(acc) // per
lo = int(sys.argv[2]) if len(sys.argv) > 2 else nproofs // 2
hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(lo + 6, nproofs - 1)
t0 = acc[lo * per][0]
tot_gap = tot_acc = 0.0
for p in range(lo, hi):
    line = []
    for k in range(per):
        s, e, n = acc[p * per + k]
        prev_e = acc[p * per + k - 1][1]
        gap = (s - prev_e) / 1e3
        tot_gap += max(gap, 0)
        tot_acc += (e - s) / 1e3
        line.append("%s%s @%.0f %.0fus (gap %.0f)" % ("G2" if "Fq2" in n else "G1", "", (s - t0) / 1e3, (e - s) / 1e3, gap))
    print("proof %d: " % p + " | ".join(line))
span = (acc[hi * per][0] - acc[lo * per][0]) / 1e3 / (hi - lo)
print("period %.0f us/proof: accumulations %.0f us, gaps between them %.0f us" % (span, tot_acc / (hi - lo), tot_gap / (hi - lo)))
busy = collections.Counter()
w0, w1 = acc[lo * per][0], acc[hi * per][0]
for s, e, n, _ in rows:
    if s >= w0 and e <= w1 and "k_msm_accumulate" not in n:
        busy[short(n)] += (e - s) / 1e3
for n, t in busy.most_common(14):
    print("  %-28s %8.0f us/proof (sum of durations, overlapping)" % (n, t / (hi - lo)))
