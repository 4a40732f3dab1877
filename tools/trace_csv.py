#!/usr/bin/env python3
"""Steady-state view of a pipelined bench run from a rocprofv3 --kernel-trace CSV.

    python tools/trace_csv.py <kernel_trace.csv> [first_proof count] [--full]

Per proof: the bucket accumulations (start, duration, idle gap in front of each), then the period, the sum of accumulation time and
of the gaps, and the busiest other kernels.  --full also lists every kernel of the window with its queue, so that one can see what
ran inside a gap and which queue it came from."""
import collections
import csv
import re
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("zk::", "")
    g2 = "Fq2" in n
    n = re.sub(r"<.*", "", n)
    return n + ("<G2>" if g2 else "")


def main():
    path = sys.argv[1]
    full = "--full" in sys.argv
    nums = [a for a in sys.argv[2:] if not a.startswith("--")]
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    acc = [r for r in rows if r[2].startswith("k_msm_accumulate")]
    # accumulation launches per proof: one of them is the G2 product (round 5: 3 -- B in G2, A, L + H merged; before: 4)
    g2n = sum(1 for r in acc if "G2" in r[2])
    per = max(1, round(len(acc) / g2n)) if g2n else 4
    nproofs = len(acc) // per
    lo = int(nums[0]) if nums else nproofs // 2
    cnt = int(nums[1]) if len(nums) > 1 else 8
    hi = min(lo + cnt, nproofs - 1)
    t0 = acc[lo * per][0]
    tot_gap = tot_acc = 0.0
    gaps = []
    for p in range(lo, hi):
        line = []
        for k in range(per):
            s, e, n, q, st = acc[p * per + k]
            prev_e = acc[p * per + k - 1][1]
            gap = (s - prev_e) / 1e3
            tot_gap += max(gap, 0)
            tot_acc += (e - s) / 1e3
            gaps.append((prev_e, s))
            line.append("%s @%.0f %.0fus (gap %.0f)" % ("G2" if "G2" in n else "G1", (s - t0) / 1e3, (e - s) / 1e3, gap))
        print("proof %d: " % p + " | ".join(line))
    n = hi - lo
    span = (acc[hi * per][0] - acc[lo * per][0]) / 1e3 / n
    print("period %.0f us/proof: accumulations %.0f us, gaps between them %.0f us" % (span, tot_acc / n, tot_gap / n))
    busy = collections.Counter()
    calls = collections.Counter()
    w0, w1 = acc[lo * per][0], acc[hi * per][0]
    for s, e, nm, q, st in rows:
        if s >= w0 and e <= w1 and not nm.startswith("k_msm_accumulate"):
            busy[nm] += (e - s) / 1e3
            calls[nm] += 1
    for nm, t in busy.most_common(40):
        print("  %-34s %8.0f us/proof over %5.1f launches/proof (sum of durations, overlapping)" % (nm, t / n, calls[nm] / n))
    # what ran inside the gaps in front of the accumulations (> 60 us)
    print("kernels overlapping gaps > 60 us:")
    for g0, g1 in gaps:
        if g1 - g0 < 60000:
            continue
        inside = [(s, e, nm, q) for s, e, nm, q, st in rows if e > g0 and s < g1 and not nm.startswith("k_msm_accumulate")]
        print("  gap @%.0f (%.0f us): " % ((g0 - t0) / 1e3, (g1 - g0) / 1e3) + ", ".join("%s[q%s %+.0f..%+.0f]" % (nm, q, (s - g0) / 1e3, (e - g0) / 1e3) for s, e, nm, q in inside[:24]))
    if full:
        print("all kernels of proofs %d..%d:" % (lo, min(lo + 2, hi)))
        w1 = acc[min(lo + 2, hi) * per][0]
        for s, e, nm, q, st in rows:
            if s >= w0 and s <= w1:
                print("  %9.0f %8.0f q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, nm))


if __name__ == "__main__":
    main()
