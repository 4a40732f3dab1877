#!/usr/bin/env python3
"""VALU instruction budget of one proof: sums SQ_INSTS_VALU per kernel over a rocprofv3 --pmc SQ_INSTS_VALU run of bench.py and divides
by the number of proofs (= k_assemble dispatches).  The prover is bound by VALU issue, so this deterministic count -- not the noisy
proofs/s of a shared box -- is what an optimisation has to move.
    python tools/valu_budget.py <rocprof output dir> [label]"""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("zk::", "")
    g2 = "Fq2" in n
    return re.sub(r"<.*", "", n) + ("<G2>" if g2 else "")


d = sys.argv[1]
per = collections.Counter()
calls = collections.Counter()
by_grid = collections.defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != "SQ_INSTS_VALU":
                continue
            n = short(row["Kernel_Name"])
            per[n] += float(row["Counter_Value"])
            calls[n] += 1
            by_grid[(n, row.get("Grid_Size", "?"))].append(float(row["Counter_Value"]))
proofs = calls.get("k_assemble", 0) or 1
setup = {"k_msm_precompute", "k_msm_precompute<G2>", "k_fixed_base_mul", "k_fixed_base_mul<G2>", "k_fixed_table", "k_fixed_table<G2>", "k_setup_comb_sparse",
         "k_lagrange_at", "k_setup_consts", "k_setup_powers", "k_powers_brev", "k_mid_table", "k_points_brev", "k_points_brev<G2>", "k_powers", "k_pts_on_curve"}
tot = sum(v for k, v in per.items() if k not in setup)
print("%s: %d proofs, %.4g VALU wave-instructions per proof" % (sys.argv[2] if len(sys.argv) > 2 else d, proofs, tot / proofs))
for k, v in per.most_common():
    if k in setup or v / proofs < 1e5:
        continue
    print("  %-28s %10.4g  (%5.2f %%)  %5.1f launches/proof" % (k, v / proofs, 100.0 * v / tot, calls[k] / proofs))
print("  per (kernel, grid) of the accumulations: median instructions per launch")
for (n, g), v in sorted(by_grid.items()):
    if "accumulate" in n:
        v.sort()
        print("    %-24s grid %9s  launches %3d  median %.4g" % (n, g, len(v), v[len(v) // 2]))
