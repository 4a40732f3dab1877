#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/<round>_pmc_traffic.json.

Usage (on the GPU box, separate passes as gpurun requires -- FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rN_pmc_traffic.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are reported in KiB; FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced streaming
reads, so it is doubled ("corrected"); WRITE_SIZE is taken as reported (uncalibrated).  The guide asks for a
calibration on a known byte count for other access patterns: tools/ubench_gather.hip (random per-lane gathers
from a 4 GiB table; profiles/r1_fetch_calibration.txt) measured FETCH_SIZE = 1.00 x the gathered bytes for
64-byte elements (the G1 accumulation's pattern) and 0.50 x for 128-byte elements and for streams, so the
factor is 1 for k_msm_accumulate<Fq> and 2 everywhere else.  Raw values are kept beside the corrected ones.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    per = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].strip()
                if name.startswith("void "):
                    name = name[5:]
                per[name].append(float(row["Counter_Value"]))
    return per


FETCH_FACTOR = {"zk::k_msm_accumulate<zk::Fp<zk::FqParams> >": 1.0}   # calibrated; default 2.0


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; corrected = (factor * fetch + write) * 1024 bytes, factor = 2 "
                    "(streams, 128 B gathers) or 1 (64 B gathers: k_msm_accumulate<Fq>), see profiles/r1_fetch_calibration.txt",
           "kernels": {}}
    # kernels of the one-off set-up (CRS, window tables) do not belong to a proof
    setup = ("k_msm_precompute", "k_fixed_base_mul", "k_fixed_table", "k_setup_", "k_lagrange_at", "k_powers", "k_mid_table", "k_points_brev",
             "k_pts_on_curve", "k_pts_to_mont", "k_g2_subgroup")
    proofs = max(len(fetch.get("zk::k_assemble", [])), len(write.get("zk::k_assemble", [])), 1)
    total = 0.0
    for name in sorted(set(fetch) | set(write)):
        if any(t in name for t in setup):
            continue
        total += (FETCH_FACTOR.get(name, 2.0) * sum(fetch.get(name, [])) + sum(write.get(name, []))) * 1024
    out["proofs_sampled"] = proofs
    out["hbm_bytes_per_proof_corrected"] = int(total / proofs)
    try:
        out["commit"] = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".build_commit")).read().strip()
    except OSError:
        out["commit"] = None
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        k = FETCH_FACTOR.get(name, 2.0)
        out["kernels"][name] = {"launches_sampled": max(len(f), len(w)), "fetch_raw_KiB_per_launch": round(fa, 1),
                                "write_raw_KiB_per_launch": round(wa, 1), "fetch_factor": k,
                                "hbm_bytes_per_launch_corrected": int((k * fa + wa) * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
