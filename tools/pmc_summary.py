#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/<round>_pmc_traffic.json.

Usage (on the GPU box, separate passes as gpurun requires -- FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rN_pmc_traffic.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are reported in KiB; FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced streaming
reads, so it is doubled ("corrected"); WRITE_SIZE is taken as reported (uncalibrated).  For gather-dominated
kernels (k_msm_accumulate reads one random 64 B / 128 B point per addition) the doubled figure is an upper
bound; both raw and corrected values are kept.
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


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; corrected = (2 * fetch + write) * 1024 bytes",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        out["kernels"][name] = {"launches_sampled": max(len(f), len(w)), "fetch_raw_KiB_per_launch": round(fa, 1),
                                "write_raw_KiB_per_launch": round(wa, 1),
                                "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
