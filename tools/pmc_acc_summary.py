#!/usr/bin/env python3
"""Counters of the bucket accumulations -> profiles/rN_pmc_acc.json (read by bench.py: nothing about the instruction mix is typed in).

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES \\
              GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline [--log-n N]
    python tools/pmc_acc_summary.py <dir> <log_n> > profiles/rN_pmc_acc.json

Per accumulation kernel: launches, median duration, the counters summed per proof, and the derived figures
  wave_instr_per_addition   SQ_INSTS_VALU per proof / additions per proof (additions = windows x points of the products, computed from
                            the automatic windows: the only number here that is not a counter);
  int64_share               SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU: the executed instructions of the 64-bit integer class (v_mad_u64_u32,
                            v_mad_i64_i32, v_lshl_add_u64, 64-bit shifts) -- the bulk of the 4-cycle issue class of tools/ubench_valu.hip;
  int32_share               SQ_INSTS_VALU_INT32 / SQ_INSTS_VALU (32-bit adds, v_mul_lo_u32, carry adds; moves, masks and selects are in
                            neither counter);
  share_4_cycle_class       = int64_share: what bench.py prices the issue peak of the mix with.  A lower bound of the class (v_mul_lo_u32
                            and the carry adds, ~3 % of the loop, also issue at 4 cycles but are counted as INT32), so the peak derived
                            from it is an upper bound and the reported fraction a lower bound.  (SQ_ACTIVE_INST_VALU, which the
                            microarchitecture guide lists in units of 4 cycles, EQUALS SQ_INSTS_VALU on this kernel on gfx950 -- it does
                            not weigh instructions by their issue cost and is kept in the file only to show that.)
  sustained_clock_GHz       GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("zk::", "")
    g2 = "Fq2" in n
    return re.sub(r"<.*", "", n) + ("<G2>" if g2 else "")


def commit():
    """the commit the profiled build was made from: tools/gpu.sh writes it into .build_commit before the snapshot leaves (the GPU box has no .git)"""
    try:
        return open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".build_commit")).read().strip()
    except OSError:
        return None


def main():
    d, log_n = sys.argv[1], int(sys.argv[2])
    from bench import msm_window
    n = 1 << log_n
    # round 5: L (2n - 1 points) and H + r B1 + s A (2n) are ONE product over xi_t | xi | sum_delta (option merge_lh)
    counts = {"k_msm_accumulate": [n, 4 * n - 1], "k_msm_accumulate<G2>": [n]}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    dur = collections.defaultdict(list)
    proofs = 0
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k == "k_assemble" and row["Counter_Name"] == "SQ_INSTS_VALU":
                    proofs += 1
                if k in counts:
                    per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    if row["Counter_Name"] == "SQ_INSTS_VALU":
                        calls[k] += 1
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k in counts:
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    proofs = max(proofs, 1)
    out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES "
                     "GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --log-n %d (counter passes "
                     "serialise the kernels: stand-alone durations)" % log_n,
           "log_n": log_n, "proofs": proofs, "commit": commit(), "kernels": {}}
    for k, pts in counts.items():
        if not calls[k]:
            continue
        g2 = k.endswith("<G2>")
        adds = sum((254 // msm_window(c, 0, g2) + 1) * c for c in pts)
        c = {name: v / proofs for name, v in per[k].items()}
        v = sorted(dur[k])
        total_ms = sum(v) / proofs
        inst = c.get("SQ_INSTS_VALU", 0.0)
        i64 = c.get("SQ_INSTS_VALU_INT64", 0.0) / inst if inst else None
        out["kernels"]["msm_accumulate_g2" if g2 else "msm_accumulate_g1"] = {
            "kernel": k, "launches_per_proof": calls[k] / proofs, "ms_per_proof_stand_alone": round(total_ms, 4), "median_launch_ms": round(v[len(v) // 2], 4),
            "additions_per_proof": adds, "counters_per_proof": {a: round(b) for a, b in sorted(c.items())},
            "wave_instr_per_addition": round(inst * 64.0 / adds, 1),
            "int64_share": round(c.get("SQ_INSTS_VALU_INT64", 0.0) / inst, 4) if inst else None,
            "int32_share": round(c.get("SQ_INSTS_VALU_INT32", 0.0) / inst, 4) if inst else None,
            "share_4_cycle_class": round(i64, 4) if i64 else None,
            "G_wave_instr_per_s_stand_alone": round(inst / (total_ms * 1e-3) / 1e9, 1),
            "sustained_clock_GHz": round(c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (total_ms * 1e-3) / 1e9, 3) if c.get("GRBM_GUI_ACTIVE") else None,
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
