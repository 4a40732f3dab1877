#!/usr/bin/env python
"""BASELINE config 4: Pippenger window-size sweep on the synthetic 2^log_n chain QAP (1 MI355X).

For each c the fixed-base tables T[w][i] = 2^(c w) P_i are rebuilt (W = floor(254/c)+1 windows) and
`steps` proofs are timed.  One JSON line per c, then a summary line.
    python tools/window_sweep.py --log-n 20 --windows 10,12,13,14,15,16
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--windows", default="10,12,13,14,15,16")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import zksnark_rs_amd as zk
    from bench import build_instance
    ctx = zk.Context(0)
    inst = build_instance(zk, ctx, args.log_n, 20260929)
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    ref = None
    results = []
    for c in [int(x) for x in args.windows.split(",")]:
        ctx.set_option("msm_window_bits", c)
        t0 = time.perf_counter()
        proof = ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), inst["m"], inst["r"], inst["s"])   # rebuilds the tables
        t_build = time.perf_counter() - t0
        assert ref is None or proof == ref, "proof bytes depend on the window size"
        ref = proof
        ctx.set_option("profile", 1)
        ctx.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), inst["m"], inst["r"], inst["s"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        prof = ctx.profile()
        ctx.set_option("profile", 0)
        row = {"window_bits": c, "windows": 254 // c + 1, "buckets": 1 << (c - 1), "ms_per_proof": round(dt * 1e3, 3),
               "proofs_per_s": round(1.0 / dt, 3), "table_build_s": round(t_build, 2),
               "accumulate_g1_ms": round(prof.get("msm_accumulate_g1", {}).get("total_ms", 0) / args.steps, 3),
               "accumulate_g2_ms": round(prof.get("msm_accumulate_g2", {}).get("total_ms", 0) / args.steps, 3)}
        results.append(row)
        print(json.dumps(row), flush=True)
    best = min(results, key=lambda r: r["ms_per_proof"])
    print(json.dumps({"sweep": "pippenger window bits", "log_n": args.log_n, "best": best}))


if __name__ == "__main__":
    main()
