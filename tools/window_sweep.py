#!/usr/bin/env python
"""BASELINE config 4: Pippenger window-size sweep on the synthetic 2^log_n chain QAP (1 MI355X), one JSON line per point.

  part "uniform"   every table at c                                   c = 13 .. 22
  part "big"       the 2^21-point tables (L, H + r B1 + s A) at c     the others automatic
  part "small"     the 2^20-point G1 table (A) at c                   the others automatic
  part "g2"        the G2 table at c                                  the others automatic
  each point: proofs/s with two proofs in flight (the metric) and ms of a lone proof; the proof bytes must not depend on c.
  part "lds"       north_star's form -- one wavefront per (window, chunk), buckets in LDS (csrc/msm_lds.hpp), c = 6 .. 9 -- against the
                   shipped form on ONE inner product of 2^log_n G1 points through zk_msm_g1 (event-timed kernels only: the call also
                   uploads the points and builds the window table, which a prover does once per CRS).

    python tools/window_sweep.py --log-n 20 > profiles/r3_window_sweep_2p20.jsonl
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
    ap.add_argument("--windows", default="13,14,15,16,17,18,19,20,21,22")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--parts", default="uniform,big,small,g2,lds")
    args = ap.parse_args()
    import torch
    import zksnark_rs_amd as zk
    from bench import build_instance, msm_window
    ctx = zk.Context(0)
    inst = build_instance(zk, ctx, args.log_n, 20260929)
    n, m = inst["n"], inst["m"]
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    cs = [int(x) for x in args.windows.split(",")]
    auto_big, auto_small, auto_g2 = msm_window(2 * n), msm_window(n), msm_window(n, 0, True)
    ref = [None]

    def point(part, big, small, g2):
        opt = g2 * 10000 + big * 100 + small
        ctx.set_option("msm_window_bits", opt)
        proof = ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"])   # rebuilds the tables
        assert ref[0] is None or proof == ref[0], "proof bytes depend on the window size"
        ref[0] = proof
        for _ in range(3):
            ctx.prove_wait(ctx.prove_submit(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inflight = []
        for _ in range(args.steps):
            if len(inflight) == 2:
                ctx.prove_wait(inflight.pop(0))
            inflight.append(ctx.prove_submit(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"]))
        while inflight:
            ctx.prove_wait(inflight.pop(0))
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / args.steps
        lone_steps = max(5, args.steps // 6)
        t0 = time.perf_counter()
        for _ in range(lone_steps):
            ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"])
        lone = (time.perf_counter() - t0) / lone_steps
        row = {"part": part, "c_2n_point_tables": big, "c_n_point_g1_table": small, "c_g2_table": g2,
               "windows": [254 // c + 1 for c in (big, small, g2)], "proofs_per_s_two_in_flight": round(1.0 / piped, 2),
               "ms_per_proof_two_in_flight": round(piped * 1e3, 3), "ms_lone_proof": round(lone * 1e3, 3)}
        print(json.dumps(row), flush=True)
        return row

    parts = args.parts.split(",")
    rows = []
    if "uniform" in parts:
        rows += [point("uniform", c, c, c) for c in cs]
    if "big" in parts:
        rows += [point("big", c, auto_small, auto_g2) for c in cs]
    if "small" in parts:
        rows += [point("small", auto_big, c, auto_g2) for c in cs]
    if "g2" in parts:
        rows += [point("g2", auto_big, auto_small, c) for c in cs]
    if rows:
        best = max(rows, key=lambda r: r["proofs_per_s_two_in_flight"])
        print(json.dumps({"summary": "best point of the sweep", "log_n": args.log_n, "automatic": [auto_big, auto_small, auto_g2], "best": best}), flush=True)
    ctx.set_option("msm_window_bits", 0)

    if "lds" in parts:
        # one inner product of n G1 points (the CRS's xi_g1) with uniform scalars, through zk_msm_g1
        pts = np.ascontiguousarray(ctx.crs_download(inst["crs"])["xi_g1"]).reshape(-1, 8)[:n]
        rng = zk.SplitMix64(99)
        sc = zk.ints_to_limbs([rng.fr() for _ in range(n)]).reshape(n, 4)
        want = None
        for form, c in [("shipped", 0), ("shipped", 17), ("shipped", 20)] + [("lds_buckets", -c) for c in (6, 7, 8, 9, 10)]:
            try:
                ctx.msm_g1(pts, sc, c)               # warm-up (allocations)
            except zk.ZkError as e:                  # a window whose buckets do not fit LDS
                print(json.dumps({"part": "lds", "form": form, "window_bits": abs(c), "refused": str(e)[:160]}), flush=True)
                continue
            ctx.set_option("profile", 2)
            ctx.profile_reset()
            got = ctx.msm_g1(pts, sc, c)
            prof = ctx.profile()
            ctx.set_option("profile", 0)
            assert want is None or np.array_equal(got, want), "the two forms disagree"
            want = got
            kern = {k: round(v["total_ms"], 3) for k, v in prof.items() if k.startswith("msm_") and "precompute" not in k}
            cc = abs(c) if c else msm_window(n)
            print(json.dumps({"part": "lds", "form": form, "window_bits": cc, "windows": 254 // cc + 1, "points": n,
                              "kernel_ms": round(sum(kern.values()), 3), "kernels": kern,
                              "G_additions_per_s": round((254 // cc + 1) * n / (sum(kern.values()) * 1e-3) / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
