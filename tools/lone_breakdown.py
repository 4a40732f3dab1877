#!/usr/bin/env python
"""A lone proof of 2^log_n gates: how long the host takes to enqueue it (zk_prove_submit) and how long it then waits (zk_prove_wait).

    python tools/lone_breakdown.py --log-n 16
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--opt", action="append", default=[], help="KEY=VALUE for zk_set_option (measurement build: ZKGPU_LIB)")
    args = ap.parse_args()
    import torch
    import zksnark_rs_amd as zk
    from bench import build_instance
    ctx = zk.Context(0)
    ctx.set_option("profile", 0)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    inst = build_instance(zk, ctx, args.log_n, 20260929)
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    a = (inst["crs"], inst["qap"], d_w.data_ptr(), inst["m"], inst["r"], inst["s"])
    for _ in range(8):
        ctx.prove_wait(ctx.prove_submit(*a))
    sub, wait = [], []
    for _ in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = ctx.prove_submit(*a)
        t1 = time.perf_counter()
        ctx.prove_wait(t)
        t2 = time.perf_counter()
        sub.append(t1 - t0); wait.append(t2 - t1)
    sub.sort(); wait.sort()
    k = len(sub) // 2
    print(" ".join(args.opt), end=" ")
    print("2^%d gates, lone proof: submit (host enqueue) median %.3f ms, then wait %.3f ms; sum %.3f ms" % (args.log_n, sub[k] * 1e3, wait[k] * 1e3, (sub[k] + wait[k]) * 1e3))


if __name__ == "__main__":
    main()
