#!/usr/bin/env python3
"""Where the HOST spends its time in a pipelined stream of proofs: per proof, the wall time of zk_prove_submit and of zk_prove_wait.
    python tools/host_pacing.py [--log-n 16] [--depth 4] [--proofs 40]
If the waits are ~0 the host is the bottleneck (the GPU pipeline is never more than one proof deep); if the submits are long, something
inside zk_prove_submit blocks on the GPU."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--proofs", type=int, default=40)
    args = ap.parse_args()
    import torch
    import zksnark_rs_amd as zk
    import bench
    ctx = zk.Context(0)
    inst = bench.build_instance(zk, ctx, args.log_n, 20260929, sets=1)
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    m = inst["m"]
    want = ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"])
    for _ in range(8):
        ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"])
    torch.cuda.synchronize()
    inflight, rows = [], []
    t0 = time.perf_counter()
    for i in range(args.proofs):
        tw = 0.0
        if len(inflight) == args.depth:
            a = time.perf_counter()
            assert ctx.prove_wait(inflight.pop(0)) == want
            tw = time.perf_counter() - a
        a = time.perf_counter()
        inflight.append(ctx.prove_submit(inst["crs"], inst["qap"], d_w.data_ptr(), m, inst["r"], inst["s"]))
        ts = time.perf_counter() - a
        rows.append((1e3 * (a - t0), 1e3 * tw, 1e3 * ts))
    while inflight:
        ctx.prove_wait(inflight.pop(0))
    total = time.perf_counter() - t0
    print("2^%d gates, depth %d: %.1f proofs/s" % (args.log_n, args.depth, args.proofs / total))
    print("  proof  submit starts at (ms)  waited before (ms)  submit took (ms)")
    for i, (at, tw, ts) in enumerate(rows):
        print("  %4d  %10.3f  %10.3f  %10.3f" % (i, at, tw, ts))


if __name__ == "__main__":
    main()
