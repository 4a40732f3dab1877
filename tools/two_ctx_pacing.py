#!/usr/bin/env python3
"""Two contexts on ONE device, each with its own streams, proofs submitted alternately (depth d each): does the GPU overlap proofs of
different contexts where it does not overlap the proofs of one?  (diagnostic for the serialisation of small circuits)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import zksnark_rs_amd as zk
import bench

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
proofs = 60
cs = []
for _ in range(nctx):
    ctx = zk.Context(0)
    inst = bench.build_instance(zk, ctx, log_n, 20260929, sets=1)
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    want = ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), inst["m"], inst["r"], inst["s"])
    for _ in range(6):
        ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), inst["m"], inst["r"], inst["s"])
    cs.append(dict(ctx=ctx, inst=inst, d_w=d_w, want=want, inflight=[]))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(proofs):
    c = cs[i % nctx]
    if len(c["inflight"]) == depth:
        assert c["ctx"].prove_wait(c["inflight"].pop(0)) == c["want"]
    inst = c["inst"]
    c["inflight"].append(c["ctx"].prove_submit(inst["crs"], inst["qap"], c["d_w"].data_ptr(), inst["m"], inst["r"], inst["s"]))
for c in cs:
    while c["inflight"]:
        c["ctx"].prove_wait(c["inflight"].pop(0))
print("2^%d gates, %d context(s) x depth %d: %.1f proofs/s" % (log_n, nctx, depth, proofs / (time.perf_counter() - t0)))
