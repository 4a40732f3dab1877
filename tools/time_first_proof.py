#!/usr/bin/env python3
"""Cold start: zk_setup, then the FIRST proof over the new CRS (it builds the fixed-base window tables) and the second.
    python tools/time_first_proof.py [log_n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch          # noqa: E402
import zksnark_rs_amd as zk   # noqa: E402
import bench          # noqa: E402

for log_n in [int(a) for a in sys.argv[1:]] or [20]:
    ctx = zk.Context(0)
    t0 = time.perf_counter()
    inst = bench.build_instance(zk, ctx, log_n, 20260929, sets=1)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    d_w = torch.from_numpy(inst["weights"].view(np.int64)).cuda()
    free0, total = torch.cuda.mem_get_info()
    ts = []
    for _ in range(3):
        a = time.perf_counter()
        pr = ctx.prove_dev(inst["crs"], inst["qap"], d_w.data_ptr(), inst["m"], inst["r"], inst["s"])
        ts.append(time.perf_counter() - a)
    free1, _ = torch.cuda.mem_get_info()
    print("2^%d gates: instance + zk_setup %.3f s | first proof (window tables) %.3f s | second %.4f s | third %.4f s | HBM in use %.2f GiB (tables + slot: +%.2f GiB) | sha %s"
          % (log_n, t_setup, ts[0], ts[1], ts[2], (total - free1) / 2**30, (free0 - free1) / 2**30, __import__("hashlib").sha256(pr).hexdigest()[:16]))
    del ctx, inst, d_w
