#!/usr/bin/env python3
"""How long does an RCCL collective of the multi-GPU exchange wait when bucket accumulations fill the chip?

One GPU: ZK_COMM_FORCE_RCCL=1 makes the one-rank zk_comm a real RCCL communicator (grouped self send / recv).  The script times
zk_comm_all_to_all of 20 MiB (one rank's share of a round's scalar exchange at 2^20 gates and 8 ranks) on an idle GPU, then
again while two 2^20 proofs are in flight (zk_prove_submit x 2: their accumulations hold every wave slot for ~11 ms each).  The
difference is what the collectives' kernels wait for compute units -- the number that decides whether RCCL needs units reserved.

    ZK_COMM_FORCE_RCCL=1 python tools/rccl_starvation.py [log_n [reserve]]
reserve > 0 (ZKGPU_LIB = the measurement build): the inner-product streams are masked as zk_mgpu_create masks them over a multi-rank
communicator (option comm_cu_reserve), and the script also reports the proofs/s the masked streams cost."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ZK_COMM_FORCE_RCCL", "1")
import torch  # noqa: E402
import zksnark_rs_amd as zk  # noqa: E402
from zksnark_rs_amd.circuits import chain_rows, chain_weights  # noqa: E402
from zksnark_rs_amd.distributed import Comm  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reserve = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # compute units per XCD kept free of inner-product work (measurement build)
    ctx = zk.Context(0)
    if reserve:
        ctx.set_option("apply_cu_reserve", reserve)
    comm = Comm(ctx, 0, 1, Comm.unique_id())
    rng = zk.SplitMix64(11)
    m, l, u, v, w = chain_rows(log_n)
    wts = chain_weights(log_n, rng.fr(), [rng.fr() for _ in range(1 << log_n)])
    qap = ctx.qap_sparse(log_n, m, l, u, v, w)
    crs = ctx.setup(qap, zk.ints_to_limbs([rng.fr() for _ in range(5)]))
    r, s = rng.fr(), rng.fr()
    dw = torch.from_numpy(np.ascontiguousarray(wts).view(np.int64)).cuda()
    nbytes = 20 << 20
    a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.zeros_like(a)
    torch.cuda.synchronize()
    want = ctx.prove_dev(crs, qap, dw.data_ptr(), m, r, s)     # tables, slots
    for _ in range(3):
        ctx.prove_wait(ctx.prove_submit(crs, qap, dw.data_ptr(), m, r, s))

    def a2a_ms():
        t0 = time.perf_counter()
        comm.all_to_all(a.data_ptr(), b.data_ptr(), nbytes)
        return 1e3 * (time.perf_counter() - t0)

    for _ in range(5):
        a2a_ms()
    idle = sorted(a2a_ms() for _ in range(30))
    loaded = []
    for i in range(30):
        t1 = ctx.prove_submit(crs, qap, dw.data_ptr(), m, r, s)
        t2 = ctx.prove_submit(crs, qap, dw.data_ptr(), m, r, s)
        time.sleep(0.001 + 0.0007 * (i % 12))     # land at different points of the proofs' 22 ms
        loaded.append(a2a_ms())
        assert ctx.prove_wait(t1) == want and ctx.prove_wait(t2) == want
    loaded.sort()
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]   # noqa: E731
    t0 = time.perf_counter()
    tk = []
    for i in range(40):
        if len(tk) == 2:
            ctx.prove_wait(tk.pop(0))
        tk.append(ctx.prove_submit(crs, qap, dw.data_ptr(), m, r, s))
    for t in tk:
        ctx.prove_wait(t)
    rate = 40 / (time.perf_counter() - t0)
    print("zk_comm_all_to_all of %d MiB through a one-rank RCCL communicator (self send / recv), 2^%d gates, %d compute units per XCD reserved; "
          "single-GPU rate with these streams %.2f proofs/s" % (nbytes >> 20, log_n, reserve, rate))
    print("  idle GPU:                 median %.3f ms   p90 %.3f   max %.3f" % (q(idle, 0.5), q(idle, 0.9), idle[-1]))
    print("  two proofs in flight:     median %.3f ms   p90 %.3f   max %.3f" % (q(loaded, 0.5), q(loaded, 0.9), loaded[-1]))
    print("  -> the collective's kernel waits ~%.3f ms (median) for compute units under the accumulations" % (q(loaded, 0.5) - q(idle, 0.5)))
    comm.close()


if __name__ == "__main__":
    main()
