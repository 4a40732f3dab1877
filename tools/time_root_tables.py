#!/usr/bin/env python
"""Once-per-root-set cost of the arbitrary-roots form (zk_qap_upload_sparse_roots: N'(r_k) in O(n^2), block matrices, node images) and
the HBM it holds, at 2^log_n gates of the chain circuit over affine images of the integers.

    python tools/time_root_tables.py 16 18 20
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import zksnark_rs_amd as zk
    from zksnark_rs_amd.circuits import chain_rows
    ctx = zk.Context(0)
    for log_n in [int(a) for a in sys.argv[1:]] or [16]:
        n = 1 << log_n
        m, l, u, v, w = chain_rows(log_n)
        k = np.arange(1, n + 1, dtype=object)
        roots = zk.ints_to_limbs([int(x) for x in (12345678901234567 * k + 987654321) % zk.R_MODULUS]).reshape(n, 4)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        qap = ctx.qap_sparse_roots(roots, m, l, u, v, w)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        free1 = torch.cuda.mem_get_info()[0]
        print("2^%d gates: zk_qap_upload_sparse_roots %.2f s, %.2f GiB of HBM held by the QAP" % (log_n, dt, (free0 - free1) / 2**30), flush=True)
        del qap


if __name__ == "__main__":
    main()
