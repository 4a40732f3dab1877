#!/usr/bin/env python3
"""One-off check at the largest supported circuit (not in the test suite: a minute of host-side witness generation):
2^23 constraints over the roots of unity (16.8 M wires; the six transforms take three passes), proof == the oracle's closed form.
Usage: python tools/check_max_size.py [log_n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zksnark_rs_amd as zk                                           # noqa: E402
from zksnark_rs_amd.circuits import chain_rows, chain_weights         # noqa: E402
import oracle_lib                                                     # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
t0 = time.time()
rng = zk.SplitMix64(2300 + log_n)
n = 1 << log_n
m, l, u, v, w = chain_rows(log_n)
weights = chain_weights(log_n, rng.fr(), [rng.next() for _ in range(n)])
td = zk.ints_to_limbs([rng.fr() for _ in range(5)])
r, s = rng.fr(), rng.fr()
print("instance built in %.1f s" % (time.time() - t0), flush=True)
ctx = zk.Context(0)
t0 = time.time()
qap = ctx.qap_sparse(log_n, m, l, u, v, w)
crs = ctx.setup(qap, td)
print("upload + setup %.2f s" % (time.time() - t0), flush=True)
t0 = time.time()
p1 = ctx.prove(crs, qap, weights, r, s)
t1 = time.time()
p2 = ctx.prove(crs, qap, weights, r, s)
t2 = time.time()
print("first proof (builds the window tables) %.2f s, second %.3f s" % (t1 - t0, t2 - t1), flush=True)
import torch                                                          # noqa: E402
free_b, total_b = torch.cuda.mem_get_info()
print("HBM in use %.1f GiB" % ((total_b - free_b) / 2**30), flush=True)
orc = oracle_lib.load()
desc = ctx.sparse_desc(log_n, m, l, u, v, w)
t0 = time.time()
want = orc.trapdoor_proof_sparse(desc, td, weights, r, s)
print("oracle closed form %.1f s" % (time.time() - t0), flush=True)
assert p1 == p2 == want, "proof bytes differ"
print("ok 2^%d gates: proof == closed form" % log_n)
