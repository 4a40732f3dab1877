import time, numpy as np, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import zksnark_rs_amd as zk
from test_integer_roots import chain_rows_integers, chain_weights_integers
from zksnark_rs_amd import SplitMix64, ints_to_limbs
ctx = zk.Context(0)
for n in (4096, 16384):
    m, l, u, v, w = chain_rows_integers(n)
    rng = SplitMix64(1)
    weights = chain_weights_integers(n, rng.fr(), [rng.fr() for _ in range(n)])
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    up = ctx.crs_upload(n, m, l, ctx.crs_download(crs))
    t0 = time.perf_counter(); ctx.prove(up, qap, weights, 5, 7); t1 = time.perf_counter(); ctx.prove(up, qap, weights, 5, 7); t2 = time.perf_counter()
    print("n = %d: first proof over the uploaded CRS (change of basis) %.2f s, second %.4f s" % (n, t1 - t0, t2 - t1))
