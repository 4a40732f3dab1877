"""Once-per-CRS cost of the change of basis of an uploaded (powers-only) CRS for an integer-roots QAP, and the check that the proof over
the derived Lagrange-basis points equals the proof over the CRS zk_setup wrote:
   python tools/time_change_of_basis.py [log2 sizes ...]     (env BASIS_TREE_MIN: option basis_tree_min, e.g. 1073741824 = n^2 form)"""
import os, time, numpy as np, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import zksnark_rs_amd as zk
from test_integer_roots import chain_rows_integers, chain_weights_integers
from zksnark_rs_amd import SplitMix64, ints_to_limbs
ctx = zk.Context(0)
if os.environ.get("BASIS_TREE_MIN"):
    ctx.set_option("basis_tree_min", int(os.environ["BASIS_TREE_MIN"]))
sizes = [int(a) for a in sys.argv[1:]] or [12, 14]
for lg in sizes:
    n = 1 << lg
    m, l, u, v, w = chain_rows_integers(n)
    rng = SplitMix64(1)
    weights = chain_weights_integers(n, rng.fr(), [rng.fr() for _ in range(n)])
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    want = ctx.prove(crs, qap, weights, 5, 7)
    up = ctx.crs_upload(n, m, l, ctx.crs_download(crs))
    t0 = time.perf_counter(); a = ctx.prove(up, qap, weights, 5, 7); t1 = time.perf_counter(); b = ctx.prove(up, qap, weights, 5, 7); t2 = time.perf_counter()
    print("n = 2^%d: first proof over the uploaded CRS (change of basis, basis_tree_min = %d) %.2f s, second %.4f s, bytes equal to the zk_setup CRS's proof: %s"
          % (lg, ctx.get_option("basis_tree_min"), t1 - t0, t2 - t1, a == want and b == want), flush=True)
    del up, crs, qap
