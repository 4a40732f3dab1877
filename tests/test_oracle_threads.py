"""-m "not gpu": the multi-threaded same-algorithm CPU path (bench.py's cpu_baseline leg) produces the
same proof bytes as the single-threaded one, which test_oracle_kats pins to the faithful restatement."""
import numpy as np
import pytest

import oracle_lib
from zksnark_rs_amd import Context, SplitMix64, ints_to_limbs
from zksnark_rs_amd.circuits import chain_rows, chain_weights


@pytest.mark.parametrize("log_n,threads", [(3, 2), (6, 8), (7, 5)])
def test_threaded_cpu_prove_matches_single_thread(log_n, threads):
    orc = oracle_lib.load()
    rng = SplitMix64(900 + log_n)
    n = 1 << log_n
    m, l, u, v, w = chain_rows(log_n)
    weights = chain_weights(log_n, rng.fr(), [rng.fr() for _ in range(n)])
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    builder = Context.__new__(Context)          # descriptor builders only; no device, no library call
    desc = builder.sparse_desc(log_n, m, l, u, v, w)
    arrs = orc.setup_sparse(desc, td, n, m, l, False)
    cdesc = Context.crs_desc(n, m, l, arrs)
    want = orc.prove_sparse(desc, cdesc, weights, r, s, False)
    sec, got = orc.time_prove_sparse_mt(desc, cdesc, weights, r, s, threads)
    assert got == want and sec > 0
    assert orc.time_prove_sparse_mt(desc, cdesc, weights, r, s, 1)[1] == want
