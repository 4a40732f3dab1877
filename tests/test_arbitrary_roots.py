"""Arbitrary root sets (RootRepresentation::roots() is caller data, circuit/mod.rs:201-214): interpolation through caller-supplied
nodes by the sub-product tree (csrc/interp.hip), and the sparse QAP form built on it."""
import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd import SplitMix64, ints_to_limbs, limbs_to_int

R = zk.R_MODULUS
pytestmark = pytest.mark.gpu


def lagrange_coeffs(roots, values):
    """the reference's way (coefficient_poly.rs:159-200 restated with Python integers): sum_k v_k prod_{j != k} (x - r_j) / (r_k - r_j)"""
    n = len(roots)
    full = [1]
    for r in roots:                       # N(x)
        nxt = [0] * (len(full) + 1)
        for i, c in enumerate(full):
            nxt[i + 1] = (nxt[i + 1] + c) % R
            nxt[i] = (nxt[i] - r * c) % R
        full = nxt
    out = [0] * n
    for k in range(n):
        q = [0] * n                        # N / (x - r_k) by synthetic division
        carry = 0
        for i in range(n, 0, -1):
            carry = (full[i] + roots[k] * carry) % R
            q[i - 1] = carry
        den = 1
        for j in range(n):
            if j != k:
                den = den * (roots[k] - roots[j]) % R
        f = values[k] * pow(den, -1, R) % R
        for i in range(n):
            out[i] = (out[i] + f * q[i]) % R
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 333])
def test_interpolation_matches_lagrange_sums(ctx, n):
    rng = SplitMix64(4000 + n)
    roots = []
    while len(roots) < n:
        r = rng.fr()
        if r not in roots:
            roots.append(r)
    if n >= 3:
        roots[0], roots[1], roots[2] = 0, 1, R - 1
    values = [rng.fr() for _ in range(n)]
    if n >= 3:
        values[1] = 0
    got = ctx.interpolate_fr(ints_to_limbs(roots).reshape(n, 4), ints_to_limbs(values).reshape(n, 4))
    assert [limbs_to_int(x) for x in got] == lagrange_coeffs(roots, values)


def test_interp_single_block(ctx):
    """The layout of InterpTree::qmat ([block][node k][coefficient i], csrc/interp.hpp) pinned through ONE bottom block of known nodes:
    with the unit vector e_k as values the interpolant IS row k of the block's matrix -- the Lagrange polynomial of node k -- so a
    transposed read (node and coefficient swapped) shows up as the wrong polynomial for every k except on the diagonal."""
    n = 64
    roots = [(7 * k * k + 3 * k + 11) % R for k in range(n)]
    assert len(set(roots)) == n
    for k in (0, 1, 31, 62, 63):
        values = [1 if j == k else 0 for j in range(n)]
        got = ctx.interpolate_fr(ints_to_limbs(roots).reshape(n, 4), ints_to_limbs(values).reshape(n, 4))
        want = lagrange_coeffs(roots, values)
        assert [limbs_to_int(x) for x in got] == want
        assert want[n - 1] != 0 and want != lagrange_coeffs(roots, [1 if j == (k + 1) % n else 0 for j in range(n)])


@pytest.mark.parametrize("log_n", [10, 14, 17, 20])
def test_interpolation_on_permuted_roots_of_unity(ctx, log_n):
    """Size-independent check: when the nodes are the 2^k-th roots of unity in a scrambled order the interpolant is the inverse
    transform of the unscrambled values (zk_ntt_fr, pinned to the reference's dft KATs through the oracle).  2^20 nodes take the
    large-tree form of interp_run (2^20 coefficients per level and more: csrc/interp.hip), which every arbitrary-roots QAP of more
    than 2^18 gates proves through."""
    n = 1 << log_n
    gen = np.random.default_rng(log_n)
    perm = gen.permutation(n)
    vals = gen.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    vals[:, 3] &= np.uint64((1 << 60) - 1)
    delta = np.zeros((n, 4), np.uint64); delta[1, 0] = 1
    w_pows = ctx.ntt_fr(delta)                                  # w^j
    natural = np.zeros_like(vals); natural[perm] = vals         # value at w^perm[k] is vals[k]
    want = ctx.ntt_fr(natural, inverse=True)
    got = ctx.interpolate_fr(w_pows[perm], vals)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [64, 65, 1000, 4096, 5000, (1 << 15) + 17])
def test_interpolation_small_and_large_tree_forms_agree(n):
    """interp_run has two forms of the upward pass (zero-padded 4s-point transforms below 2^20 coefficients per level; above, the
    parents' images assembled from the children's 2s-point ones and a twisted second half).  Option interp_large_log moves the
    switch: both forms on the same nodes and values must give the same coefficients, which also evaluate back at the nodes."""
    rng = SplitMix64(515 + n)
    roots = [rng.fr() for _ in range(n)]
    values = [rng.fr() for _ in range(n)]
    rl, vl = ints_to_limbs(roots).reshape(n, 4), ints_to_limbs(values).reshape(n, 4)
    got = []
    for log in (40, 0):      # always the small-tree form, always the large-tree form
        c = zk.Context(0)
        c.set_option("interp_large_log", log)
        got.append(c.interpolate_fr(rl, vl))
    assert np.array_equal(got[0], got[1])
    coef = [limbs_to_int(x) for x in got[1]]
    for k in (0, n // 3, n - 1):
        acc = 0
        for cf in reversed(coef):
            acc = (acc * roots[k] + cf) % R
        assert acc == values[k], k


def test_interpolation_with_n_not_a_power_of_two_evaluates_back(ctx):
    """n = 5000 random nodes: the coefficients, evaluated by Horner at 40 of the nodes and at their own index, give the values back"""
    n = 5000
    rng = SplitMix64(91)
    roots = [rng.fr() for _ in range(n)]
    values = [rng.fr() for _ in range(n)]
    coef = [limbs_to_int(x) for x in ctx.interpolate_fr(ints_to_limbs(roots).reshape(n, 4), ints_to_limbs(values).reshape(n, 4))]
    for k in list(range(0, n, 131)) + [n - 1]:
        acc = 0
        for c in reversed(coef):
            acc = (acc * roots[k] + c) % R
        assert acc == values[k], k


def test_interpolation_refuses_repeated_roots(ctx):
    roots = ints_to_limbs([5, 7, 5]).reshape(3, 4)
    with pytest.raises(zk.ZkError) as e:
        ctx.interpolate_fr(roots, roots)
    assert e.value.status == zk._lib.ZK_ERR_ARG


# ---- the sparse QAP form over the caller's roots (zk_qap_upload_sparse_roots, csrc/arbroots.hip) ---------------------------------
from test_integer_roots import random_rows, chain_rows_integers, chain_weights_integers   # noqa: E402


def distinct_roots(rng, n):
    seen, out = set(), []
    while len(out) < n:
        r = rng.fr()
        if r not in seen:
            seen.add(r); out.append(r)
    return out


def dense_from_rows(roots, rows, m):
    """QAP::from(root_rep) restated (fr.rs:140-173): every wire polynomial by the Lagrange sums above -> (m, n, 4) coefficient limbs"""
    n = len(roots)
    ptr, gate, val = rows
    out = np.zeros((m, n, 4), np.uint64)
    for i in range(m):
        vals = [0] * n
        for e in range(int(ptr[i]), int(ptr[i + 1])):
            vals[int(gate[e])] = limbs_to_int(val[e])
        if any(vals):
            out[i] = ints_to_limbs(lagrange_coeffs(roots, vals)).reshape(n, 4)
    return out


def root_poly(roots):
    full = [1]
    for r in roots:
        nxt = [0] * (len(full) + 1)
        for i, c in enumerate(full):
            nxt[i + 1] = (nxt[i + 1] + c) % R
            nxt[i] = (nxt[i] - r * c) % R
        full = nxt
    return ints_to_limbs(full).reshape(len(roots) + 1, 4)


@pytest.mark.parametrize("n,m,l", [(1, 4, 1), (2, 5, 0), (5, 12, 2), (17, 30, 3), (40, 70, 1)])
def test_arbitrary_roots_match_the_faithful_oracle_and_the_dense_form(ctx, orc, n, m, l):
    """random distinct roots, random rows: CRS arrays and proof bytes of the sparse arbitrary-roots form == the oracle's faithful
    restatement of setup / prove over the dense QAP that QAP::from builds from the same root representation == the dense device form;
    satisfying-or-not witnesses of three lengths; a CRS uploaded from the reference's arrays serves the form as it is."""
    rng = SplitMix64(7700 + n)
    roots = distinct_roots(rng, n)
    if n >= 5:
        roots[0], roots[1] = 0, 1
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    du, dv, dw, dt = dense_from_rows(roots, u, m), dense_from_rows(roots, v, m), dense_from_rows(roots, w, m), root_poly(roots)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    qs = ctx.qap_sparse_roots(ints_to_limbs(roots).reshape(n, 4), m, l, u, v, w)
    qd = ctx.qap_dense(du, dv, dw, dt, l)
    cs, cd = ctx.setup(qs, td), ctx.setup(qd, td)
    a_s, a_d, a_o = ctx.crs_download(cs), ctx.crs_download(cd), orc.setup_dense(du, dv, dw, dt, l, td)
    for k in a_o:
        assert np.array_equal(a_s[k], a_o[k]), k
        assert np.array_equal(a_d[k], a_o[k]), k
    up = ctx.crs_upload(n, m, l, a_o)
    crs_desc = ctx.crs_desc(n, m, l, a_o)
    r, s = rng.fr(), rng.fr()
    for count in (m, max(l + 1, m - 2), m + 2):
        wts = ints_to_limbs([1] + [rng.fr() for _ in range(count - 1)])
        want = orc.prove_dense(du, dv, dw, dt, l, crs_desc, wts, r, s)
        assert ctx.prove(cs, qs, wts, r, s) == want, count
        assert ctx.prove(cd, qd, wts, r, s) == want, count
        assert ctx.prove(up, qs, wts, r, s) == want, count


@pytest.mark.parametrize("n", [3, 64, 65, 600, 4099])
def test_the_integers_as_arbitrary_roots(ctx, n):
    """roots 1..n handed over as caller data: the CRS and the bytes of the integer-roots form (which the suite pins to the faithful
    oracle, the dense form and the closed form)"""
    rng = SplitMix64(7800 + n)
    m, l = 2 * n + 7, 2
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    qi = ctx.qap_sparse_integers(n, m, l, u, v, w)
    qa = ctx.qap_sparse_roots(ints_to_limbs(list(range(1, n + 1))).reshape(n, 4), m, l, u, v, w)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    ci, ca = ctx.setup(qi, td), ctx.setup(qa, td)
    ai, aa = ctx.crs_download(ci), ctx.crs_download(ca)
    for k in ai:
        assert np.array_equal(ai[k], aa[k]), k
    r, s = rng.fr(), rng.fr()
    for count in (m, m - 3):
        wts = ints_to_limbs([1] + [rng.fr() for _ in range(count - 1)])
        assert ctx.prove(ca, qa, wts, r, s) == ctx.prove(ci, qi, wts, r, s)


@pytest.mark.parametrize("n", [1000, (1 << 16) + 3, 1 << 18, (1 << 19) + 1])
def test_affine_images_of_the_integers_match_the_closed_form(ctx, orc, n):
    """Size-independent property.  Over the roots r_k = a k + b the wire polynomials are u_i((x - b) / a), so a proof with trapdoor x'
    is the integer-roots proof with trapdoor x = (x' - b) / a (A, B and every term of C are values of the same polynomials) -- which
    the oracle's closed form gives at any size.  The device treats the roots as arbitrary field elements (dense form at 2^18 gates:
    3 m n x 32 B = 13 TB).  Valid and invalid witness; zk_verify accepts / rejects; pipelined submissions.  2^19 + 1 gates: the tree is
    padded to 2^20 leaves and the two vectors of a proof make 2^21 coefficients per level -- the large-tree form of interp_run."""
    m, l, u, v, w = chain_rows_integers(n)
    rng = SplitMix64(7900 + (n & 0xFFFF))
    x, avals = rng.fr(), [rng.fr() for _ in range(n)]
    weights = chain_weights_integers(n, x, avals)
    a, b = rng.fr() | 1, rng.fr()
    k = np.arange(1, n + 1, dtype=object)
    roots = ints_to_limbs([int(v_) for v_ in (a * k + b) % R]).reshape(n, 4)
    desc = ctx.sparse_desc(0, m, l, u, v, w)
    qap = ctx.qap_sparse_roots(roots, m, l, u, v, w)
    td_ints = [rng.fr() for _ in range(5)]
    td = ints_to_limbs(td_ints)
    td_int = ints_to_limbs(td_ints[:4] + [(td_ints[4] - b) * pow(a, -1, R) % R])
    crs = ctx.setup(qap, td)
    r, s = rng.fr(), rng.fr()
    good = ctx.prove(crs, qap, weights, r, s)
    assert good == orc.trapdoor_proof_integers(desc, n, td_int, weights, r, s)
    assert ctx.verify(crs, [x, limbs_to_int(weights[2])], good)
    bad = weights.copy(); bad[n // 2, 0] ^= np.uint64(1)
    got_bad = ctx.prove(crs, qap, bad, r, s)
    assert got_bad == orc.trapdoor_proof_integers(desc, n, td_int, bad, r, s)
    assert not ctx.verify(crs, [x, limbs_to_int(weights[2])], got_bad)
    hosts = [np.ascontiguousarray(w_) for w_ in (weights, bad, weights)]
    t = [ctx.prove_submit_host(crs, qap, w_.ctypes.data, w_.shape[0], r, s) for w_ in hosts]
    got = [ctx.prove_wait(x_) for x_ in t]
    assert got == [good, got_bad, good]


def test_arbitrary_roots_container_limits_and_errors(ctx, tmp_path):
    rng = SplitMix64(8100)
    n, m, l = 150, 333, 2
    roots = ints_to_limbs(distinct_roots(rng, n)).reshape(n, 4)
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    qap = ctx.qap_sparse_roots(roots, m, l, u, v, w)
    assert ctx.lib.zk_qap_kind(qap.ptr) == 3
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    wts = ints_to_limbs([1] + [rng.fr() for _ in range(m - 1)])
    r, s = rng.fr(), rng.fr()
    want = ctx.prove(crs, qap, wts, r, s)
    path = tmp_path / "arb.zkqap"
    ctx.qap_save(qap, path)
    back = ctx.qap_load(path)
    assert ctx.lib.zk_qap_kind(back.ptr) == 3 and (back.n, back.m, back.input) == (n, m, l)
    assert ctx.prove(crs, back, wts, r, s) == want
    raw = bytearray(open(path, "rb").read())
    raw[-5] ^= 1                                      # a root altered: the checksum catches it
    open(tmp_path / "bad.zkqap", "wb").write(raw)
    with pytest.raises(zk.ZkError) as e:
        ctx.qap_load(tmp_path / "bad.zkqap")
    assert e.value.status == zk._lib.ZK_ERR_IO
    open(tmp_path / "short.zkqap", "wb").write(raw[:-40])
    with pytest.raises(zk.ZkError):
        ctx.qap_load(tmp_path / "short.zkqap")
    dup = roots.copy(); dup[17] = dup[3]
    with pytest.raises(zk.ZkError) as e:
        ctx.qap_sparse_roots(dup, m, l, u, v, w)
    assert e.value.status == zk._lib.ZK_ERR_ARG
    big = roots.copy(); big[5] = ints_to_limbs([R])[0]
    with pytest.raises(zk.ZkError) as e:
        ctx.qap_sparse_roots(big, m, l, u, v, w)
    assert e.value.status == zk._lib.ZK_ERR_RANGE
    # the trapdoor's x on a root: refused like the reference's division by zero would be
    td_bad = td.copy(); td_bad[4] = roots[9]
    with pytest.raises(zk.ZkError):
        ctx.setup(qap, td_bad)
    # a batch of one is a proof
    import torch
    d = torch.from_numpy(wts.view(np.int64)).cuda()
    torch.cuda.synchronize()
    assert ctx.prove_batch_wait(ctx.prove_batch_submit(crs, qap, [d.data_ptr()], [m], [r], [s]), 1) == [ctx.prove(crs, qap, wts, r, s)]


def test_arbitrary_roots_window_and_point_sharded(ctx):
    """zk_prove_partial for every rank of worlds 1, 2, 3, 8 (partial sums by Pippenger windows and by point ranges) + zk_prove_combine ==
    zk_prove: the latency form of the multi-GPU prover takes this QAP form as it is."""
    import torch
    rng = SplitMix64(8200)
    n, m, l = 700, 1500, 2
    roots = ints_to_limbs(distinct_roots(rng, n)).reshape(n, 4)
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    qap = ctx.qap_sparse_roots(roots, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    wts = ints_to_limbs([1] + [rng.fr() for _ in range(m - 1)])
    r, s = rng.fr(), rng.fr()
    want = ctx.prove(crs, qap, wts, r, s)
    dw = torch.from_numpy(wts.view(np.int64)).cuda()
    try:
        for by_points in (0, 1):
            ctx.set_option("msm_shard_points", by_points)
            for world in (1, 2, 3, 8):
                buf = torch.zeros(world * zk.PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
                for rank in range(world):
                    ctx.prove_partial(crs, qap, dw.data_ptr(), m, r, s, rank, world, buf.data_ptr() + rank * zk.PARTIAL_BYTES)
                torch.cuda.synchronize()
                assert ctx.prove_combine(crs, buf.data_ptr(), world, r, s) == want, (by_points, world)
    finally:
        ctx.set_option("msm_shard_points", 0)


@pytest.mark.parametrize("n", [100, 700, 5000])
def test_arbitrary_roots_in_batches_and_in_the_scalar_exchange(ctx, n):
    """Round 4: the arbitrary-roots form is a first-class citizen of every entry point.  Batches (zk_prove_batch_*: whole, truncated and
    unsatisfying witnesses in one batch, two batches in flight) and the scalar exchange (zk_prove_scalars_submit -> all-to-all by
    slicing -> zk_prove_msm_submit per rank over the rank's own window tables -> zk_prove_combine) at worlds 1, 2, 3, 8, all ranks played
    by one device; then the C pipeline (zk_mgpu_*) at world 1.  Every proof == zk_prove's bytes (which the oracle pins above)."""
    import torch
    from zksnark_rs_amd.distributed import Comm, MgpuProver
    rng = SplitMix64(8300 + n)
    m, l = 2 * n + 20, 2
    roots = ints_to_limbs(distinct_roots(rng, n)).reshape(n, 4)
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    qap = ctx.qap_sparse_roots(roots, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    proofs = [(ints_to_limbs([1] + [rng.fr() for _ in range(m - 1)]), rng.fr(), rng.fr()),
              (ints_to_limbs([1] + [rng.fr() for _ in range(m - 4)]), rng.fr(), rng.fr()),     # truncated (zip, mod.rs:233-253)
              (ints_to_limbs([1] + [rng.next() & 1 for _ in range(m - 1)]), rng.fr(), rng.fr())]  # boolean: heavy buckets
    want = [ctx.prove(crs, qap, wt, r, s) for wt, r, s in proofs]
    dws = [torch.from_numpy(np.ascontiguousarray(wt).view(np.int64)).cuda() for wt, _, _ in proofs]
    torch.cuda.synchronize()
    # batches
    args = lambda idx: ([dws[j].data_ptr() for j in idx], [proofs[j][0].shape[0] for j in idx], [proofs[j][1] for j in idx], [proofs[j][2] for j in idx])   # noqa: E731
    t1 = ctx.prove_batch_submit(crs, qap, *args([0, 1, 2]))
    t2 = ctx.prove_batch_submit(crs, qap, *args([2, 0]))
    assert ctx.prove_batch_wait(t1, 3) == want and ctx.prove_batch_wait(t2, 2) == [want[2], want[0]]
    # the scalar exchange, rank by rank
    for world in (1, 2, 3, 8):
        elems = ctx.prove_exchange_elems(qap, world)
        assert all(e % world == 0 for e in elems) and elems[1] >= n and elems[3] >= 2 * n - 1
        send = [[torch.zeros(32 * e, dtype=torch.uint8, device="cuda") for e in elems] for _ in proofs]
        for j, (wt, r, s) in enumerate(proofs):
            ctx.prove_wait(ctx.prove_scalars_submit(crs, qap, dws[j].data_ptr(), wt.shape[0], r, s, world, [x.data_ptr() for x in send[j]]), partial=True)
        blobs = [[None] * world for _ in proofs]
        for g in range(world):
            recv = []
            for k, e in enumerate(elems):
                c = 32 * e // world
                recv.append(torch.cat([send[j][k][g * c:(g + 1) * c] for j in range(len(proofs))]))
            part = torch.zeros(len(proofs) * zk.PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
            ctx.prove_wait(ctx.prove_msm_submit(crs, qap, len(proofs), g, world, [x.data_ptr() for x in recv], part.data_ptr()), partial=True)
            for j in range(len(proofs)):
                blobs[j][g] = part[j * zk.PARTIAL_BYTES:(j + 1) * zk.PARTIAL_BYTES].clone()
        for j, (wt, r, s) in enumerate(proofs):
            assert ctx.prove_combine(crs, torch.cat(blobs[j]).data_ptr(), world, r, s) == want[j], (world, j)
    comm = Comm(ctx, 0, 1)
    mp = MgpuProver(ctx, comm, crs, qap)
    jobs = [(dws[j].data_ptr(), proofs[j][0].shape[0], proofs[j][1], proofs[j][2]) for j in (0, 1, 2, 0)]
    assert list(mp.prove_stream(jobs, ahead=2)) == [want[0], want[1], want[2], want[0]]
    mp.close()
    comm.close()
