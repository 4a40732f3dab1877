"""Sparse QAPs over the roots ASTParser emits, the integers 1..n (circuit/mod.rs:517), at any size -- SURVEY.md 8-f4.

CPU: the oracle's closed form for that domain (oracle/fast.hpp lagrange_at_integers) is pinned to the fixture proofs of
simple.zk and deg_15.zk, i.e. to the faithful coefficient-form restatement of mod.rs:199-290.
GPU: zk_qap_upload_sparse_integers / zk_circuit_qap_sparse give the bytes of the dense (reference-shaped) path wherever that
path runs, and the closed form's bytes beyond it, up to BASELINE's 2^20 gates.
"""
import json
import os

import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd import SplitMix64, ints_to_limbs, limbs_to_int
from zksnark_rs_amd.circuit import Circuit

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROOFS = json.load(open(os.path.join(GOLD, "proofs.json")))
H = lambda s: int(s, 16)   # noqa: E731


def case(name):
    return next(c for c in PROOFS["cases"] if c["name"] == name)


def case_inputs(c):
    return [v if isinstance(v, int) else H(v) for v in c["inputs"]]


def chain_program(n):
    """deg_15.zk generalised to n gates: t1 = x*a1, tk = x*(t(k-1) + ak), y = 1*(t(n-1) + an)."""
    ins = " ".join("a%d" % k for k in range(1, n + 1))
    body = ["    (= t1 (* x a1))"]
    body += ["    (= t%d (* x (+ t%d a%d)))" % (k, k - 1, k) for k in range(2, n)]
    body.append("    (= y (* 1 (+ t%d a%d))))" % (n - 1, n))
    return "(in x %s)\n(out y)\n(verify x y)\n\n(program\n%s\n" % (ins, "\n".join(body))


def random_rows(rng, n, m, density):
    ptr, gates, vals = [0], [], []
    for _ in range(m):
        k = rng.next() % (density + 1)
        gs = sorted({int(rng.next() % n) for _ in range(k)})
        gates += gs
        vals += [rng.fr() for _ in gs]
        ptr.append(len(gates))
    val = ints_to_limbs(vals) if vals else np.zeros((0, 4), np.uint64)
    return np.array(ptr, np.uint64), np.array(gates, np.uint32), val


def chain_rows_integers(n):
    """Rows of chain_program(n) in the parser's own wire order, without going through the parser (2^20 gates of program text
    would be 40 MB): wires 1 | x | y | t_1 a_1 t_2 a_2 .. t_(n-1) a_(n-1) | a_n; gate k: x * (t_(k-1) + a_k) = t_k, the last
    gate 1 * (t_(n-1) + a_n) = y."""
    m = 2 * n + 2
    one = ints_to_limbs([1])[0]
    tw = lambda k: 3 + 2 * (k - 1)                         # noqa: E731  wire of t_k, k = 1..n-1
    aw = lambda k: 4 + 2 * (k - 1) if k < n else 2 * n + 1   # noqa: E731  wire of a_k, k = 1..n
    u = [[] for _ in range(m)]; v = [[] for _ in range(m)]; w = [[] for _ in range(m)]
    for g in range(n):                                     # gate g computes t_(g+1) (y for the last)
        (u[1] if g < n - 1 else u[0]).append(g)
        v[aw(g + 1)].append(g)
        if g > 0:
            v[tw(g)].append(g)
        (w[tw(g + 1)] if g < n - 1 else w[2]).append(g)

    def pack(rows):
        ptr = np.zeros(m + 1, np.uint64)
        ptr[1:] = np.cumsum([len(r) for r in rows])
        gate = np.array([g for r in rows for g in r], np.uint32)
        return ptr, gate, np.tile(one, (len(gate), 1))
    return m, 2, pack(u), pack(v), pack(w)


def chain_weights_integers(n, x, avals):
    P = zk.R_MODULUS
    t, out = 0, [1, x, 0]
    for k in range(n - 1):
        t = x * ((t + avals[k]) % P) % P
        out += [t, avals[k]]
    out[2] = (t + avals[n - 1]) % P
    return ints_to_limbs(out + [avals[n - 1]])


# ---------------------------------------------------------------- CPU
@pytest.mark.parametrize("name", ["simple.zk", "deg_15.zk"])
def test_closed_form_for_integer_roots_reproduces_fixture_proofs(orc, name):
    c = case(name)
    circ = Circuit(open(os.path.join(GOLD, "zk", name)).read())
    weights = circ.weights(case_inputs(c))
    assert [limbs_to_int(w) for w in weights] == [H(w) for w in c["weights"]]
    desc = zk.Context.sparse_desc(0, circ.m, circ.input, circ.rows(0), circ.rows(1), circ.rows(2))
    td = ints_to_limbs([H(t) for t in c["trapdoor"]])
    assert orc.trapdoor_proof_integers(desc, circ.n, td, weights, H(c["r"]), H(c["s"])).hex() == c["proof"]
    bad = weights.copy(); bad[2, 0] ^= np.uint64(1)
    q = orc.zk_qap_dense(open(os.path.join(GOLD, "zk", name)).read())
    want_bad = orc.trapdoor_proof_dense(q["u"], q["v"], q["w"], q["t"], q["input"], td, bad, H(c["r"]), H(c["s"]))
    assert orc.trapdoor_proof_integers(desc, circ.n, td, bad, H(c["r"]), H(c["s"])) == want_bad


def test_chain_rows_helper_matches_the_parser():
    n = 9
    circ = Circuit(chain_program(n))
    m, l, u, v, w = chain_rows_integers(n)
    assert (circ.n, circ.m, circ.input) == (n, m, l)
    for k, mine in enumerate((u, v, w)):
        ptr, gate, val = circ.rows(k)
        assert np.array_equal(ptr, mine[0]) and np.array_equal(gate, mine[1]) and np.array_equal(val, mine[2]), k
    rng = SplitMix64(5)
    ins = [rng.fr() for _ in range(n + 1)]
    assert np.array_equal(circ.weights(ins), chain_weights_integers(n, ins[0], ins[1:]))


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["simple.zk", "deg_15.zk"])
def test_gpu_sparse_form_reproduces_fixture_proofs(ctx, name):
    c = case(name)
    circ = Circuit(open(os.path.join(GOLD, "zk", name)).read())
    weights = circ.weights(case_inputs(c))
    qap = circ.qap_sparse(ctx)
    crs = ctx.setup(qap, ints_to_limbs([H(t) for t in c["trapdoor"]]))
    proof = ctx.prove(crs, qap, weights, H(c["r"]), H(c["s"]))
    assert proof.hex() == c["proof"]
    assert ctx.verify(crs, [v if isinstance(v, int) else H(v) for v in c["verify_inputs"]], proof)


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["simple.zk", "lispesque_quad.zk", "lispesque_cubic.zk", "deg_15.zk"])
def test_gpu_sparse_form_matches_faithful_oracle(ctx, orc, prog):
    """same CRS arrays and same proof bytes as the reference's coefficient-form path (mod.rs:134-290) on QAP::from(root_rep)"""
    code = open(os.path.join(GOLD, "zk", prog)).read()
    q = orc.zk_qap_dense(code)
    circ = Circuit(code)
    rng = SplitMix64(77 + len(code))
    weights = circ.weights([rng.fr() for _ in range(q["n_in"])])
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    qap = circ.qap_sparse(ctx)
    crs = ctx.setup(qap, td)
    arrs = ctx.crs_download(crs)
    want = orc.setup_dense(q["u"], q["v"], q["w"], q["t"], q["input"], td)
    for k in want:
        assert np.array_equal(arrs[k], want[k]), k
    cdesc = ctx.crs_desc(q["n"], q["m"], q["input"], arrs)
    bad = weights.copy(); bad[2, 0] += np.uint64(1)
    for wts in (weights, bad, weights[:-1], np.concatenate([weights, weights[:2]])):
        assert ctx.prove(crs, qap, wts, r, s) == orc.prove_dense(q["u"], q["v"], q["w"], q["t"], q["input"], cdesc, wts, r, s)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 3, 5, 600, 1025, 4096])
def test_gpu_sparse_form_equals_dense_form(ctx, n):
    """the two device forms of one ASTParser circuit: identical CRS, identical bytes for a valid and an invalid witness"""
    circ = Circuit(chain_program(n))
    rng = SplitMix64(9100 + n)
    weights = circ.weights([rng.fr() for _ in range(n + 1)])
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    qd, qs = circ.qap(ctx), circ.qap_sparse(ctx)
    cd, cs = ctx.setup(qd, td), ctx.setup(qs, td)
    ad, as_ = ctx.crs_download(cd), ctx.crs_download(cs)
    for k in ad:
        assert np.array_equal(ad[k], as_[k]), k
    r, s = rng.fr(), rng.fr()
    bad = weights.copy(); bad[5 % circ.m, 0] ^= np.uint64(1)
    good = ctx.prove(cs, qs, weights, r, s)
    assert good == ctx.prove(cd, qd, weights, r, s)
    assert ctx.prove(cs, qs, bad, r, s) == ctx.prove(cd, qd, bad, r, s)
    pub = [limbs_to_int(weights[1]), limbs_to_int(weights[2])]
    assert ctx.verify(cs, pub, good)
    # pipelined submissions of the sparse form (slots alternate main streams)
    hosts = [np.ascontiguousarray(w) for w in (weights, bad, weights)]
    t = [ctx.prove_submit_host(cs, qs, w.ctypes.data, w.shape[0], r, s) for w in hosts]
    got = [ctx.prove_wait(x) for x in t]
    assert got[0] == good and got[2] == good and got[1] != good


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,l", [(1, 4, 1), (2, 5, 0), (7, 20, 3), (100, 260, 5), (1000, 1500, 2), (4099, 3000, 1)])
def test_gpu_random_rows_match_closed_form(ctx, orc, n, m, l):
    """arbitrary rows (empty wires, m unrelated to n, n not a power of two), unsatisfying witnesses of three lengths"""
    rng = SplitMix64(9300 + n)
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    desc = ctx.sparse_desc(0, m, l, u, v, w)
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    r, s = rng.fr(), rng.fr()
    for count in (m, max(l + 1, m - 3), m + 2):
        wts = ints_to_limbs([1] + [rng.fr() for _ in range(count - 1)])
        assert ctx.prove(crs, qap, wts, r, s) == orc.trapdoor_proof_integers(desc, n, td, wts, r, s), count


@pytest.mark.gpu
@pytest.mark.parametrize("n", [(1 << 16) + 3, 1 << 20])
def test_gpu_chain_circuit_at_full_size(ctx, orc, n):
    """BASELINE's 2^20 gates with the roots the reference's parser would give that circuit (dense form: 3 m n field elements
    = 211 TB)"""
    m, l, u, v, w = chain_rows_integers(n)
    rng = SplitMix64(9500 + (n & 0xFFFF))
    x, avals = rng.fr(), [rng.fr() for _ in range(n)]
    weights = chain_weights_integers(n, x, avals)
    desc = ctx.sparse_desc(0, m, l, u, v, w)
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    r, s = rng.fr(), rng.fr()
    good = ctx.prove(crs, qap, weights, r, s)
    assert good == orc.trapdoor_proof_integers(desc, n, td, weights, r, s)
    assert ctx.verify(crs, [x, limbs_to_int(weights[2])], good)
    bad = weights.copy(); bad[n // 2, 0] ^= np.uint64(1)
    got_bad = ctx.prove(crs, qap, bad, r, s)
    assert got_bad == orc.trapdoor_proof_integers(desc, n, td, bad, r, s)
    assert not ctx.verify(crs, [x, limbs_to_int(weights[2])], got_bad)


def ctx_bytes_of_crs(ctx, crs, tmp_path):
    ctx.crs_save(crs, tmp_path / "ref.zkcrs")
    return (tmp_path / "ref.zkcrs").read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["simple.zk", "deg_15.zk"])
def test_gpu_uploaded_crs_serves_integer_roots_programs(ctx, prog):
    """groth16::prove takes any (&SigmaG1, &SigmaG2) (mod.rs:213-217): the CRS of the fixture proofs, uploaded as the reference's arrays
    alone (zk_crs_upload), with the program's rows over ASTParser's roots 1..n (zk_qap_upload_sparse_integers) gives the fixture bytes --
    the Lagrange-basis points are derived from the powers (csrc/basis.hip)."""
    c = case(prog)
    circ = Circuit(open(os.path.join(GOLD, "zk", prog)).read())
    weights = circ.weights(case_inputs(c))
    td = ints_to_limbs([H(t) for t in c["trapdoor"]])
    r, s = H(c["r"]), H(c["s"])
    crs_dense = ctx.setup(circ.qap(ctx), td)                                   # the reference's arrays (power basis) ...
    crs_up = ctx.crs_upload(circ.n, circ.m, circ.input, ctx.crs_download(crs_dense))   # ... and nothing else
    assert ctx.prove(crs_up, circ.qap_sparse(ctx), weights, r, s).hex() == c["proof"]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096, 65539])
def test_gpu_uploaded_crs_change_of_basis_at_size(ctx, orc, tmp_path, n):
    """The same at 4096 and 2^16 + 3 gates: zk_crs_upload + zk_qap_upload_sparse_integers + zk_prove == the zk_setup path == the closed
    form, and the derived Lagrange-basis arrays equal the ones zk_setup wrote (the CRS containers are identical byte for byte)."""
    rng = SplitMix64(8800 + n)
    m, l, u, v, w = chain_rows_integers(n)
    x = rng.fr()
    weights = chain_weights_integers(n, x, [rng.fr() for _ in range(n)])
    desc = ctx.sparse_desc(0, m, l, u, v, w)
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    r, s = rng.fr(), rng.fr()
    good = ctx.prove(crs, qap, weights, r, s)
    assert good == orc.trapdoor_proof_integers(desc, n, td, weights, r, s)
    crs_up = ctx.crs_upload(n, m, l, ctx.crs_download(crs))
    assert ctx.prove(crs_up, qap, weights, r, s) == good
    # the derived Lagrange-basis arrays are the ones zk_setup wrote: the two CRS containers (ZKCRSv2) agree byte for byte
    ctx.crs_save(crs, tmp_path / "setup.zkcrs")
    ctx.crs_save(crs_up, tmp_path / "derived.zkcrs")
    a, b = (tmp_path / "setup.zkcrs").read_bytes(), (tmp_path / "derived.zkcrs").read_bytes()
    assert a[:8] == b"ZKCRSv2\0" and a == b


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 5, 64, 65, 200, 1000])
def test_gpu_change_of_basis_by_the_transposed_tree(ctx, tmp_path, n):
    """csrc/gbasis.hip: the Lagrange-basis points of an uploaded CRS as the TRANSPOSE of the interpolation tree run over curve points
    (O(n log^2 n) point operations; the default from 512 gates on).  Forced at small sizes here -- one block, one level, ragged last
    block, several levels --: the derived arrays equal the ones zk_setup wrote from the trapdoor, and the ones of the n^2 inner
    products (the three CRS containers agree byte for byte)."""
    rng = SplitMix64(8900 + n)
    m, l, u, v, w = chain_rows_integers(n)
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    weights = chain_weights_integers(n, rng.fr(), [rng.fr() for _ in range(n)])
    r, s = rng.fr(), rng.fr()
    good = ctx.prove(crs, qap, weights, r, s)
    ctx.crs_save(crs, tmp_path / "setup.zkcrs")
    want = (tmp_path / "setup.zkcrs").read_bytes()
    old = ctx.get_option("basis_tree_min")
    try:
        for name, tree_min in (("tree", 1), ("squares", 1 << 30)):
            ctx.set_option("basis_tree_min", tree_min)
            up = ctx.crs_upload(n, m, l, ctx.crs_download(crs))
            assert ctx.prove(up, qap, weights, r, s) == good, name
            ctx.crs_save(up, tmp_path / (name + ".zkcrs"))
            assert (tmp_path / (name + ".zkcrs")).read_bytes() == want, name
    finally:
        ctx.set_option("basis_tree_min", old)


@pytest.mark.gpu
def test_gpu_uploaded_crs_beyond_the_change_of_basis(ctx, orc):
    """Above 2^16 + 2^10 gates the O(n^2) change of basis is not attempted.  With the transposed tree switched off (option basis_tree_min
    < 0: the round-3 behaviour) an integer-roots QAP over a CRS that carries only the reference's arrays proves through the sub-product
    tree of the roots 1..n instead (csrc/arbroots.hip) -- the same bytes, which the closed form pins, from the uploaded CRS and again
    from the CRS zk_setup made (evaluation basis).  The default (csrc/gbasis.hip) is covered by test_gpu_change_of_basis_at_2_17."""
    ctx.set_option("basis_tree_min", -1)
    try:
        _beyond_the_n2_change_of_basis(ctx, orc, True)
    finally:
        ctx.set_option("basis_tree_min", 16384)


@pytest.mark.gpu
def test_gpu_change_of_basis_at_2_17(ctx, orc):
    """The default above 16384 gates: the Lagrange-basis points by the transposed interpolation tree (csrc/gbasis.hip, seconds at this
    size, O(n log^2 n)), after which the uploaded CRS proves in the evaluation basis like the one zk_setup wrote: closed-form bytes,
    valid and invalid witness."""
    assert ctx.get_option("basis_tree_min") == 16384
    _beyond_the_n2_change_of_basis(ctx, orc, False)


def _beyond_the_n2_change_of_basis(ctx, orc, others):
    n = (1 << 17) + 5
    m, l, u, v, w = chain_rows_integers(n)
    rng = SplitMix64(9650)
    x, avals = rng.fr(), [rng.fr() for _ in range(n)]
    weights = chain_weights_integers(n, x, avals)
    desc = ctx.sparse_desc(0, m, l, u, v, w)
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    up = ctx.crs_upload(n, m, l, ctx.crs_download(crs))
    r, s = rng.fr(), rng.fr()
    want = orc.trapdoor_proof_integers(desc, n, td, weights, r, s)
    assert ctx.prove(up, qap, weights, r, s) == want
    assert ctx.prove(crs, qap, weights, r, s) == want
    bad = weights.copy(); bad[n // 3, 0] ^= np.uint64(1)
    want_bad = orc.trapdoor_proof_integers(desc, n, td, bad, r, s)
    assert ctx.prove(up, qap, bad, r, s) == want_bad
    if not others:
        return
    # round 4: every entry point decides the form in one place (prove_form) -- a batch and the multi-GPU pipeline over the uploaded CRS
    # take the same fall-back and give the same bytes (they used to answer ZK_ERR_UNSUPPORTED after the scalars had been exchanged)
    torch = pytest.importorskip("torch")
    from zksnark_rs_amd.distributed import Comm, MgpuProver
    dw, db = (torch.from_numpy(np.ascontiguousarray(x_).view(np.int64)).cuda() for x_ in (weights, bad))
    torch.cuda.synchronize()
    t = ctx.prove_batch_submit(up, qap, [dw.data_ptr(), db.data_ptr()], [m, m], [r, r], [s, s])
    assert ctx.prove_batch_wait(t, 2) == [want, want_bad]
    comm = Comm(ctx, 0, 1)
    mp = MgpuProver(ctx, comm, up, qap)
    assert list(mp.prove_stream([(dw.data_ptr(), m, r, s), (db.data_ptr(), m, r, s)], ahead=1)) == [want, want_bad]
    mp.close()
    comm.close()


@pytest.mark.gpu
def test_gpu_integer_roots_limits_and_errors(ctx, tmp_path):
    n = 12
    circ = Circuit(chain_program(n))
    rng = SplitMix64(9700)
    weights = circ.weights([rng.fr() for _ in range(n + 1)])
    qap = circ.qap_sparse(ctx)
    td = ints_to_limbs([rng.fr() for _ in range(5)])
    crs = ctx.setup(qap, td)
    r, s = rng.fr(), rng.fr()
    good = ctx.prove(crs, qap, weights, r, s)
    # a CRS that carries only the reference's arrays (zk_crs_upload): the plain container keeps the v1 format ...
    crs2 = ctx.crs_upload(circ.n, circ.m, circ.input, ctx.crs_download(crs))
    ctx.crs_save(crs2, tmp_path / "plain.zkcrs")
    assert (tmp_path / "plain.zkcrs").read_bytes()[:8] == b"ZKCRSv1\0"
    # ... serves the dense form of the same circuit, with the same bytes ...
    assert ctx.prove(crs2, circ.qap(ctx), weights, r, s) == good
    # ... and this form after the change of basis of its points (csrc/basis.hip, once per CRS), which a saved copy then carries
    assert ctx.prove(crs2, qap, weights, r, s) == good
    ctx.crs_save(crs2, tmp_path / "based.zkcrs")
    assert (tmp_path / "based.zkcrs").read_bytes() == ctx_bytes_of_crs(ctx, crs, tmp_path)
    # x among 1..2n-1: the Lagrange denominators vanish
    with pytest.raises(zk.ZkError) as e:
        ctx.setup(qap, ints_to_limbs([5, 6, 7, 8, 2 * n - 1]))
    assert e.value.status == -7
    # size limit
    one = (np.zeros(3, np.uint64), np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64))
    with pytest.raises(zk.ZkError) as e:
        ctx.qap_sparse_integers((1 << 23) + 1, 2, 0, one, one, one)
    assert e.value.status == -4
    # CRS file: the Lagrange-basis arrays travel with it (ZKCRSv2), so a reloaded CRS serves this form again
    ctx.crs_save(crs, tmp_path / "c.zkcrs")
    raw = (tmp_path / "c.zkcrs").read_bytes()
    assert raw[:8] == b"ZKCRSv2\0" and len(raw) == 40 + 64 * (3 + n + 3 + (circ.m - 3) + (n - 1)) + 128 * (3 + n) + 64 * (2 * n - 1) + 128 * n
    crs3 = ctx.crs_load(tmp_path / "c.zkcrs")
    assert ctx.prove(crs3, qap, weights, r, s) == good
    assert ctx.prove(crs3, circ.qap(ctx), weights, r, s) == good
    (tmp_path / "bad.zkcrs").write_bytes(raw[:-9] + bytes([raw[-9] ^ 1]) + raw[-8:])
    with pytest.raises(zk.ZkError) as e:
        ctx.crs_load(tmp_path / "bad.zkcrs")
    assert e.value.status == -8
    # container: kind 2 round trip
    ctx.qap_save(qap, tmp_path / "c.zkqap")
    q2 = ctx.qap_load(tmp_path / "c.zkqap")
    assert (q2.n, q2.m, q2.input, q2.dense) == (n, circ.m, circ.input, False)
    assert ctx.prove(ctx.setup(q2, td), q2, weights, r, s) == good


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 37, 1000])
def test_gpu_integer_roots_on_several_ranks(ctx, n):
    """the multi-GPU data paths for this QAP form, all ranks of worlds 1..8 played by one device: the scalar exchange
    (zk_prove_scalars_submit -> all-to-all by slicing -> zk_prove_msm_submit per rank -> zk_prove_combine), the latency form
    (zk_prove_partial by windows and by point ranges) and the C pipeline at world 1"""
    torch = pytest.importorskip("torch")
    from zksnark_rs_amd.distributed import Comm, MgpuProver
    rng = SplitMix64(9900 + n)
    m, l = 3 * n + 4, min(2, 3 * n + 3)
    u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
    qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    proofs = [(ints_to_limbs([1] + [rng.fr() for _ in range(m - 1)]), rng.fr(), rng.fr()),
              (ints_to_limbs([1] + [rng.fr() for _ in range(max(l + 1, m - 3) - 1)]), rng.fr(), rng.fr())]     # the second one truncated
    want = [ctx.prove(crs, qap, wt, r, s) for wt, r, s in proofs]
    dws = [torch.from_numpy(np.ascontiguousarray(wt).view(np.int64)).cuda() for wt, _, _ in proofs]
    torch.cuda.synchronize()
    for world in (1, 2, 3, 8):
        elems = ctx.prove_exchange_elems(qap, world)
        assert all(e % world == 0 for e in elems) and elems[1] >= n and elems[3] >= 2 * n - 1
        send = [[torch.zeros(32 * e, dtype=torch.uint8, device="cuda") for e in elems] for _ in proofs]
        for j, (wt, r, s) in enumerate(proofs):
            t = ctx.prove_scalars_submit(crs, qap, dws[j].data_ptr(), wt.shape[0], r, s, world, [x.data_ptr() for x in send[j]])
            ctx.prove_wait(t, partial=True)
        blobs = [[None] * world for _ in proofs]
        for g in range(world):
            recv = []
            for k, e in enumerate(elems):
                c = 32 * e // world
                recv.append(torch.cat([send[j][k][g * c:(g + 1) * c] for j in range(len(proofs))]))
            part = torch.zeros(len(proofs) * zk.PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
            t = ctx.prove_msm_submit(crs, qap, len(proofs), g, world, [x.data_ptr() for x in recv], part.data_ptr())
            ctx.prove_wait(t, partial=True)
            for j in range(len(proofs)):
                blobs[j][g] = part[j * zk.PARTIAL_BYTES:(j + 1) * zk.PARTIAL_BYTES].clone()
        for j, (wt, r, s) in enumerate(proofs):
            assert ctx.prove_combine(crs, torch.cat(blobs[j]).data_ptr(), world, r, s) == want[j], (world, j)
    wt, r, s = proofs[0]
    for by_points in (0, 1):
        ctx.set_option("msm_shard_points", by_points)
        try:
            for world in (2, 3):
                buf = torch.zeros(world * zk.PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
                for rank in range(world):
                    ctx.prove_partial(crs, qap, dws[0].data_ptr(), m, r, s, rank, world, buf.data_ptr() + rank * zk.PARTIAL_BYTES)
                torch.cuda.synchronize()
                assert ctx.prove_combine(crs, buf.data_ptr(), world, r, s) == want[0], (by_points, world)
        finally:
            ctx.set_option("msm_shard_points", 0)
    comm = Comm(ctx, 0, 1)
    mp = MgpuProver(ctx, comm, crs, qap)
    assert list(mp.prove_stream([(dws[0].data_ptr(), m, r, s)] * 3, ahead=2)) == [want[0]] * 3
    mp.close(); comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 16, 100, 3000])
def test_gpu_integer_roots_batches(ctx, n):
    """zk_prove_batch_* on this QAP form (many proofs of one small ASTParser circuit): == zk_prove one by one, with a truncated
    and an all-zero witness in the batch, batches of 1, 3 and 5, two batches in flight"""
    torch = pytest.importorskip("torch")
    rng = SplitMix64(9950 + n)
    if n == 16:
        circ = Circuit(open(os.path.join(GOLD, "zk", "deg_15.zk")).read())
        qap, m, l = circ.qap_sparse(ctx), circ.m, circ.input
        assert circ.n == 16
    else:
        m, l = 2 * n + 5, 1
        u, v, w = (random_rows(rng, n, m, 3) for _ in range(3))
        qap = ctx.qap_sparse_integers(n, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    wits = [ints_to_limbs([1] + [rng.fr() for _ in range(m - 1)]) for _ in range(5)]
    wits[2] = wits[2][:max(l + 1, m - 2)]
    wits[3] = np.zeros_like(wits[3])
    rs = [rng.fr() for _ in wits]
    ss = [rng.fr() for _ in wits]
    want = [ctx.prove(crs, qap, wt, r, s) for wt, r, s in zip(wits, rs, ss)]
    dws = [torch.from_numpy(np.ascontiguousarray(wt).view(np.int64)).cuda() for wt in wits]
    torch.cuda.synchronize()
    for count in (1, 3, 5):
        t = ctx.prove_batch_submit(crs, qap, [d.data_ptr() for d in dws[:count]], [wt.shape[0] for wt in wits[:count]], rs[:count], ss[:count])
        assert ctx.prove_batch_wait(t, count) == want[:count], count
    t1 = ctx.prove_batch_submit(crs, qap, [d.data_ptr() for d in dws[:2]], [wt.shape[0] for wt in wits[:2]], rs[:2], ss[:2])
    t2 = ctx.prove_batch_submit(crs, qap, [d.data_ptr() for d in dws[2:]], [wt.shape[0] for wt in wits[2:]], rs[2:], ss[2:])
    assert ctx.prove_batch_wait(t1, 2) == want[:2] and ctx.prove_batch_wait(t2, 3) == want[2:]
