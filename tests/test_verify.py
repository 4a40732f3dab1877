"""groth16::verify and the pairing (host code in the reference and here).

CPU: zk_pairing against the big-int twin (oracle/pyref.py) coefficient for coefficient, plus
bilinearity / non-degeneracy.  GPU: verify accepts honest proofs made by the GPU prover and rejects
wrong public inputs -- the reference's own end-to-end tests (lib.rs:156-190, fr.rs:248-416)."""
import os
import sys

import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd import ints_to_limbs, SplitMix64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
ZK_DIR = os.path.join(ROOT, "tests", "golden", "zk")


def g1_words(P):
    return np.zeros(8, np.uint64) if P is None else ints_to_limbs([P[0], P[1]]).reshape(8)


def g2_words(P):
    return np.zeros(16, np.uint64) if P is None else ints_to_limbs([P[0][0], P[0][1], P[1][0], P[1][1]]).reshape(16)


def test_pairing_matches_bigint_twin():
    import pyref
    for a, b in ((1, 1), (69, 96), (123456789, 987654321)):
        P, Qt = pyref.g1_mul(pyref.G1_GEN, a), pyref.g2_mul(pyref.G2_GEN, b)
        assert zk.pairing(g1_words(P), g2_words(Qt)) == pyref.fq12_flat(pyref.pairing(P, Qt))


def test_pairing_bilinear_and_degenerate():
    import pyref
    one = [1] + [0] * 11
    assert zk.pairing(g1_words(None), g2_words(pyref.G2_GEN)) == one
    assert zk.pairing(g1_words(pyref.G1_GEN), g2_words(None)) == one
    e = zk.pairing(g1_words(pyref.G1_GEN), g2_words(pyref.G2_GEN))
    assert e != one
    a = 0xdeadbeefcafe
    assert zk.pairing(g1_words(pyref.g1_mul(pyref.G1_GEN, a)), g2_words(pyref.G2_GEN)) == \
        zk.pairing(g1_words(pyref.G1_GEN), g2_words(pyref.g2_mul(pyref.G2_GEN, a)))
    with pytest.raises(zk.ZkError):
        zk.pairing(ints_to_limbs([1, 3]).reshape(8), g2_words(pyref.G2_GEN))      # (1,3) is not on the curve


def _fq_sqrt(a):
    import pyref
    s = pow(a, (pyref.Q + 1) // 4, pyref.Q)
    return s if s * s % pyref.Q == a % pyref.Q else None


def _fq2_sqrt(a):
    """square root in Fq[i]/(i^2+1) by the norm method (q = 3 mod 4); None for non-residues"""
    import pyref
    q = pyref.Q
    if a[1] == 0:
        s = _fq_sqrt(a[0])
        if s is not None:
            return (s, 0)
        s = _fq_sqrt(-a[0] % q)
        return None if s is None else (0, s)
    n = _fq_sqrt((a[0] * a[0] + a[1] * a[1]) % q)
    if n is None:
        return None
    inv2 = pow(2, q - 2, q)
    for sg in (n, -n):
        x0 = _fq_sqrt((a[0] + sg) * inv2 % q)
        if x0:
            return (x0, a[1] * pow(2 * x0, q - 2, q) % q)
    return None


def g2_mul_raw(P, k):
    """k * P on the twist WITHOUT reducing k modulo r (pyref.g2_mul reduces: it is only meant for G2)"""
    import pyref
    acc = None
    for bit in bin(k)[2:]:
        acc = pyref.g2_add(acc, acc)
        if bit == "1":
            acc = pyref.g2_add(acc, P)
    return acc


def twist_point_outside_g2(seed):
    """a point of E'(Fq2): y^2 = x^3 + 3/(9+i) that is NOT in the order-r subgroup (the cofactor 2q - r is ~2^254, so a
    random twist point is outside G2 except with negligible probability; checked with [r]P != infinity)"""
    import pyref
    rng = SplitMix64(seed)
    while True:
        x = (rng.fr() % pyref.Q, rng.fr() % pyref.Q)
        y = _fq2_sqrt(pyref.fq2_add(pyref.fq2_mul(pyref.fq2_mul(x, x), x), pyref.B2))
        if y is None:
            continue
        P = (x, y)
        assert pyref.g2_on_curve(P)
        if g2_mul_raw(P, pyref.R) is not None:
            return P


def test_pairing_rejects_twist_point_outside_g2():
    """ADVICE r1 (verify.hip): the ate Miller loop is bilinear only on the r-torsion subgroup of the twist"""
    import pyref
    P = twist_point_outside_g2(11)
    with pytest.raises(zk.ZkError):
        zk.pairing(g1_words(pyref.G1_GEN), g2_words(P))
    # the same coordinates after clearing the cofactor are accepted
    h = 2 * pyref.Q - pyref.R
    Pg = g2_mul_raw(P, h)
    assert Pg is not None and g2_mul_raw(Pg, pyref.R) is None
    assert zk.pairing(g1_words(pyref.G1_GEN), g2_words(Pg)) == pyref.fq12_flat(pyref.pairing(pyref.G1_GEN, Pg))


@pytest.mark.gpu
def test_verify_rejects_non_canonical_and_out_of_subgroup_proofs(ctx):
    """The byte decoder of zk_verify is a trust boundary the reference does not have (it never deserialises proofs):
    tag 0x00 must be followed by zeros only, tag 0x04 must not carry (0, 0), B must lie in G2."""
    import pyref
    from zksnark_rs_amd.circuit import Circuit
    c = Circuit(open(os.path.join(ZK_DIR, "simple.zk")).read())
    weights = c.weights([3, 2, 4])
    qap = c.qap(ctx)
    rng = SplitMix64(78)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    proof = ctx.prove(crs, qap, weights, rng.fr(), rng.fr())
    assert ctx.verify(crs, [2, 34], proof)
    for off, size in ((0, 65), (65, 129), (194, 65)):
        bad = bytearray(proof); bad[off] = 0                       # infinity tag with the old coordinates behind it
        assert not ctx.verify(crs, [2, 34], bytes(bad))
        bad = bytearray(proof); bad[off + 1:off + size] = bytes(size - 1)   # tag 0x04 with (0, 0)
        assert not ctx.verify(crs, [2, 34], bytes(bad))
        bad = bytearray(proof); bad[off] = 2                       # unknown tag
        assert not ctx.verify(crs, [2, 34], bytes(bad))
    P = twist_point_outside_g2(5)
    bad = proof[:65] + pyref.enc_g2(P) + proof[194:]
    assert len(bad) == len(proof) and not ctx.verify(crs, [2, 34], bad)


@pytest.mark.gpu
def test_simple_circuit_test(ctx):
    """lib.rs:156-190: simple.zk, a=3 b=2 c=4 -> verify(b=2, x=34) true, verify(b=2, x=25) false."""
    from zksnark_rs_amd.circuit import Circuit
    c = Circuit(open(os.path.join(ZK_DIR, "simple.zk")).read())
    weights = c.weights([3, 2, 4])
    qap = c.qap(ctx)
    rng = SplitMix64(77)
    for _ in range(2):
        crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
        proof = ctx.prove(crs, qap, weights, rng.fr(), rng.fr())
        assert ctx.verify(crs, [2, 34], proof)
        assert not ctx.verify(crs, [2, 25], proof)
    bad = bytearray(proof); bad[40] ^= 1
    assert not ctx.verify(crs, [2, 34], bytes(bad))


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["lispesque_quad.zk", "lispesque_cubic.zk", "deg_15.zk"])
def test_bn_encrypt_programs_verify(ctx, prog):
    """fr.rs:273-416 (bn_encrypt_quad/cubic/deg_15_test): random inputs, verify(&weights[1..3])."""
    from zksnark_rs_amd.circuit import Circuit
    c = Circuit(open(os.path.join(ZK_DIR, prog)).read())
    rng = SplitMix64(len(prog))
    weights = c.weights([rng.fr() for _ in range(c.n_in)])
    qap = c.qap(ctx)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    proof = ctx.prove(crs, qap, weights, rng.fr(), rng.fr())
    assert ctx.verify(crs, weights[1:3], proof)
    wrong = weights[1:3].copy(); wrong[1, 0] ^= np.uint64(1)
    assert not ctx.verify(crs, wrong, proof)


@pytest.mark.gpu
def test_verify_large_sparse_proof(ctx):
    """2^12-gate chain circuit through the sparse pipeline: the proof verifies; an unsatisfying witness does not."""
    from zksnark_rs_amd.circuits import chain_rows, chain_weights
    log_n = 12
    rng = SplitMix64(4242)
    m, l, u, v, w = chain_rows(log_n)
    weights = chain_weights(log_n, rng.fr(), [rng.fr() for _ in range(1 << log_n)])
    qap = ctx.qap_sparse(log_n, m, l, u, v, w)
    crs = ctx.setup(qap, ints_to_limbs([rng.fr() for _ in range(5)]))
    r, s = rng.fr(), rng.fr()
    assert ctx.verify(crs, weights[1:3], ctx.prove(crs, qap, weights, r, s))
    bad = weights.copy(); bad[5, 0] ^= np.uint64(1)
    assert not ctx.verify(crs, bad[1:3], ctx.prove(crs, qap, bad, r, s))


@pytest.mark.gpu
def test_reference_doc_example(ctx):
    """The crate's doc example (lib.rs:80-114) through the reference-named API."""
    from zksnark_rs_amd import groth16
    code = open(os.path.join(ZK_DIR, "simple.zk")).read()
    qap = groth16.QAP.from_zk(ctx, code)
    w = groth16.weights(code, [3, 2, 4])
    sigmag1, sigmag2 = groth16.setup(qap)
    proof = groth16.prove(qap, (sigmag1, sigmag2), w)
    assert groth16.verify((sigmag1, sigmag2), [2, 34], proof)
    assert not groth16.verify((sigmag1, sigmag2), [2, 25], proof)
    # the CRS halves carry the reference's fields (mod.rs:105-121): simple.zk has 2 gates, 6 wires, 2 verifier inputs
    assert sigmag1.xi.shape == (2, 8) and sigmag1.sum_gamma.shape == (3, 8) and sigmag1.sum_delta.shape == (3, 8) and sigmag1.xi_t.shape == (1, 8)
    assert sigmag2.xi.shape == (2, 16) and sigmag2.gamma.shape == (16,)
    # the same program kept as sparse rows over the roots 1..n: same CRS from the same trapdoor, same proof from the same (r, s)
    td, rs = [11, 12, 13, 14, 15], (21, 22)
    qs = groth16.QAP.from_zk(ctx, code, sparse=True)
    s1, s2 = groth16.setup(qap, td)
    t1, t2 = groth16.setup(qs, td)
    assert np.array_equal(s1.xi_t, t1.xi_t) and np.array_equal(s2.xi, t2.xi) and np.array_equal(s1.sum_delta, t1.sum_delta)
    assert groth16.prove(qs, (t1, t2), w, rs) == groth16.prove(qap, (s1, s2), w, rs)
