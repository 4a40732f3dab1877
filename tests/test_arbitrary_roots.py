"""Arbitrary root sets (RootRepresentation::roots() is caller data, circuit/mod.rs:201-214): interpolation through caller-supplied
nodes by the sub-product tree (csrc/interp.hip), and the sparse QAP form built on it."""
import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd import SplitMix64, ints_to_limbs, limbs_to_int

R = zk.R_MODULUS
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return zk.Context(0)


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


@pytest.mark.parametrize("log_n", [10, 14, 17])
def test_interpolation_on_permuted_roots_of_unity(ctx, log_n):
    """Size-independent check: when the nodes are the 2^k-th roots of unity in a scrambled order the interpolant is the inverse
    transform of the unscrambled values (zk_ntt_fr, pinned to the reference's dft KATs through the oracle)."""
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
