"""-m gpu: parity of the HIP building blocks (through the C ABI) against the CPU oracle.

Bit-exact: everything here is integer arithmetic mod the BN254 primes.
Mirrors: FrLocal ops (fr.rs:18-71), exp_encrypted_g1/g2 (fr.rs:114-119; reference test
exp_encrypted_test fr.rs:240-246), Sum/Add for G1Local/G2Local (fr.rs:175-223),
field::dft/idft (field/mod.rs:508-537; reference tests dft_test/idft_test :606-635),
the SigmaG1/SigmaG2 inner products of prove (mod.rs:255-272).
"""
import os

import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd import SplitMix64, ints_to_limbs, limbs_to_int, R_MODULUS, Q_MODULUS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def rand_fr(rng, n):
    return ints_to_limbs([rng.fr() for _ in range(n)])


def rand_fq(rng, n):
    return ints_to_limbs([rng.fr() % Q_MODULUS for _ in range(n)])


def edge_values(p):
    return ints_to_limbs([0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 253) % p, (1 << 128) - 1, 1 << 32, (1 << 64) - 1])


@pytest.mark.parametrize("field", ["fr", "fq"])
@pytest.mark.parametrize("op", ["add", "sub", "mul", "inv", "inv_euclid", "inv_divsteps"])
def test_field_ops(ctx, orc, field, op):
    rng = SplitMix64(11)
    p = R_MODULUS if field == "fr" else Q_MODULUS
    a = np.concatenate([edge_values(p), rand_fr(rng, 500) if field == "fr" else rand_fq(rng, 500)])
    b = np.concatenate([edge_values(p)[::-1], rand_fr(rng, 500) if field == "fr" else rand_fq(rng, 500)])
    if op.startswith("inv"):
        a = a[1:]          # 0 has no inverse
        b = None
    g = getattr(ctx, field + "_batch")(op, a, b)
    rc, o = getattr(orc, field + "_batch")("inv" if op.startswith("inv") else op, a, b)
    assert rc == 0
    assert np.array_equal(g, o)


def test_field_errors(ctx):
    zero = ints_to_limbs([5, 0, 7])
    with pytest.raises(zk.ZkError) as e:
        ctx.fr_batch("inv", zero)           # FrLocal::mul_inv panics on zero (fr.rs:69)
    assert e.value.status == zk._lib.ZK_ERR_DIV_BY_ZERO
    with pytest.raises(zk.ZkError) as e:
        ctx.fr_batch("add", ints_to_limbs([R_MODULUS]), ints_to_limbs([1]))
    assert e.value.status == zk._lib.ZK_ERR_RANGE


def g1_points(orc, rng, n):
    base = np.tile(orc.enc_base_g1(), (n, 1))
    return orc.g1_mul_batch(base, rand_fr(rng, n))


def g2_points(orc, rng, n):
    base = np.tile(orc.enc_base_g2(), (n, 1))
    return orc.g2_mul_batch(base, rand_fr(rng, n))


def test_point_mul_and_add(ctx, orc):
    rng = SplitMix64(5)
    n = 24
    p1, p2 = g1_points(orc, rng, n), g2_points(orc, rng, n)
    k = rand_fr(rng, n)
    k[0] = 0; k[1] = ints_to_limbs([1])[0]; k[2] = ints_to_limbs([R_MODULUS - 1])[0]
    assert np.array_equal(ctx.g1_mul_batch(p1, k), orc.g1_mul_batch(p1, k))
    assert np.array_equal(ctx.g2_mul_batch(p2, k), orc.g2_mul_batch(p2, k))
    # additions incl. P+P, P+(-P), P+inf, inf+inf
    q1, q2 = g1_points(orc, rng, n), g2_points(orc, rng, n)
    q1[0] = p1[0]; q2[0] = p2[0]
    neg = ints_to_limbs([R_MODULUS - 1])
    q1[1] = orc.g1_mul_batch(p1[1:2], neg)[0]; q2[1] = orc.g2_mul_batch(p2[1:2], neg)[0]
    q1[2] = 0; q2[2] = 0
    p1[3] = 0; p2[3] = 0
    p1[4] = 0; q1[4] = 0; p2[4] = 0; q2[4] = 0
    assert np.array_equal(ctx.g1_add_batch(p1, q1), orc.g1_add_batch(p1, q1))
    assert np.array_equal(ctx.g2_add_batch(p2, q2), orc.g2_add_batch(p2, q2))


def test_exp_encrypted(ctx, orc):
    """fr.rs:240-246: a.exp_encrypted_g1(b.encrypt_g1()) == (a*b).encrypt_g1()"""
    rng = SplitMix64(6)
    n = 8
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    base1 = np.tile(orc.enc_base_g1(), (n, 1))
    lhs = ctx.g1_mul_batch(ctx.g1_mul_batch(base1, b), a)
    rhs = ctx.g1_mul_batch(base1, ctx.fr_batch("mul", a, b))
    assert np.array_equal(lhs, rhs)


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 8])
def test_ntt_matches_reference_dft(ctx, orc, log_n):
    """GPU NTT == the reference's naive O(n^2) dft/idft restated over Fr (field/mod.rs:508-537)."""
    rng = SplitMix64(100 + log_n)
    a = rand_fr(rng, 1 << log_n)
    w = orc.root_of_unity(log_n)
    f = ctx.ntt_fr(a)
    assert np.array_equal(f, orc.dft_fr(a, w))
    assert np.array_equal(ctx.ntt_fr(f, inverse=True), a)
    assert np.array_equal(ctx.ntt_fr(a, inverse=True), orc.dft_fr(a, w, inverse=True))


@pytest.mark.parametrize("log_n", [10, 11, 12, 13, 16])
@pytest.mark.parametrize("coset", [False, True])
def test_ntt_matches_fast_oracle(ctx, orc, log_n, coset):
    rng = SplitMix64(200 + log_n)
    a = rand_fr(rng, 1 << log_n)
    f = ctx.ntt_fr(a, coset=coset)
    assert np.array_equal(f, orc.ntt_fr(a, coset=coset))
    assert np.array_equal(ctx.ntt_fr(f, inverse=True, coset=coset), a)


def test_ntt_full_size_properties(ctx):
    """2^20 (BASELINE size): round trip + linearity + a known transform (delta -> all ones)."""
    log_n = 20
    n = 1 << log_n
    rng = np.random.default_rng(1)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    fa = ctx.ntt_fr(a)
    assert np.array_equal(ctx.ntt_fr(fa, inverse=True), a)
    b = np.roll(a, 1, axis=0)
    fb = ctx.ntt_fr(b)
    assert np.array_equal(ctx.ntt_fr(ctx.fr_batch("add", a, b)), ctx.fr_batch("add", fa, fb))
    delta = np.zeros((n, 4), np.uint64); delta[0, 0] = 1
    ones = np.zeros((n, 4), np.uint64); ones[:, 0] = 1
    assert np.array_equal(ctx.ntt_fr(delta), ones)


@pytest.mark.parametrize("log_n", [23, 24])
def test_ntt_three_passes(ctx, orc, log_n):
    """sizes above 2^22 take one more column pass (ntt.hip): == the oracle's NTT at 2^23 (plain and coset), round trips, a
    delta -> all ones, a shifted delta -> the powers of w at sampled positions"""
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    fa = ctx.ntt_fr(a)
    assert np.array_equal(ctx.ntt_fr(fa, inverse=True), a)
    if log_n == 23:
        assert np.array_equal(fa, orc.ntt_fr(a))
        fc = ctx.ntt_fr(a, coset=True)
        assert np.array_equal(fc, orc.ntt_fr(a, coset=True))
        assert np.array_equal(ctx.ntt_fr(fc, inverse=True, coset=True), a)
    del fa
    delta = np.zeros((n, 4), np.uint64); delta[0, 0] = 1
    ones = np.zeros((n, 4), np.uint64); ones[:, 0] = 1
    assert np.array_equal(ctx.ntt_fr(delta), ones)
    delta[0, 0] = 0; delta[1, 0] = 1            # e_1 -> (w^j)_j
    f1 = ctx.ntt_fr(delta)
    w = zk.limbs_to_int(orc.root_of_unity(log_n))
    for j in (0, 1, 2, 12345, (1 << 22) - 1, 1 << 22, (1 << 22) + 7, n // 2 + 3, n - 1):
        assert zk.limbs_to_int(f1[j]) == pow(w, j, R_MODULUS), j


@pytest.mark.parametrize("n,c", [(0, 0), (1, 0), (2, 3), (17, 4), (100, 0), (300, 7), (1000, 11), (1000, 13), (1000, 16), (3000, 17), (3000, 20), (500, 22)])
def test_msm_matches_reference_sum(ctx, orc, n, c):
    """GPU Pippenger == n double-and-add multiplications folded sequentially (mod.rs:255-272)."""
    rng = SplitMix64(300 + n)
    p1, p2, k = g1_points(orc, rng, n), g2_points(orc, rng, n), rand_fr(rng, n)
    if n >= 17:   # zero / one / r-1 scalars, infinity and repeated points
        k[0] = 0; k[1] = ints_to_limbs([1])[0]; k[2] = ints_to_limbs([R_MODULUS - 1])[0]
        p1[3] = 0; p2[3] = 0
        p1[5] = p1[4]; p2[5] = p2[4]; k[5] = k[4]
        p1[7] = p1[6]; p2[7] = p2[6]; k[7] = ints_to_limbs([(R_MODULUS - int(zk.limbs_to_int(k[6]))) % R_MODULUS])[0]
    assert np.array_equal(ctx.msm_g1(p1, k, c), orc.msm_g1(p1, k, 0))
    assert np.array_equal(ctx.msm_g2(p2, k, c), orc.msm_g2(p2, k, 0))


@pytest.mark.parametrize("n,c", [(1, 2), (17, 3), (300, 5), (1000, 8), (3000, 9), (70000, 9), (5000, 10)])
def test_msm_lds_bucket_form(ctx, orc, n, c):
    """The Pippenger form BASELINE.json's north_star words (one wavefront per (window, chunk), buckets in LDS under per-bucket locks,
    wave-level fold; csrc/msm_lds.hpp: zk_msm_g1 with window_bits = -c) gives the same point as the folded double-and-add --
    zero / one / r-1 scalars, infinity, repeated points and P + (-P) in one bucket included.  It is a measured comparator only."""
    rng = SplitMix64(900 + n)
    p1, k = g1_points(orc, rng, n), rand_fr(rng, n)
    if n >= 17:
        k[0] = 0; k[1] = ints_to_limbs([1])[0]; k[2] = ints_to_limbs([R_MODULUS - 1])[0]
        p1[3] = 0
        p1[5] = p1[4]; k[5] = k[4]
        p1[7] = p1[6]; k[7] = ints_to_limbs([(R_MODULUS - int(zk.limbs_to_int(k[6]))) % R_MODULUS])[0]
    assert np.array_equal(ctx.msm_g1(p1, k, -c), orc.msm_g1(p1, k, 0))
    if n == 17:
        with pytest.raises(zk.ZkError):
            ctx.msm_g1(p1, k, -11)        # the buckets of a wider window do not fit LDS (c = 10: 76 KB, two waves per compute unit)
        with pytest.raises(zk.ZkError):
            ctx.msm_g2(g2_points(orc, rng, n), k, -5)   # G1 only


@pytest.mark.parametrize("c", [0, 6, 13])
def test_msm_skewed_scalars(ctx, orc, c):
    """Boolean-heavy witnesses: most scalars are 0, 1, 2 or r-1, so a few buckets hold almost every digit
    and most buckets are empty (lanes cross many bucket boundaries; repeated points force doublings)."""
    n = 2000
    rng = SplitMix64(4242 + c)
    p1, p2, k = g1_points(orc, rng, n), g2_points(orc, rng, n), rand_fr(rng, n)
    small = ints_to_limbs([0, 1, 2, R_MODULUS - 1, 1 << 200, 3])
    for i in range(n):
        if i % 8:
            k[i] = small[(i * 7) % len(small)]
    for i in range(0, n - 40, 40):   # repeated points with equal scalars land in the same bucket
        p1[i + 1] = p1[i]; p2[i + 1] = p2[i]; k[i + 1] = k[i]
    assert np.array_equal(ctx.msm_g1(p1, k, c), orc.msm_g1(p1, k, 0))
    assert np.array_equal(ctx.msm_g2(p2, k, c), orc.msm_g2(p2, k, 0))


def test_msm_mid_size_vs_pippenger_oracle(ctx, orc):
    rng = SplitMix64(77)
    n = 1 << 12
    base1 = g1_points(orc, rng, 64)
    base2 = g2_points(orc, rng, 64)
    p1 = np.tile(base1, (n // 64, 1)); p2 = np.tile(base2, (n // 64, 1))
    k = rand_fr(rng, n)
    for c in (0, 8, 16):
        assert np.array_equal(ctx.msm_g1(p1, k, c), orc.msm_g1(p1, k, 10))
    assert np.array_equal(ctx.msm_g2(p2, k, 0), orc.msm_g2(p2, k, 10))


def test_msm_one_heavy_bucket(ctx, orc):
    """Nearly all scalars equal 1 (or r-1): one bucket holds ~2^15 digits, spread over >1000 lanes of the
    accumulation, and takes the workgroup-per-bucket merge path."""
    rng = SplitMix64(78)
    n = 1 << 15
    base1 = g1_points(orc, rng, 512)
    base2 = g2_points(orc, rng, 512)
    p1 = np.tile(base1, (n // 512, 1)); p2 = np.tile(base2, (n // 512, 1))
    k = np.tile(ints_to_limbs([1]), (n, 1))
    k[::3] = ints_to_limbs([R_MODULUS - 1])[0]
    k[::97] = rand_fr(rng, len(k[::97]))
    for c in (0, 9):
        assert np.array_equal(ctx.msm_g1(p1, k, c), orc.msm_g1(p1, k, 10))
    assert np.array_equal(ctx.msm_g2(p2, k, 0), orc.msm_g2(p2, k, 10))


@pytest.mark.parametrize("c", [0, 5, 12, 16, 18])
def test_msm_tail_forms_agree(ctx, orc, c):
    """The reduction tail (merge, row / column folds, weighted sum) exists in two forms -- one lane per point addition, and four lanes
    sharing each addition (csrc/quad29.cuh, inner products of at most `msm_quad_buckets` buckets).  Both against the folded
    double-and-add on the same input: repeated points (doublings inside the tail), P + (-P), infinity, buckets of many runs."""
    n = 3000
    rng = SplitMix64(5150 + c)
    p1, p2, k = g1_points(orc, rng, n), g2_points(orc, rng, n), rand_fr(rng, n)
    k[0] = 0; k[1] = ints_to_limbs([1])[0]; k[2] = ints_to_limbs([R_MODULUS - 1])[0]
    p1[3] = 0; p2[3] = 0
    for i in range(8, n - 40, 40):     # equal points with equal scalars; equal points with opposite scalars
        p1[i + 1] = p1[i]; p2[i + 1] = p2[i]; k[i + 1] = k[i]
        p1[i + 3] = p1[i + 2]; p2[i + 3] = p2[i + 2]
        k[i + 3] = ints_to_limbs([(R_MODULUS - int(zk.limbs_to_int(k[i + 2]))) % R_MODULUS])[0]
    k[1000:1400] = ints_to_limbs([5])[0]                      # one bucket of 400 entries per window 0
    want1, want2 = orc.msm_g1(p1, k, 0), orc.msm_g2(p2, k, 0)
    try:
        for quad_buckets in (0, 1 << 22):
            ctx.set_option("msm_quad_buckets", quad_buckets)
            assert np.array_equal(ctx.msm_g1(p1, k, c), want1), (c, quad_buckets)
            assert np.array_equal(ctx.msm_g2(p2, k, c), want2), (c, quad_buckets)
    finally:
        ctx.set_option("msm_quad_buckets", 65536)
    assert ctx.get_option("msm_quad_buckets") == 65536


@pytest.mark.parametrize("pattern", ["equal", "rows", "columns", "holes"])
@pytest.mark.parametrize("c", [9, 12])
def test_msm_fold_equal_and_opposite_images(ctx, orc, c, pattern):
    """The row / column sums of the tail add bucket IMAGES with the general XYZZ addition; its same-x branch (equal images: doubling in
    place, dbl_xyzz; opposite images: infinity) never fires on random data.  Here every bucket of one window holds the one point P
    or -P, so every first addition of a fold is P + P or P + (-P), and the later ones meet infinity on either side -- in the
    one-lane form (add_xyzz_from over Fq2, add_xyzz over Fq) and in the four-lane form, against the folded double-and-add."""
    buckets = 1 << (c - 1)
    kbits = c // 2                                    # columns = 2^kbits (msm_impl.hpp: kbits = c / 2)
    s = 0x1234567
    base1, base2 = orc.enc_base_g1(), orc.enc_base_g2()
    pos = ints_to_limbs([s]); neg = ints_to_limbs([R_MODULUS - s])
    P1, N1 = orc.g1_mul_batch(base1[None, :], pos)[0], orc.g1_mul_batch(base1[None, :], neg)[0]
    P2, N2 = orc.g2_mul_batch(base2[None, :], pos)[0], orc.g2_mul_batch(base2[None, :], neg)[0]
    idx = np.arange(buckets)
    if pattern == "equal":
        minus, keep = np.zeros(buckets, bool), np.ones(buckets, bool)
    elif pattern == "rows":
        minus, keep = ((idx >> kbits) & 1).astype(bool), np.ones(buckets, bool)
    elif pattern == "columns":
        minus, keep = (idx & 1).astype(bool), np.ones(buckets, bool)
    else:                                             # empty buckets (infinity images) between equal and opposite ones
        minus, keep = ((idx >> kbits) % 3 == 1), (idx % 5 != 2)
    digits = (idx + 1)[keep]                          # scalar = digit of window 0 (<= 2^(c-1): no carry into window 1)
    k = ints_to_limbs([int(d) for d in digits])
    p1 = np.where(minus[keep][:, None], N1[None, :], P1[None, :])
    p2 = np.where(minus[keep][:, None], N2[None, :], P2[None, :])
    p1, p2 = np.ascontiguousarray(p1), np.ascontiguousarray(p2)
    want1, want2 = orc.msm_g1(p1, k, 0), orc.msm_g2(p2, k, 0)
    try:
        for quad_buckets in (0, 1 << 22):
            ctx.set_option("msm_quad_buckets", quad_buckets)
            assert np.array_equal(ctx.msm_g1(p1, k, c), want1), (c, pattern, quad_buckets)
            assert np.array_equal(ctx.msm_g2(p2, k, c), want2), (c, pattern, quad_buckets)
    finally:
        ctx.set_option("msm_quad_buckets", 65536)


@pytest.mark.parametrize("c", [10, 13])
def test_msm_accumulation_meets_its_own_sum(ctx, orc, c):
    """The G1 accumulation's fast loop (csrc/madd_asm.inc: the whole mixed addition as one asm body, in place) overwrites the accumulator
    before it knows that the point has the accumulator's x; such a lane rebuilds 2 P or infinity from the affine point alone, leaves the
    loop in the canonical state and finishes its run in the generic loop.  Random data never goes there.  Here every bucket is ONE run of
    four entries whose sum pattern forces it, whatever order the sort leaves them in:
      {A, B, D, -(A+B+D)}   any three sum to minus the fourth: P + (-P) at position 4 (the odd body), always
      {A, B, D,   A+B+D }   the same point at position 4 when the sum comes last (a quarter of the buckets): doubling
      {A, B, -(A+B), D}     P + (-P) at position 3 (the even body) when D comes last; then D starts the accumulator again
      {A, B,   A+B,  D}     doubling at position 3 in a twelfth of the buckets
      {A, inf, B, D}        a table entry at infinity inside the run (the lane leaves the fast loop without consuming it)
    Scalars are single digits of window 0, so bucket d holds exactly the points given scalar d (and G2 takes the same route through its
    own loop).  Against the oracle's folded double-and-add."""
    rng = SplitMix64(6100 + c)
    nb = 320                                               # buckets used: digits 1 .. nb (<= 2^(c-1))
    A, B, D = g1_points(orc, rng, nb), g1_points(orc, rng, nb), g1_points(orc, rng, nb)
    A2, B2, D2 = g2_points(orc, rng, nb), g2_points(orc, rng, nb), g2_points(orc, rng, nb)
    minus1 = np.tile(ints_to_limbs([R_MODULUS - 1]), (nb, 1))
    AB, AB2 = orc.g1_add_batch(A, B), orc.g2_add_batch(A2, B2)
    ABD, ABD2 = orc.g1_add_batch(AB, D), orc.g2_add_batch(AB2, D2)
    neg = lambda p: orc.g1_mul_batch(p, minus1)             # noqa: E731
    neg2 = lambda p: orc.g2_mul_batch(p, minus1)            # noqa: E731
    pts, pts2, sc = [], [], []
    for d in range(nb):
        kind = d % 5
        quad = {0: (A[d], B[d], D[d], neg(ABD[d:d + 1])[0]), 1: (A[d], B[d], D[d], ABD[d]), 2: (A[d], B[d], neg(AB[d:d + 1])[0], D[d]),
                3: (A[d], B[d], AB[d], D[d]), 4: (A[d], np.zeros(8, np.uint64), B[d], D[d])}[kind]
        quad2 = {0: (A2[d], B2[d], D2[d], neg2(ABD2[d:d + 1])[0]), 1: (A2[d], B2[d], D2[d], ABD2[d]), 2: (A2[d], B2[d], neg2(AB2[d:d + 1])[0], D2[d]),
                 3: (A2[d], B2[d], AB2[d], D2[d]), 4: (A2[d], np.zeros(16, np.uint64), B2[d], D2[d])}[kind]
        for q, q2 in zip(quad, quad2):
            pts.append(q); pts2.append(q2); sc.append(d + 1)
    order = np.random.default_rng(c).permutation(len(sc))    # the entries of a bucket come from all over the scalar array
    p1 = np.ascontiguousarray(np.array(pts, dtype=np.uint64)[order])
    p2 = np.ascontiguousarray(np.array(pts2, dtype=np.uint64)[order])
    k = ints_to_limbs([sc[i] for i in order])
    assert np.array_equal(ctx.msm_g1(p1, k, c), orc.msm_g1(p1, k, 0))
    assert np.array_equal(ctx.msm_g2(p2, k, c), orc.msm_g2(p2, k, 0))
    # and with the signs of all scalars flipped (negative digits: the gathered y is negated before it meets the accumulator)
    kn = ints_to_limbs([R_MODULUS - sc[i] for i in order])
    assert np.array_equal(ctx.msm_g1(p1, kn, c), orc.msm_g1(p1, kn, 0))


def test_msm_linearity_large(ctx, orc):
    """Size-independent property at 2^18 points: MSM(P, a) + MSM(P, b) == MSM(P, a+b)."""
    rng = SplitMix64(78)
    n = 1 << 18
    base1 = g1_points(orc, rng, 256)
    p1 = np.tile(base1, (n // 256, 1))
    gen = np.random.default_rng(3)
    a = gen.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    b = gen.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); b[:, 3] &= np.uint64((1 << 60) - 1)
    sa, sb = ctx.msm_g1(p1, a), ctx.msm_g1(p1, b)
    sab = ctx.msm_g1(p1, ctx.fr_batch("add", a, b))
    assert np.array_equal(ctx.g1_add_batch(sa.reshape(1, 8), sb.reshape(1, 8))[0], sab)


# ---- the first stage of prove on its own: u_sum = sum_i qap.u[i] * weights[i] ------------------------------------------
def _weighted_sum_dense(mat, weights):
    """Restatement of the reference's fold (groth16/mod.rs:233-253): zip(qap.u, weights) -- the shorter one ends it --, every
    polynomial scaled by its weight (Mul<T>, coefficient_poly.rs:132-146) and the results summed coefficient-wise (Sum, :75-91)."""
    m, n = mat.shape[0], mat.shape[1]
    acc = [0] * n
    for i in range(min(m, weights.shape[0])):
        a = limbs_to_int(weights[i])
        for j in range(n):
            acc[j] = (acc[j] + a * limbs_to_int(mat[i, j])) % zk.R_MODULUS
    return ints_to_limbs(acc).reshape(n, 4)


@pytest.mark.parametrize("prog", ["simple.zk", "deg_15.zk"])
def test_dense_matvec_block(ctx, orc, prog):
    """k_dense_matvec alone (zk_qap_weighted_sum) on the reference's own programs: u_sum, v_sum, w_sum == the fold restated above,
    with a witness of exactly m, fewer and more than m elements (zip truncation)."""
    code = open(os.path.join(ROOT, "tests", "golden", "zk", prog)).read()
    q = orc.zk_qap_dense(code)
    qap = ctx.qap_dense(q["u"], q["v"], q["w"], q["t"], q["input"])
    rng = zk.SplitMix64(233)
    for count in (q["m"], q["m"] - 2, q["m"] + 3, 1):
        wts = ints_to_limbs([rng.fr() for _ in range(count)]).reshape(count, 4)
        for which, key in enumerate(("u", "v", "w")):
            assert np.array_equal(ctx.qap_weighted_sum(qap, wts, which), _weighted_sum_dense(q[key], wts)), (prog, count, key)
    zero = np.zeros((q["m"], 4), dtype=np.uint64)
    assert not ctx.qap_weighted_sum(qap, zero, 0).any()


@pytest.mark.parametrize("roots", ["unity", "integers"])
def test_spmv_block(ctx, roots):
    """k_spmv alone: the values of u_sum and v_sum on the QAP's domain == sum over the rows' (wire, gate, value) entries, restated
    here, for random sparse rows (wires that feed many gates, empty wires, repeated (wire, gate) entries) over both domains."""
    rng = zk.SplitMix64(253)
    n, m, l = (64, 150, 3) if roots == "unity" else (50, 150, 3)

    def rows():
        ptr, gate, val = [0], [], []
        for wire in range(m):
            cnt = 0 if wire % 7 == 3 else (n if wire == 5 else rng.next() % 5)
            for _ in range(cnt):
                gate.append(rng.next() % n)
                val.append(rng.fr())
            ptr.append(len(gate))
        return (np.array(ptr, dtype=np.uint64), np.array(gate, dtype=np.uint32), ints_to_limbs(val).reshape(-1, 4) if val else np.zeros((0, 4), np.uint64))

    u, v, w = rows(), rows(), rows()
    qap = ctx.qap_sparse(6, m, l, u, v, w) if roots == "unity" else ctx.qap_sparse_integers(n, m, l, u, v, w)
    for count in (m, m - 5, m + 2):
        wts = ints_to_limbs([rng.fr() for _ in range(count)]).reshape(count, 4)
        for which, (ptr, gate, val) in enumerate((u, v)):
            acc = [0] * n
            for wire in range(min(m, count)):
                a = limbs_to_int(wts[wire])
                for e in range(int(ptr[wire]), int(ptr[wire + 1])):
                    acc[gate[e]] = (acc[gate[e]] + a * limbs_to_int(val[e])) % zk.R_MODULUS
            assert np.array_equal(ctx.qap_weighted_sum(qap, wts, which), ints_to_limbs(acc).reshape(n, 4)), (roots, count, which)
    with pytest.raises(zk.ZkError):
        ctx.qap_weighted_sum(qap, wts, 2)        # W is not held by gate in the sparse forms
