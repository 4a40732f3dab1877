"""The lazy radix-2^29 arithmetic (csrc/lazy29.cuh, the NTT tile's fr_reduce / fr_store_exact) at the EXTREMES of the bounds
its comments argue, through the diagnostic entry point zk_lazy29_batch -- random MSM / NTT data never reaches them, and a
column overflow there would be a silent wrong proof on a rare input.  Reference: Python big ints."""
import ctypes as C

import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd import _lib

P = {0: zk.R_MODULUS, 1: zk.Q_MODULUS}
MONT, SQR, MONT_DIFF, NORM, STORE, FR_REDUCE, FR_STORE = range(7)
RINV = {f: pow(1 << 261, -1, p) for f, p in P.items()}
M29 = (1 << 29) - 1


def value(limbs):
    return sum(int(v) << (29 * i) for i, v in enumerate(limbs))


def normal_form(v):
    """limbs 0..7 in [0, 2^29), signed top limb"""
    out = []
    for _ in range(8):
        out.append(v & M29)
        v >>= 29
    out.append(v)
    return out


def run(ctx, field, op, *ops, raw=False):
    arrs = [np.ascontiguousarray(np.array(o, dtype=np.int64).astype(np.int32).reshape(-1, 9)) for o in ops]
    n = arrs[0].shape[0]
    ptrs = [a.ctypes.data_as(_lib.i32p) for a in arrs] + [None] * (4 - len(arrs))
    out = np.zeros((n, 4), np.uint64)
    rawa = np.zeros((n, 9), np.int32)
    ctx._check(ctx.lib.zk_lazy29_batch(ctx.ptr, field, op, *ptrs, n, out.ctypes.data_as(_lib.u64p), rawa.ctypes.data_as(_lib.i32p) if raw else None))
    res = [zk.limbs_to_int(o) for o in out]
    return (res, rawa) if raw else res


def patterns(rng, bound, p, vmax, count=48):
    """limb vectors with |low limbs| <= bound and |value| < vmax p: the corners, alternating signs, values next to k p, random.
    (The multiplier's column bound is a LIMB bound -- 18 products per 64-bit column -- on every limb; the top limb is what
    carries the value bound: a residue that is a sum of a few reduced values has a small top limb whatever its low limbs are.)"""
    top = (vmax - 3) * p >> 232          # top limbs that keep |value| < vmax p whatever the 8 low limbs are (they add < 3 units)
    pats = []
    for t in (top, -top, 0):
        pats += [[bound] * 8 + [t], [-bound] * 8 + [t], [bound if i % 2 else -bound for i in range(8)] + [t],
                 [-bound if i % 2 else bound for i in range(8)] + [t]]
    pats += [[0] * 9, [1] + [0] * 8, [bound] + [0] * 8, [0] * 7 + [bound, 0], [0] * 7 + [-bound, 0]]
    for k in (-2, -1, 1, 2, 7):
        for d in (-1, 0, 1):
            pats.append(normal_form(k * p + d))
    while len(pats) < count:
        pats.append([int(x) for x in rng.integers(-bound, bound + 1, size=8)] + [int(rng.integers(-top, top + 1))])
    assert all(abs(value(q)) < vmax * p for q in pats)
    return pats


@pytest.mark.gpu
@pytest.mark.parametrize("field", [0, 1])
def test_mont_sqr_and_mont_diff_at_the_limb_bounds(ctx, field):
    p, rinv = P[field], RINV[field]
    rng = np.random.default_rng(29 + field)
    A30 = patterns(rng, 1 << 30, p, 32)        # |a limbs| <= 2^30 on ONE side (sums of two forms), |value| < 32 p
    B29 = patterns(rng, M29, p, 16)            # |b limbs| < 2^29 on the other, |value| < 16 p
    pairs = [(a, b) for a in A30 for b in B29[:12]]
    got, raw = run(ctx, field, MONT, [a for a, _ in pairs], [b for _, b in pairs], raw=True)
    for (a, b), g, r in zip(pairs, got, raw):
        assert g == value(a) * value(b) * rinv % p
        # output contract: normal form; value = a b / 2^261 + (0 .. p), i.e. within (-3.1 p, 4.1 p) for these operands and
        # within (-0.4 p, 1.4 p) when both are below 8 p (what lazy29.cuh calls "contracts back")
        assert all(0 <= int(x) <= M29 for x in r[:8])
        lim = abs(value(a) * value(b)) // (1 << 261) + 1
        assert -lim <= value(r) <= lim + p
    got = run(ctx, field, SQR, B29)
    assert got == [value(a) ** 2 * rinv % p for a in B29]
    quads = [(B29[i], B29[(i * 7 + 1) % len(B29)], B29[(i * 5 + 2) % len(B29)], B29[(i * 3 + 3) % len(B29)]) for i in range(len(B29))]
    hi, lo = B29[0], B29[1]            # every low limb at +(2^29 - 1) / -(2^29 - 1), top limb at the value bound
    neg = [-x for x in hi]
    quads += [(hi, hi, hi, neg), (neg, hi, hi, hi), (lo, lo, hi, neg)]   # both products with the same sign: 18 column terms add up
    got, raw = run(ctx, field, MONT_DIFF, *[[q[k] for q in quads] for k in range(4)], raw=True)
    for q, g, r in zip(quads, got, raw):
        assert g == (value(q[0]) * value(q[1]) - value(q[2]) * value(q[3])) * rinv % p
        assert all(0 <= int(x) <= M29 for x in r[:8]) and abs(value(r)) < 5 * p


@pytest.mark.gpu
@pytest.mark.parametrize("field", [0, 1])
def test_store_exact_accepts_everything_below_16p(ctx, field):
    """store_exact's range was 8 p in round 1 -- and dbl_lazy's outputs reach +-13 p (found by the row / column form of the
    MSM tail, which stores k * P for even k straight out of a doubling)"""
    p = P[field]
    rng = np.random.default_rng(31 + field)
    vals = [k * p + d for k in range(-16, 16) for d in (-1, 0, 1, p // 2)]
    vals = [v for v in vals if abs(v) < 16 * p] + [16 * p - 1, -16 * p + 1] + [int(rng.integers(-(1 << 62), 1 << 62)) * p // (1 << 59) for _ in range(64)]
    vals = [v for v in vals if abs(v) < 16 * p]
    assert run(ctx, field, STORE, [normal_form(v) for v in vals]) == [v % p for v in vals]
    # un-normalised limbs (signed sums of up to 3 normal forms: |limb| < 2^31): the same residue, with and without NORM
    loose = []
    for _ in range(64):
        terms = [normal_form(int(rng.integers(0, 1 << 62)) * p // (1 << 62)) for _ in range(int(rng.integers(1, 4)))]
        sign = [1 if rng.integers(0, 2) else -1 for _ in terms]
        loose.append([sum(s * t[i] for s, t in zip(sign, terms)) for i in range(9)])
    assert run(ctx, field, STORE, loose) == [value(l) % p for l in loose]
    got, raw = run(ctx, field, NORM, loose, raw=True)
    assert got == [value(l) % p for l in loose]
    for l, r in zip(loose, raw):
        assert value(r) == value(l) and all(0 <= int(x) <= M29 for x in r[:8])


@pytest.mark.gpu
def test_ntt_tile_reductions(ctx):
    """fr_reduce: any limbs within int32 and |value| < 2^9 r in -> normal form in (-r - eps, 2 r) out; fr_store_exact: canonical"""
    r = zk.R_MODULUS
    rng = np.random.default_rng(37)
    pats = []
    for k in (-511, -45, -2, -1, 0, 1, 2, 44, 45, 511):
        for d in (-1, 0, 1, r // 3):
            pats.append(normal_form(k * r + d))
    # un-normalised limbs: every low limb at +-(2^31 - 1) with the top limb keeping |value| < 2^9 r
    for s in (1, -1):
        low = [s * ((1 << 31) - 1)] * 8
        for top in (0, 1 << 27, -(1 << 27), (1 << 30), -(1 << 30)):
            pats.append(low + [top])
    while len(pats) < 128:
        limbs = [int(x) for x in rng.integers(-(1 << 31) + 1, 1 << 31, size=8)] + [int(rng.integers(-(1 << 30), 1 << 30))]
        pats.append(limbs)
    pats = [q for q in pats if abs(value(q)) < (1 << 9) * r]
    got, raw = run(ctx, 0, FR_REDUCE, pats, raw=True)
    eps = r >> 12
    for q, g, rw in zip(pats, got, raw):
        assert g == value(q) % r
        assert value(rw) % r == value(q) % r and -r - eps < value(rw) < 2 * r and all(0 <= int(x) <= M29 for x in rw[:8])
    assert run(ctx, 0, FR_STORE, pats) == [value(q) % r for q in pats]


@pytest.mark.gpu
def test_lazy29_argument_errors(ctx):
    a = np.zeros((1, 9), np.int32)
    out = np.zeros((1, 4), np.uint64)
    ap, op = a.ctypes.data_as(_lib.i32p), out.ctypes.data_as(_lib.u64p)
    assert ctx.lib.zk_lazy29_batch(ctx.ptr, 2, MONT, ap, ap, None, None, 1, op, None) == -1          # field
    assert ctx.lib.zk_lazy29_batch(ctx.ptr, 1, FR_REDUCE, ap, None, None, None, 1, op, None) == -1   # Fr-only op
    assert ctx.lib.zk_lazy29_batch(ctx.ptr, 0, MONT, ap, None, None, None, 1, op, None) == -1        # missing operand
    assert ctx.lib.zk_lazy29_batch(ctx.ptr, 0, 9, ap, None, None, None, 1, op, None) == -1


def test_three_inversions_agree_on_the_host(tmp_path):
    """ff.cuh is __host__ __device__: Fermat's ladder, the binary extended Euclid and the division steps in batches of 30 that close
    every proof (inv_divsteps) on small values, powers of two and 20000 random values per field, plus x * x^-1 == 1 -- compiled with
    hipcc and run on the CPU (tests/cpp/inverse_check.hip).  The GPU side of the same code: test_field_ops[inv_divsteps]."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "inverse_check")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "zksnark_rs_amd", "csrc"),
                    os.path.join(root, "tests", "cpp", "inverse_check.hip"), "-o", exe], check=True, capture_output=True, text=True, timeout=600)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Fr: 20850 cases, 0 bad" in res.stdout and "Fq: 20850 cases, 0 bad" in res.stdout, res.stdout + res.stderr
