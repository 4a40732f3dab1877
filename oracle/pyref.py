"""Python twin of the CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Big-int restatement of the reference's Groth16 prove/setup hot path over BN254,
used (a) to cross-check the C++ oracle in ``oracle/`` with an independent code
path (affine arithmetic on Python ints vs. Montgomery/Jacobian in C++), and
(b) by ``tests/golden/make_golden.py`` to generate the committed fixtures.

Reference behaviour followed (file:line into /root/reference):
  * groth16::setup            src/groth16/mod.rs:134-197
  * groth16::prove            src/groth16/mod.rs:213-296
  * CoefficientPoly Mul/Div   src/groth16/coefficient_poly.rs:93-157
  * polynomial_division       src/field/mod.rs:428-469
  * Lagrange / root_poly      src/groth16/coefficient_poly.rs:159-200
  * dft / idft                src/field/mod.rs:508-537
  * encrypt bases 69*G1, 96*G2  src/groth16/fr.rs:106-113

PARITY STATUS: the curve/field arithmetic lives in the third-party crate
``bn = "0.4.3"`` (Cargo.toml:14) which is not vendored and cannot be built here
(no Rust toolchain).  Group elements are therefore pinned by mathematics
(canonical affine coordinates over the published alt_bn128 constants), not by
reference known-answer vectors: byte-level parity with bn is UNPINNED.
"""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
TWO_ADICITY = 28
OMEGA_2_28 = pow(5, (R - 1) >> TWO_ADICITY, R)

G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
B1 = 3
# b2 = 3/(9+i)
def _fq2_inv(a):
    a0, a1 = a
    d = pow((a0 * a0 + a1 * a1) % Q, -1, Q)
    return (a0 * d % Q, (-a1 * d) % Q)
def _fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
B2 = _fq2_mul((3, 0), _fq2_inv((9, 1)))


# ----------------------------------------------------------------------------
# Deterministic randomness shared with the C++ oracle, the product and bench.py
# ----------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        """Uniform non-zero Fr: 4 limbs little-endian, top limb masked to 254 bits,
        rejection-sample < r, reject 0 (mirrors Random for FrLocal, fr.rs:90-99)."""
        while True:
            l = [self.next() for _ in range(4)]
            l[3] &= (1 << 62) - 1
            v = l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)
            if 0 < v < R:
                return v


# ----------------------------------------------------------------------------
# Generic field abstraction (prime p) so the same algorithms run over Z251 / Fr
# ----------------------------------------------------------------------------
class PrimeField:
    def __init__(self, p):
        self.p = p

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def neg(self, a): return (-a) % self.p
    def mul(self, a, b): return (a * b) % self.p
    def inv(self, a):
        if a % self.p == 0:
            raise ZeroDivisionError("Tried to divide by zero")
        return pow(a, -1, self.p)
    def div(self, a, b): return a * self.inv(b) % self.p

Z251 = PrimeField(251)
FR = PrimeField(R)


# ---- Polynomial helpers (coefficient vectors, low degree first) --------------
def degree(c):
    """field/mod.rs:291-297 -- degree ignoring high zeros; 0 for empty/zero."""
    n = len(c)
    while n > 0 and c[n - 1] == 0:
        n -= 1
    return 0 if n == 0 else n - 1

def strip(c):
    """remove_leading_zeros, field/mod.rs:344-355 (may return [])."""
    n = len(c)
    while n > 0 and c[n - 1] == 0:
        n -= 1
    return list(c[:n])

def poly_add(F, a, b):
    """coefficient_poly.rs:24-49 -- result length = max, no stripping."""
    n = max(len(a), len(b))
    return [F.add(a[i] if i < len(a) else 0, b[i] if i < len(b) else 0) for i in range(n)]

def poly_neg(F, a): return [F.neg(x) for x in a]
def poly_sub(F, a, b): return poly_add(F, a, poly_neg(F, b))
def poly_scale(F, a, s): return [F.mul(x, s) for x in a]

def poly_sum(F, polys):
    """coefficient_poly.rs:75-91 -- fold seeded with [0]."""
    acc = [0]
    for p in polys:
        acc = poly_add(F, acc, p)
    return acc

def poly_mul(F, a, b):
    """coefficient_poly.rs:93-130 -- schoolbook after stripping; length deg a+deg b+1."""
    a, b = strip(a), strip(b)
    da, db = degree(a), degree(b)
    out = [0] * (da + db + 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = F.add(out[i + j], F.mul(x, y))
    return out

def poly_divmod(F, num, den):
    """field/mod.rs:428-469 (long division, one field inversion per quotient term)."""
    if len(strip(den)) == 0:
        raise ZeroDivisionError("Dividend must be non-zero")
    if degree(den) > degree(num):
        return [0], [0]
    num, den = strip(num), strip(den)
    d = degree(den)
    c = den[d]
    q = [0] * (degree(num) + 1 - d)
    r = list(num)
    while len(r) != 0 and degree(r) >= d:
        dr = degree(r)
        s = F.div(r[dr], c)
        q[dr - d] = s
        for k in range(d + 1):
            r[dr - d + k] = F.sub(r[dr - d + k], F.mul(den[k], s))
        r = strip(r)
    return q, r

def poly_eval(F, c, x):
    acc = 0
    for y in reversed(c):
        acc = F.add(F.mul(acc, x), y)
    return acc

def powers(F, x, n):
    out, s = [], 1 % F.p
    for _ in range(n):
        out.append(s)
        s = F.mul(s, x)
    return out

def dft(F, seq, root):
    """field/mod.rs:508-520 -- out[k] = sum_j seq[j] * root^(j k)."""
    n = len(seq)
    return [sum(seq[j] * pow(root, j * k, F.p) for j in range(n)) % F.p for k in range(n)]

def idft(F, seq, root):
    n = len(seq)
    ninv = F.inv(n % F.p)
    return [F.mul(v, ninv) for v in dft(F, seq, F.inv(root))]

def lagrange_basis(F, roots, x):
    """coefficient_poly.rs:173-190."""
    acc = [1]
    for m in roots:
        if m == x:
            continue
        s = F.div(1, F.sub(x, m))
        acc = poly_mul(F, poly_scale(F, [F.neg(m), 1], s), acc)
    return acc

def poly_from_points(F, roots, points):
    """coefficient_poly.rs:159-171."""
    return poly_sum(F, [poly_scale(F, lagrange_basis(F, roots, x), y) for (x, y) in points])

def root_poly(F, roots):
    acc = [1]
    for r_ in roots:
        acc = poly_mul(F, acc, [F.neg(r_), 1])
    return acc


# ----------------------------------------------------------------------------
# BN254 groups, affine coordinates, None = point at infinity
# ----------------------------------------------------------------------------
def fq2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def fq2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def fq2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
fq2_mul = _fq2_mul
fq2_inv = _fq2_inv

class _Fq1Ops:
    zero = 0
    @staticmethod
    def add(a, b): return (a + b) % Q
    @staticmethod
    def sub(a, b): return (a - b) % Q
    @staticmethod
    def mul(a, b): return (a * b) % Q
    @staticmethod
    def inv(a): return pow(a, -1, Q)
    @staticmethod
    def neg(a): return (-a) % Q
    @staticmethod
    def small(k): return k % Q

class _Fq2Ops:
    zero = (0, 0)
    add = staticmethod(fq2_add)
    sub = staticmethod(fq2_sub)
    mul = staticmethod(fq2_mul)
    inv = staticmethod(fq2_inv)
    neg = staticmethod(fq2_neg)
    @staticmethod
    def small(k): return (k % Q, 0)

def _ec_add(K, P, Qp):
    if P is None: return Qp
    if Qp is None: return P
    x1, y1 = P
    x2, y2 = Qp
    if x1 == x2:
        if y1 == y2:
            if y1 == K.zero:
                return None
            lam = K.mul(K.mul(K.small(3), K.mul(x1, x1)), K.inv(K.add(y1, y1)))
        else:
            return None
    else:
        lam = K.mul(K.sub(y2, y1), K.inv(K.sub(x2, x1)))
    x3 = K.sub(K.sub(K.mul(lam, lam), x1), x2)
    y3 = K.sub(K.mul(lam, K.sub(x1, x3)), y1)
    return (x3, y3)

def _ec_neg(K, P):
    return None if P is None else (P[0], K.neg(P[1]))

def _ec_mul(K, P, k):
    """MSB-first double-and-add (what bn's Mul<Fr> does [recollection]); the result as a
    group element does not depend on the algorithm."""
    k %= R
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = _ec_add(K, acc, acc)
        if bit == "1":
            acc = _ec_add(K, acc, P)
    return acc

def g1_add(P, Qp): return _ec_add(_Fq1Ops, P, Qp)
def g1_neg(P): return _ec_neg(_Fq1Ops, P)
def g1_mul(P, k): return _ec_mul(_Fq1Ops, P, k)
def g2_add(P, Qp): return _ec_add(_Fq2Ops, P, Qp)
def g2_neg(P): return _ec_neg(_Fq2Ops, P)
def g2_mul(P, k): return _ec_mul(_Fq2Ops, P, k)

def g1_on_curve(P):
    return P is None or (P[1] * P[1] - P[0] ** 3 - B1) % Q == 0
def g2_on_curve(P):
    if P is None: return True
    x, y = P
    return fq2_sub(fq2_mul(y, y), fq2_add(fq2_mul(fq2_mul(x, x), x), B2)) == (0, 0)

# fr.rs:106-113: encryption bases
ENC_G1 = g1_mul(G1_GEN, 69)
ENC_G2 = g2_mul(G2_GEN, 96)
def encrypt_g1(a): return g1_mul(ENC_G1, a)
def encrypt_g2(a): return g2_mul(ENC_G2, a)

def msm_g1(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = g1_add(acc, g1_mul(p, s))
    return acc
def msm_g2(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = g2_add(acc, g2_mul(p, s))
    return acc


# ----------------------------------------------------------------------------
# Canonical proof encoding (build-defined; documented in DESIGN.md)
#   G1: 0x04 | x | y          (32-byte big-endian each), infinity = 0x00 + 64 zero bytes
#   G2: 0x04 | x.c1 | x.c0 | y.c1 | y.c0,               infinity = 0x00 + 128 zero bytes
# ----------------------------------------------------------------------------
def enc_g1(P):
    if P is None: return b"\x00" + bytes(64)
    return b"\x04" + P[0].to_bytes(32, "big") + P[1].to_bytes(32, "big")
def enc_g2(P):
    if P is None: return b"\x00" + bytes(128)
    (x0, x1), (y0, y1) = P
    return b"\x04" + b"".join(v.to_bytes(32, "big") for v in (x1, x0, y1, y0))
def enc_proof(A, B, C): return enc_g1(A) + enc_g2(B) + enc_g1(C)


# ----------------------------------------------------------------------------
# QAP (dense coefficient polynomials, as QAP<CoefficientPoly<FrLocal>>, fr.rs:140-173)
# root_rep = dict(u=[[ (root,val), ...] per wire], v=..., w=..., roots=[...], input=l)
# ----------------------------------------------------------------------------
def qap_from_root_rep(F, rr):
    roots = rr["roots"]
    qap = {k: [poly_from_points(F, roots, pts) for pts in rr[k]] for k in ("u", "v", "w")}
    assert len(qap["u"]) == len(qap["v"]) == len(qap["w"])
    qap["t"] = root_poly(F, roots)
    qap["input"] = rr["input"]
    qap["degree"] = degree(qap["t"])
    return qap

def setup_with_trapdoor(qap, trapdoor):
    """groth16/mod.rs:134-197 with the five thread_rng draws injected (alpha,beta,gamma,delta,x)."""
    F = FR
    alpha, beta, gamma, delta, x = trapdoor
    n, l = qap["degree"], qap["input"]
    xi = powers(F, x, n)
    comb = [F.add(F.add(F.mul(beta, poly_eval(F, u, x)), F.mul(alpha, poly_eval(F, v, x))), poly_eval(F, w, x))
            for u, v, w in zip(qap["u"], qap["v"], qap["w"])]
    tx = poly_eval(F, qap["t"], x)
    s1 = dict(
        alpha=encrypt_g1(alpha), beta=encrypt_g1(beta), delta=encrypt_g1(delta),
        xi=[encrypt_g1(e) for e in xi],
        sum_gamma=[encrypt_g1(F.div(c, gamma)) for c in comb[: l + 1]],
        sum_delta=[encrypt_g1(F.div(c, delta)) for c in comb[l + 1:]],
        xi_t=[encrypt_g1(F.div(F.mul(e, tx), delta)) for e in xi[: len(xi) - 1]],
    )
    s2 = dict(beta=encrypt_g2(beta), gamma=encrypt_g2(gamma), delta=encrypt_g2(delta),
              xi=[encrypt_g2(e) for e in xi])
    return s1, s2

def prove_with_rs(qap, s1, s2, weights, r_, s_):
    """groth16/mod.rs:213-296 with (r, s) injected. zips truncate as the reference's do."""
    F = FR
    def wsum(polys):
        return poly_sum(F, [poly_scale(F, p, a) for p, a in zip(polys, weights)])
    u_sum, v_sum, w_sum = wsum(qap["u"]), wsum(qap["v"]), wsum(qap["w"])
    a_g1 = msm_g1(s1["xi"], u_sum)
    b_g1 = msm_g1(s1["xi"], v_sum)
    b_g2 = msm_g2(s2["xi"], v_sum)
    a = g1_add(g1_add(a_g1, s1["alpha"]), g1_mul(s1["delta"], r_))
    b = g2_add(g2_add(b_g2, s2["beta"]), g2_mul(s2["delta"], s_))
    h, _rem = poly_divmod(F, poly_sub(F, poly_mul(F, u_sum, v_sum), w_sum), qap["t"])
    l = qap["input"]
    c = msm_g1(s1["xi_t"], h)
    c = g1_add(c, msm_g1(s1["sum_delta"], weights[l + 1:]))
    c = g1_add(c, g1_mul(a, s_))
    c = g1_add(c, g1_mul(g1_add(g1_add(s1["beta"], b_g1), g1_mul(s1["delta"], s_)), r_))
    c = g1_add(c, g1_neg(g1_mul(s1["delta"], F.mul(r_, s_))))
    return a, b, c

def trapdoor_proof(qap, trapdoor, weights, r_, s_):
    """Pairing-free validity oracle: closed form of an honest proof given the
    trapdoor; independent of MSM/NTT/summation order."""
    F = FR
    alpha, beta, gamma, delta, x = trapdoor
    l = qap["input"]
    ux = [poly_eval(F, p, x) for p in qap["u"]]
    vx = [poly_eval(F, p, x) for p in qap["v"]]
    wx = [poly_eval(F, p, x) for p in qap["w"]]
    U = sum(a * e for a, e in zip(weights, ux)) % R
    V = sum(a * e for a, e in zip(weights, vx)) % R
    W = sum(a * e for a, e in zip(weights, wx)) % R
    tx = poly_eval(F, qap["t"], x)
    # h(x) t(x) with h the QUOTIENT (remainder dropped, as the reference does)
    def wsum(polys):
        return poly_sum(F, [poly_scale(F, p, a) for p, a in zip(polys, weights)])
    h, _ = poly_divmod(F, poly_sub(F, poly_mul(F, wsum(qap["u"]), wsum(qap["v"])), wsum(qap["w"])), qap["t"])
    hx = poly_eval(F, h[: max(qap["degree"] - 1, 0)], x)
    a_log = (alpha + U + r_ * delta) % R
    b_log = (beta + V + s_ * delta) % R
    L = sum(weights[i] * (beta * ux[i] + alpha * vx[i] + wx[i]) for i in range(l + 1, min(len(weights), len(ux)))) % R
    c_log = (F.div((L + hx * tx) % R, delta) + s_ * a_log + r_ * b_log - r_ * s_ * delta) % R
    return encrypt_g1(a_log), encrypt_g2(b_log), encrypt_g1(c_log)


# ----------------------------------------------------------------------------
# Synthetic "chain" circuit = deg_15.zk generalised, roots = omega^j
# ----------------------------------------------------------------------------
def omega(log_n):
    return pow(OMEGA_2_28, 1 << (TWO_ADICITY - log_n), R)

def chain_root_rep(n, roots):
    """gate k (1-based): t_k = x*(t_{k-1}+a_k) [k<n], y = 1*(t_{n-1}+a_n) [k=n].
    wires: 0:1 1:x 2:y 3:t1 4:a1 5:t2 6:a2 ... (2k+1: t_k, 2k+2: a_k), last: a_n at 2n+1; m=2n+2."""
    m = 2 * n + 2
    u = [[] for _ in range(m)]
    v = [[] for _ in range(m)]
    w = [[] for _ in range(m)]
    t = lambda k: 2 * k + 1            # wire of t_k, 1<=k<=n-1
    a = lambda k: 2 * k + 2 if k < n else 2 * n + 1
    for k in range(1, n + 1):
        rt = roots[k - 1]
        if k < n:
            w[t(k)].append((rt, 1)); u[1].append((rt, 1))
        else:
            w[2].append((rt, 1)); u[0].append((rt, 1))
        if k >= 2:
            v[t(k - 1)].append((rt, 1))
        v[a(k)].append((rt, 1))
    return dict(u=u, v=v, w=w, roots=list(roots), input=2)

def chain_weights(n, x, avals):
    F = FR
    m = 2 * n + 2
    wts = [0] * m
    wts[0], wts[1] = 1, x
    prev = 0
    for k in range(1, n + 1):
        ak = avals[k - 1]
        if k < n:
            wts[2 * k + 2] = ak
            prev = F.mul(x, F.add(prev, ak))
            wts[2 * k + 1] = prev
        else:
            wts[2 * n + 1] = ak
            wts[2] = F.add(prev, ak)
    return wts


# ----------------------------------------------------------------------------
# Optimal ate pairing on BN254 (what bn::pairing computes, fr.rs:120-122) -- big-int twin used to
# pin the product's host pairing (zk_pairing / zk_verify).  Tower: Fq2 = Fq[i]/(i^2+1),
# Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + i, Fq12 = Fq6[w]/(w^2 - v).  D-type twist
# psi(x', y') = (x' w^2, y' w^3).  Affine line functions; final exponentiation by plain
# square-and-multiply with (q^12 - 1)/r.
# ----------------------------------------------------------------------------
BN_U = 4965661367192848881
ATE_LOOP = 6 * BN_U + 2
XI = (9, 1)
FINAL_EXP = (Q ** 12 - 1) // R

def fq2_sqr(a): return fq2_mul(a, a)
def fq2_mul_xi(a): return ((9 * a[0] - a[1]) % Q, (9 * a[1] + a[0]) % Q)
def fq2_conj(a): return (a[0], (-a[1]) % Q)
def fq2_pow(a, e):
    r_ = (1, 0)
    while e:
        if e & 1: r_ = fq2_mul(r_, a)
        a = fq2_mul(a, a); e >>= 1
    return r_
FQ2_ZERO, FQ2_ONE = (0, 0), (1, 0)

def fq6_add(a, b): return tuple(fq2_add(x, y) for x, y in zip(a, b))
def fq6_sub(a, b): return tuple(fq2_sub(x, y) for x, y in zip(a, b))
def fq6_neg(a): return tuple(fq2_neg(x) for x in a)
def fq6_mul(a, b):
    a0, a1, a2 = a; b0, b1, b2 = b
    c0 = fq2_add(fq2_mul(a0, b0), fq2_mul_xi(fq2_add(fq2_mul(a1, b2), fq2_mul(a2, b1))))
    c1 = fq2_add(fq2_add(fq2_mul(a0, b1), fq2_mul(a1, b0)), fq2_mul_xi(fq2_mul(a2, b2)))
    c2 = fq2_add(fq2_add(fq2_mul(a0, b2), fq2_mul(a1, b1)), fq2_mul(a2, b0))
    return (c0, c1, c2)
def fq6_mul_v(a): return (fq2_mul_xi(a[2]), a[0], a[1])
def fq6_inv(a):
    a0, a1, a2 = a
    t0 = fq2_sub(fq2_sqr(a0), fq2_mul_xi(fq2_mul(a1, a2)))
    t1 = fq2_sub(fq2_mul_xi(fq2_sqr(a2)), fq2_mul(a0, a1))
    t2 = fq2_sub(fq2_sqr(a1), fq2_mul(a0, a2))
    d = fq2_add(fq2_mul(a0, t0), fq2_mul_xi(fq2_add(fq2_mul(a2, t1), fq2_mul(a1, t2))))
    di = fq2_inv(d)
    return (fq2_mul(t0, di), fq2_mul(t1, di), fq2_mul(t2, di))
FQ6_ZERO = (FQ2_ZERO, FQ2_ZERO, FQ2_ZERO)
FQ6_ONE = (FQ2_ONE, FQ2_ZERO, FQ2_ZERO)

def fq12_mul(a, b):
    a0, a1 = a; b0, b1 = b
    t0, t1 = fq6_mul(a0, b0), fq6_mul(a1, b1)
    return (fq6_add(t0, fq6_mul_v(t1)), fq6_add(fq6_mul(a0, b1), fq6_mul(a1, b0)))
def fq12_inv(a):
    a0, a1 = a
    d = fq6_inv(fq6_sub(fq6_mul(a0, a0), fq6_mul_v(fq6_mul(a1, a1))))
    return (fq6_mul(a0, d), fq6_neg(fq6_mul(a1, d)))
FQ12_ONE = (FQ6_ONE, FQ6_ZERO)
def fq12_pow(a, e):
    r_ = FQ12_ONE
    for bit in bin(e)[2:]:
        r_ = fq12_mul(r_, r_)
        if bit == "1": r_ = fq12_mul(r_, a)
    return r_

def _line(T, Q2, P):
    """Line through T and Q2 (tangent when equal) on the twist, evaluated at P in G1, and T + Q2."""
    (xt, yt), (xq, yq) = T, Q2
    if T == Q2:
        lam = fq2_mul(fq2_mul((3, 0), fq2_sqr(xt)), fq2_inv(fq2_add(yt, yt)))
    else:
        lam = fq2_mul(fq2_sub(yq, yt), fq2_inv(fq2_sub(xq, xt)))
    x3 = fq2_sub(fq2_sub(fq2_sqr(lam), xt), xq)
    y3 = fq2_sub(fq2_mul(lam, fq2_sub(xt, x3)), yt)
    xp, yp = P
    # l = yP - lam xP w + (lam xT - yT) w^3 ;  w^3 = v w
    c0 = ((yp % Q, 0), FQ2_ZERO, FQ2_ZERO)
    c1 = (fq2_neg(fq2_mul(lam, (xp % Q, 0))), fq2_sub(fq2_mul(lam, xt), yt), FQ2_ZERO)
    return (c0, c1), (x3, y3)

GAMMA_X = fq2_pow(XI, (Q - 1) // 3)        # Frobenius on the twist: x' -> conj(x') * xi^((q-1)/3)
GAMMA_Y = fq2_pow(XI, (Q - 1) // 2)        #                         y' -> conj(y') * xi^((q-1)/2)
def twist_frobenius(Pt):
    return (fq2_mul(fq2_conj(Pt[0]), GAMMA_X), fq2_mul(fq2_conj(Pt[1]), GAMMA_Y))

def miller_loop(P, Qt):
    if P is None or Qt is None:
        return FQ12_ONE
    f, T = FQ12_ONE, Qt
    for bit in bin(ATE_LOOP)[3:]:
        l, T = _line(T, T, P)
        f = fq12_mul(fq12_mul(f, f), l)
        if bit == "1":
            l, T = _line(T, Qt, P)
            f = fq12_mul(f, l)
    Q1 = twist_frobenius(Qt)
    Q2n = g2_neg(twist_frobenius(Q1))
    l, T = _line(T, Q1, P)
    f = fq12_mul(f, l)
    l, T = _line(T, Q2n, P)
    f = fq12_mul(f, l)
    return f

def final_exponentiation(f): return fq12_pow(f, FINAL_EXP)
def pairing(P, Qt): return final_exponentiation(miller_loop(P, Qt))

def verify_bn(s1, s2, inputs, proof):
    """groth16::verify (mod.rs:299-320): e(alpha,beta) e(sum_term,gamma) e(C,delta) == e(A,B)."""
    a, b, c = proof
    sum_term = None
    for g, x in zip(s1["sum_gamma"], [1] + list(inputs)):
        sum_term = g1_add(sum_term, g1_mul(g, x))
    lhs = fq12_mul(fq12_mul(pairing(s1["alpha"], s2["beta"]), pairing(sum_term, s2["gamma"])), pairing(c, s2["delta"]))
    return lhs == pairing(a, b)

def fq12_flat(f):
    """12 Fq coefficients in the order c0.a0.c0, c0.a0.c1, c0.a1.c0, ... c1.a2.c1 (the ABI order of zk_pairing)."""
    return [x for six in f for two in six for x in two]
