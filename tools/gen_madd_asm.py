#!/usr/bin/env python3
"""Generator of the bucket accumulation's mixed addition as ONE inline-asm body per field (csrc/madd_asm.inc).

    python tools/gen_madd_asm.py            # check the bodies on the CPU, then write zksnark_rs_amd/csrc/madd_asm.inc
    python tools/gen_madd_asm.py --check    # only the check

Why: with one asm statement per MULTIPLICATION (mont_asm.inc) every product has early-clobber outputs, so the
accumulator changes registers on every trip and the compiler copies it back at the loop's joins -- 36 v_mov per G1 addition in
round 5's kernel, plus a negated copy of Y for the one difference of products, plus the phi copies of the rare paths.  Here the whole
madd-2008-s addition is one statement whose multipliers work IN PLACE:

  * the m_k of the Montgomery reduction live in their own nine registers (shared by all products), so output limb k -- written
    at column k + 9 -- may take the register of input limb a_k, which was last read at column k + 8: U2 = x2 ZZ lands on x2,
    P PP on P, Q = X PP on X, ZZ PP on ZZ, ZZZ PPP on ZZZ, and R D + Y PPP on Y;
  * X3 = R^2 - PPP - 2Q is formed on the registers of R^2, so X changes place once per addition: the loop is unrolled by two and
    the second body takes the two register sets the other way round (XA -> XB, then XB -> XA): no copy;
  * Y3 = R (Q - X3) - Y PPP needs one negated operand.  The bodies alternate instead: the EVEN body computes
    R (X3 - Q) + Y PPP = -Y3 and leaves the accumulator as (X3, -Y3, ZZ3, ZZZ3); the ODD body starts from that (R = S2 + N,
    N = -Y) and computes R (Q - X3) + N PPP = +Y3.  No negation is ever executed; a run that ends after an even body negates once.
  * the test "same x" (PP == 0 mod p) is a two-compare filter on limb 0 of PP; only when a lane of the wave passes it the exact
    comparison (with 0 and with p: PP is a Montgomery output in (-p/4, 1.3 p), whose normal form is unique) runs, and then the same
    for R^2.  A lane with PP == 0 computes garbage in place; the caller rebuilds its accumulator from the affine point alone
    (same x: the accumulator IS +-the point, so the sum is 2 P or infinity).

Every body is a list of abstract instructions that `simulate` executes on Python integers with the wrap-around semantics of the
hardware instructions -- register aliasing, ordering and the 64-bit column bounds are checked here, on the CPU -- against (a) the
limb-exact C++ definition (lazy29.cuh madd_xyzz_nz: FpR::mont / mont_sum / norm) and (b) the affine sum computed with plain
integers.  The 64-bit accumulator is the fixed pair v[4:5] (gfx950 inline asm cannot name the low half of a 64-bit operand).
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_mont_asm import FQ, M29, column_range, model, s32, s64, value   # noqa: E402

ACC = "v[4:5]"
ACC_LO = "v4"


class Prog:
    """instructions over symbolic 32-bit registers (strings), one 64-bit accumulator and lane masks (SGPR pairs, names 'z', ...)"""

    def __init__(self):
        self.ins = []

    def emit(self, *t):
        self.ins.append(t)

    def valu(self):
        """VALU instructions on the hot path (the rare blocks between 'rare_begin' and 'rare_end' do not run)"""
        n, rare = 0, 0
        for i in self.ins:
            if i[0] == "rare_begin":
                rare += 1
            elif i[0] == "rare_end":
                rare -= 1
            elif not rare and i[0] not in ("s_and", "s_or", "s_mov0", "comment"):
                n += 1
        return n


def mul_inplace(q, a, b, r, m, movlo=True):
    """r = a b / R, product scanning, one accumulator; r may be a (limb k of r is written after the last read of a_k)"""
    first = True
    for k in range(17):
        lo, hi = column_range(k)
        for i in range(lo, hi + 1):
            q.emit("mad", a[i], b[k - i], first)
            first = False
        reduce_col(q, k, m, r, "shr" if movlo else "acc")


def sqr_inplace(q, a, r, m, t, movlo=True):
    """r = a a / R; t = eight registers for the doubled limbs; r must not be a (a is read again)"""
    for i in range(8):
        q.emit("shl1", t[i], a[i])
    first = True
    for k in range(17):
        lo, hi = column_range(k)
        for i in range(lo, hi + 1):
            j = k - i
            if i < j:
                q.emit("mad", t[i], a[j], first)
            elif i == j:
                q.emit("mad", a[i], a[i], first)
            else:
                continue
            first = False
        reduce_col(q, k, m, r, "shr" if movlo else "acc")


def sum_inplace(q, a, b, c, d, r, m):
    """r = (a b + c d) / R with one reduction; r may be c"""
    first = True
    for k in range(17):
        lo, hi = column_range(k)
        for i in range(lo, hi + 1):
            q.emit("mad", a[i], b[k - i], first)
            first = False
        for i in range(lo, hi + 1):
            q.emit("mad", c[i], d[k - i], False)
        reduce_col(q, k, m, r)


def reduce_col(q, k, m, r, last="shr"):
    """last (column 16 only): "shr" = limb 8 of the result straight into r[8] (one v_alignbit_b32); "acc" = leave it in the
    accumulator's low half (the caller subtracts from it)"""
    lo, hi = column_range(k)
    for i in range(lo, hi + 1):
        if i < k or k >= 9:
            q.emit("mad", m[i], "p%d" % (k - i), False)
    if k < 9:
        q.emit("mullo", m[k], "inv")
        q.emit("and29", m[k], m[k])
        q.emit("mad", m[k], "p0", False)
    else:
        q.emit("and29", r[k - 9], "ACC")
    if k == 16 and last == "shr":
        q.emit("shr29lo", r[8])
    else:
        q.emit("ashr")


def exact_zero(q, v, z, s1, s2, tmp):
    """z = (v == 0 or v == p as nine limbs); v: a Montgomery output in normal form.  Hot path: two compares and one scalar or."""
    q.emit("cmp_eq", z, 0, v[0])
    q.emit("cmp_eq", s1, "p0", v[0])
    q.emit("s_or", s2, z, s1)
    q.emit("rare_begin", "scc0")              # skipped unless a lane of the wave passed the filter (then z = s1 = 0 everywhere)
    q.emit("or3", tmp[0], v[1], v[2], v[3])
    q.emit("or3", tmp[0], tmp[0], v[4], v[5])
    q.emit("or3", tmp[0], tmp[0], v[6], v[7])
    q.emit("or2", tmp[0], tmp[0], v[8])
    q.emit("cmp_eq", s2, 0, tmp[0])
    q.emit("s_and", z, z, s2)
    for i in range(1, 9):
        q.emit("xor", tmp[i], "p%d" % i, v[i])
    q.emit("or3", tmp[1], tmp[1], tmp[2], tmp[3])
    q.emit("or3", tmp[1], tmp[1], tmp[4], tmp[5])
    q.emit("or3", tmp[1], tmp[1], tmp[6], tmp[7])
    q.emit("or2", tmp[1], tmp[1], tmp[8])
    q.emit("cmp_eq", s2, 0, tmp[1])
    q.emit("s_and", s1, s1, s2)
    q.emit("s_or", z, z, s1)
    q.emit("rare_end")


def regs(pfx):
    return ["%s%d" % (pfx, i) for i in range(9)]


def gen_madd_g1(odd):
    """One mixed XYZZ addition over Fq, in place.  Register groups (nine limbs each):
         XA   in: X (normal form)              out: garbage (Q)
         XB   in: -                            out: X3 (normal form)
         Y    in: Y (even) / -Y (odd)          out: -Y3 (even) / Y3 (odd)
         ZZ, ZZZ  in / out
         QX, QY   in: the affine point in the lazy radix (QY possibly negated: |limb| < 2^29)   out: garbage
         PP, M    scratch
       masks: z = lanes whose PP == 0 mod p (same x; their outputs are garbage), z2 = z and R^2 == 0 mod p (same point)"""
    q = Prog()
    XA, XB, Y, ZZ, ZZZ, QX, QY, PP, M = (regs(n) for n in ("XA", "XB", "Y", "ZZ", "ZZZ", "QX", "QY", "PP", "M"))
    mul_inplace(q, QX, ZZ, QX, M, movlo=False)           # U2 = x2 ZZ
    for i in range(8):
        q.emit("sub", QX[i], QX[i], XA[i])              # P = U2 - X
    q.emit("sub", QX[8], "ACC", XA[8])
    mul_inplace(q, QY, ZZZ, QY, M, movlo=False)          # S2 = y2 ZZZ
    for i in range(9):
        src = QY[i] if i < 8 else "ACC"
        q.emit("add" if odd else "sub", QY[i], src, Y[i])   # R = S2 - Y  (odd: the register holds -Y)
    sqr_inplace(q, QX, PP, M, XB)                        # PP = P^2 (doubled limbs on XB, which is free until R^2)
    exact_zero(q, PP, "z", "s1", "s2", M)
    mul_inplace(q, XA, PP, XA, M)                        # Q = X PP
    mul_inplace(q, QX, PP, QX, M)                        # PPP = P PP
    mul_inplace(q, ZZ, PP, ZZ, M)                        # ZZ3
    mul_inplace(q, ZZZ, QX, ZZZ, M)                      # ZZZ3
    sqr_inplace(q, QY, XB, M, PP, movlo=False)           # R^2 (doubled limbs on PP, dead by now); limb 8 still in the accumulator
    q.emit("s_mov0", "z2")
    q.emit("rare_begin", "z==0")                         # only when a lane has the same x: is it the same point?
    q.emit("movlo", XB[8])
    exact_zero(q, XB, "z2", "s1", "s2", M)
    q.emit("s_and", "z2", "z2", "z")
    q.emit("rare_end")
    # X3 = R^2 - PPP - 2 Q, normalised on XB
    for i in range(9):
        q.emit("lshl1_add", PP[0], XA[i], QX[i])        # 2 Q + PPP
        q.emit("sub", XB[i], XB[i] if i < 8 else "ACC", PP[0])
    for i in range(8):                                  # carry propagation (FpR::norm)
        if i:
            q.emit("add", XB[i], XB[i], PP[0])
        q.emit("ashr32", PP[0], XB[i])
        q.emit("and29", XB[i], XB[i])
    q.emit("add", XB[8], XB[8], PP[0])
    for i in range(9):                                  # even: X3 - Q, odd: Q - X3
        if odd:
            q.emit("sub", PP[i], XA[i], XB[i])
        else:
            q.emit("sub", PP[i], XB[i], XA[i])
    sum_inplace(q, QY, PP, Y, QX, Y, M)                  # even: R (X3 - Q) + Y PPP = -Y3;  odd: R (Q - X3) + (-Y) PPP = Y3
    return q


# ---- simulation --------------------------------------------------------------------------------------------------------
def simulate(q, regs_in, P29, INV29):
    r = dict(regs_in)
    for i in range(9):
        r["p%d" % i] = P29[i]
    r["inv"] = INV29
    acc = None
    masks = {}

    def val(x):
        if x == "ACC":
            return s32(acc)
        if isinstance(x, int):
            return x
        return r[x]

    for ins in q.ins:
        op = ins[0]
        if op == "mad":
            _, x, y, fresh = ins
            t = s32(r[x]) * s32(r[y]) + (0 if fresh else acc)
            assert abs(t) < (1 << 63), "column overflow"
            acc = s64(t)
        elif op == "mullo":
            r[ins[1]] = ((acc & 0xffffffff) * (r[ins[2]] & 0xffffffff)) & 0xffffffff
        elif op == "and29":
            r[ins[1]] = (val(ins[2]) & 0xffffffff) & M29
        elif op == "ashr":
            acc = acc >> 29
        elif op == "ashr32":
            r[ins[1]] = s32(r[ins[2]]) >> 29
        elif op == "movlo":
            r[ins[1]] = s32(acc)
        elif op == "shr29lo":
            r[ins[1]] = s32(acc >> 29)
        elif op == "shl1":
            r[ins[1]] = s32(r[ins[2]] << 1)
        elif op == "sub":
            r[ins[1]] = s32(val(ins[2]) - val(ins[3]))
        elif op == "add":
            r[ins[1]] = s32(val(ins[2]) + val(ins[3]))
        elif op == "lshl1_add":
            r[ins[1]] = s32((val(ins[2]) << 1) + val(ins[3]))
        elif op == "cmp_eq":
            masks[ins[1]] = (val(ins[2]) & 0xffffffff) == (val(ins[3]) & 0xffffffff)
        elif op == "s_and":
            masks[ins[1]] = masks[ins[2]] and masks[ins[3]]
        elif op == "s_or":
            masks[ins[1]] = masks[ins[2]] or masks[ins[3]]
        elif op == "s_mov0":
            masks[ins[1]] = False
        elif op == "or3":
            r[ins[1]] = (val(ins[2]) | val(ins[3]) | val(ins[4])) & 0xffffffff
        elif op == "or2":
            r[ins[1]] = (val(ins[2]) | val(ins[3])) & 0xffffffff
        elif op == "xor":
            r[ins[1]] = (val(ins[2]) ^ val(ins[3])) & 0xffffffff
        elif op in ("rare_begin", "rare_end", "comment"):
            pass     # the rare blocks are executed unconditionally here: skipping them leaves the masks they refine at zero (see exact_zero)
        else:
            raise ValueError(op)
    return r, masks


# ---- the C++ definition, limb-exact (lazy29.cuh madd_xyzz_nz) -----------------------------------------------------------
def norm(v):
    out, c = [], 0
    for i in range(8):
        t = v[i] + c
        out.append(t & M29)
        c = t >> 29
    out.append(v[8] + c)
    return out


def ref_madd(X, Y, ZZ, ZZZ, qx, qy, params):
    mont = lambda a, b: model([(a, b)], **params)          # noqa: E731
    sub = lambda a, b: [x - y for x, y in zip(a, b)]       # noqa: E731
    U2, S2 = mont(qx, ZZ), mont(qy, ZZZ)
    P, R = sub(U2, X), sub(S2, Y)
    PP = mont(P, P)
    PPP, Q = mont(P, PP), mont(X, PP)
    RR = mont(R, R)
    X3 = norm([RR[i] - PPP[i] - 2 * Q[i] for i in range(9)])
    Y3 = model([(R, sub(Q, X3)), ([-y for y in Y], PPP)], **params)
    return X3, Y3, mont(ZZ, PP), mont(ZZZ, PPP), PP, RR


def normal_form(v):
    """limbs 0..7 in [0, 2^29), limb 8 signed"""
    l = []
    for i in range(8):
        l.append(v & M29)
        v >>= 29
    l.append(v)
    return l


def check(trials=200):
    rng = random.Random(20261001)
    params = FQ
    p = value(params["P29"])
    R = 1 << 261
    Rinv = pow(R, -1, p)
    bodies = {False: gen_madd_g1(False), True: gen_madd_g1(True)}

    def lazy(v):
        """a representative of v mod p as a Montgomery output would be: value in (-p/4, 1.3 p), normal form"""
        k = rng.choice((-1, 0, 0, 0, 1))
        w = v % p + k * p
        if not (-p // 4 < w < 13 * p // 10):
            w = v % p
        return normal_form(w)

    def rand_point():
        while True:
            x = rng.randrange(p)
            y2 = (x * x * x + 3) % p
            y = pow(y2, (p + 1) // 4, p)
            if y * y % p == y2:
                return x, (y if rng.random() < 0.5 else p - y)

    def affine_add(a, b):
        (x1, y1), (x2, y2) = a, b
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    for trial in range(trials):
        a, b = rand_point(), rand_point()
        zz = rng.randrange(1, p)
        # accumulator = a in XYZZ with ZZ = zz^2, ZZZ = zz^3, everything in Montgomery form (x R)
        ZZv, ZZZv = zz * zz % p, zz * zz * zz % p
        Xv, Yv = a[0] * ZZv % p, a[1] * ZZZv % p
        neg = trial & 1
        mont_of = lambda v: v * R % p                      # noqa: E731
        X, Y, ZZ, ZZZ = lazy(mont_of(Xv)), lazy(mont_of(Yv)), lazy(mont_of(ZZv)), lazy(mont_of(ZZZv))
        extreme = trial % 7 == 0
        if extreme:             # limb extremes of a normal form (no longer the point a: the limb comparison only)
            ZZ = [M29] * 8 + [ZZ[8]]
        qx = normal_form(mont_of(b[0]))
        qy = normal_form(mont_of(b[1]))
        if neg:
            qy = [-v for v in qy]
        bb = (b[0], (p - b[1]) % p) if neg else b
        want = ref_madd(X, Y, ZZ, ZZZ, qx, qy, params)
        for odd in (False, True):
            rin = {}
            Yin = normal_form(-value(Y)) if odd else Y        # the odd body starts from a normal form of -Y
            if odd:
                want_o = ref_madd(X, [-v for v in Yin], ZZ, ZZZ, qx, qy, params)   # the definition on the value the register stands for
            for n, l in (("XA", X), ("Y", Yin), ("ZZ", ZZ), ("ZZZ", ZZZ), ("QX", qx), ("QY", qy)):
                for i in range(9):
                    rin["%s%d" % (n, i)] = l[i]
            for n in ("XB", "PP", "M"):
                for i in range(9):
                    rin["%s%d" % (n, i)] = rng.getrandbits(32)     # scratch starts as garbage
            out, masks = simulate(bodies[odd], rin, **params)
            got = {n: [out["%s%d" % (n, i)] for i in range(9)] for n in ("XB", "Y", "ZZ", "ZZZ")}
            w = want_o if odd else want
            assert got["XB"] == w[0], "X3 limbs"
            assert got["ZZ"] == w[2] and got["ZZZ"] == w[3], "ZZ3 / ZZZ3 limbs"
            # Y: even = a normal form of -Y3, odd = +Y3; same value mod p as the definition's, and a Montgomery output's range
            ysign = 1 if odd else -1
            assert (ysign * value(got["Y"]) - value(w[1])) % p == 0, "Y3 value"
            assert -p // 4 < value(got["Y"]) < 3 * p and all(0 <= v <= M29 for v in got["Y"][:8]), "Y3 range"
            assert not masks["z"] and not masks["z2"]
            if extreme:
                continue
            # the group law: (X3 / ZZ3, Y3 / ZZZ3) == a + b
            x3 = value(got["XB"]) * Rinv * pow(value(got["ZZ"]) * Rinv, -1, p) % p
            y3 = ysign * value(got["Y"]) * Rinv * pow(value(got["ZZZ"]) * Rinv, -1, p) % p
            assert (x3, y3) == affine_add(a, bb), "not the sum"
    # same x: the masks (exact comparison; PP's limb 0 alone may also coincide by chance -- forced below)
    for trial in range(40):
        a = rand_point()
        zz = rng.randrange(1, p)
        ZZv, ZZZv = zz * zz % p, zz * zz * zz % p
        X, Y = lazy(a[0] * ZZv % p * R % p), lazy(a[1] * ZZZv % p * R % p)
        ZZ, ZZZ = lazy(ZZv * R % p), lazy(ZZZv * R % p)
        same = trial & 1
        qx = normal_form(a[0] * R % p)
        qy = normal_form((a[1] if same else p - a[1]) * R % p)
        for odd in (False, True):
            rin = {}
            Yin = normal_form(-value(Y)) if odd else Y
            for n, l in (("XA", X), ("Y", Yin), ("ZZ", ZZ), ("ZZZ", ZZZ), ("QX", qx), ("QY", qy)):
                for i in range(9):
                    rin["%s%d" % (n, i)] = l[i]
            for n in ("XB", "PP", "M"):
                for i in range(9):
                    rin["%s%d" % (n, i)] = rng.getrandbits(32)
            out, masks = simulate(bodies[odd], rin, **params)
            assert masks["z"] and masks["z2"] == bool(same), "same-x masks"
    # a filter hit that is not a zero: PP with limb 0 == 0 or == p0 but another limb off
    q = Prog()
    exact_zero(q, regs("V"), "z", "s1", "s2", regs("M"))
    for base in ([0] * 9, list(params["P29"])):
        for i in range(-1, 9):
            v = list(base)
            if i >= 0:
                v[i] ^= 1 << rng.randrange(29)
            rin = {"V%d" % j: v[j] for j in range(9)}
            rin.update({"M%d" % j: rng.getrandbits(32) for j in range(9)})
            _, masks = simulate(q, rin, **params)
            assert masks["z"] == (i < 0), "exact comparison"
    counts = {("odd" if o else "even"): b.valu() for o, b in bodies.items()}
    print("madd bodies verified on the CPU; VALU instructions on the hot path:", counts)
    return counts


# ---- rendering ---------------------------------------------------------------------------------------------------------
def render(q, name, groups_io, groups_out, doc):
    def reg(x):
        if x == "ACC":
            return ACC_LO
        if isinstance(x, int):
            return str(x)
        return "%[" + x + "]"

    lines = []
    depth = []
    label = 0
    for ins in q.ins:
        op = ins[0]
        if op == "mad":
            lines.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC, reg(ins[1]), reg(ins[2]), "0" if ins[3] else ACC))
        elif op == "mullo":
            lines.append("v_mul_lo_u32 %s, %s, %s" % (reg(ins[1]), ACC_LO, reg(ins[2])))
        elif op == "and29":
            lines.append("v_and_b32 %s, %s, %s" % (reg(ins[1]), reg("mask"), reg(ins[2])))
        elif op == "ashr":
            lines.append("v_ashrrev_i64 %s, 29, %s" % (ACC, ACC))
        elif op == "ashr32":
            lines.append("v_ashrrev_i32 %s, 29, %s" % (reg(ins[1]), reg(ins[2])))
        elif op == "movlo":
            lines.append("v_mov_b32 %s, %s" % (reg(ins[1]), ACC_LO))
        elif op == "shr29lo":
            lines.append("v_alignbit_b32 %s, v5, %s, 29" % (reg(ins[1]), ACC_LO))
        elif op == "shl1":
            lines.append("v_lshlrev_b32 %s, 1, %s" % (reg(ins[1]), reg(ins[2])))
        elif op == "sub":
            lines.append("v_sub_u32 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "add":
            lines.append("v_add_u32 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "lshl1_add":
            lines.append("v_lshl_add_u32 %s, %s, 1, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "cmp_eq":
            lines.append("v_cmp_eq_u32_e64 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "s_and":
            lines.append("s_and_b64 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "s_or":
            lines.append("s_or_b64 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "s_mov0":
            lines.append("s_mov_b64 %s, 0" % reg(ins[1]))
        elif op == "or3":
            lines.append("v_or3_b32 %s, %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3]), reg(ins[4])))
        elif op == "or2":
            lines.append("v_or_b32 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "xor":
            lines.append("v_xor_b32 %s, %s, %s" % (reg(ins[1]), reg(ins[2]), reg(ins[3])))
        elif op == "rare_begin":
            label += 1
            l = ".Lmadd_%s_%d_%%=" % (name, label)
            depth.append(l)
            if ins[1] == "scc0":              # the s_or in front set SCC = (result != 0)
                lines.append("s_cbranch_scc0 %s" % l)
            else:                              # "z==0"
                lines.append("s_cmp_eq_u64 %s, 0" % reg("z"))
                lines.append("s_cbranch_scc1 %s" % l)
        elif op == "rare_end":
            lines.append("%s:" % depth.pop())
        else:
            raise ValueError(op)
    args = ", ".join("int32_t* __restrict__ %s" % g for g in groups_io + groups_out[:1]) + ", uint64_t& same_x, uint64_t& same_point"
    o = [doc, "template <class PR>", "__device__ __forceinline__ void %s(%s) {" % (name, args)]
    o.append("    int32_t PP[9], M[9];")
    o.append("    uint64_t z, z2, s1, s2;")
    o.append("    asm volatile(")
    for l in lines:
        o.append('        "%s\\n\\t"' % l)
    outs = []
    for g in groups_io:
        outs += ['[%s%d] "+v"(%s[%d])' % (g, i, g, i) for i in range(9)]
    for g in groups_out:
        outs += ['[%s%d] "=&v"(%s[%d])' % (g, i, g, i) for i in range(9)]
    outs += ['[z] "=&s"(z)', '[z2] "=&s"(z2)', '[s1] "=&s"(s1)', '[s2] "=&s"(s2)']
    inps = ['[p%d] "s"((int32_t)PR::P29[%d])' % (i, i) for i in range(9)]
    inps += ['[inv] "s"((int32_t)PR::INV29)', '[mask] "s"(0x1fffffff)']
    o.append("        : " + ", ".join(outs))
    o.append("        : " + ", ".join(inps))
    o.append('        : "vcc", "scc", "v4", "v5");')
    o.append("    same_x = z;")
    o.append("    same_point = z2;")
    o.append("}")
    return "\n".join(o)


HEADER = """// madd_asm.inc -- GENERATED by tools/gen_madd_asm.py; do not edit.
//
// The mixed XYZZ addition of the bucket accumulation over Fq (k_msm_accumulate<Fq>) as ONE inline-asm body, multipliers in place;
// replaces Sum / Add for G1Local (/root/reference/src/groth16/fr.rs:191-198) over the terms of exp_encrypted_g1 (fr.rs:114-119).
// Same formulas and the same limbs as madd_xyzz_nz (lazy29.cuh); verified on the CPU by the generator (hardware wrap-around semantics,
// register aliasing, column bounds, the group law).  Accumulator v[4:5] fixed, as in mont_asm.inc.
"""

DOC_E = """// EVEN body: (XA, Y, ZZ, ZZZ) += (QX, QY).  Out: X3 in XB, -Y3 (!) in Y, ZZ3, ZZZ3 in place; XA, QX, QY are clobbered.
// same_x: lanes whose point has the accumulator's x (their outputs are garbage: the sum is 2 P or infinity); same_point: of those, the
// lanes with the same y."""
DOC_O = """// ODD body: the accumulator's Y register holds -Y (what the even body leaves).  Out: X3 in XB, +Y3 in Y, ZZ3, ZZZ3 in place."""


def write(path):
    parts = [HEADER]
    io = ["XA", "Y", "ZZ", "ZZZ", "QX", "QY"]
    parts.append(render(gen_madd_g1(False), "madd_asm_g1_even", io, ["XB", "PP", "M"], DOC_E))
    parts.append(render(gen_madd_g1(True), "madd_asm_g1_odd", io, ["XB", "PP", "M"], DOC_O))
    open(path, "w").write("\n\n".join(parts) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    check()
    if "--check" not in sys.argv:
        write(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zksnark_rs_amd", "csrc", "madd_asm.inc"))
