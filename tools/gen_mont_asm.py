#!/usr/bin/env python3
"""Generator of the hand-scheduled Montgomery multipliers of the bucket accumulation (csrc/mont_asm.inc).

    python tools/gen_mont_asm.py            # check the schedules on the CPU, then write zksnark_rs_amd/csrc/mont_asm.inc
    python tools/gen_mont_asm.py --check    # only the check

Why: the accumulation loop is bound by VALU issue, and hipcc's code for the product-scanning multiplier of lazy29.cuh keeps the
carry chain and the column sum in two 64-bit accumulators that it joins with one v_lshl_add_u64 per column (17 per reduction,
137 of the 2262 instructions of a G1 addition), subtracts the second product of mont_diff with 64-bit v_sub_co / v_subb_co pairs
and copies operands between the unrolled iterations.  The sequences below are the same algorithm with ONE accumulator per
column chain: the first multiply-add of a column takes the shifted carry of the previous one as its addend, m_k lives in the
register that later receives output limb k, and nothing else is issued: 162 + 43 instructions per multiplication (round 6: the
last column's 64-bit shift and the copy of its low half are one v_alignbit_b32).

Every sequence is a list of abstract instructions; `simulate` executes the list on Python integers with the wrap-around
semantics of the hardware instructions (so register reuse and ordering are checked here, on the CPU), and `render` prints it
as one inline-asm statement.  The 64-bit accumulators are FIXED registers (v[4:5], v[6:7]): gfx950 inline asm has no operand
modifier that names the low half of a 64-bit operand, which v_mul_lo_u32 / v_and_b32 need.
"""
import random
import sys
import os

M29 = (1 << 29) - 1
ACC = ["v[4:5]", "v[6:7]"]
ACC_LO = ["v4", "v6"]
ACC_HI = ["v5", "v7"]
CLOBBER = {0: ["v4", "v5"], 1: ["v6", "v7"]}


def s32(x):
    x &= 0xffffffff
    return x - (1 << 32) if x >> 31 else x


def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


class Seq:
    """instructions: (op, dst, srcs...) over symbolic 32-bit registers (strings) and accumulators (ints)"""

    def __init__(self):
        self.ins = []

    def mad(self, acc, x, y, fresh=False):      # acc = x * y + (0 if fresh else acc), signed 32 x 32 + 64
        self.ins.append(("mad", acc, x, y, fresh))

    def mullo(self, d, acc, s):                  # d = lo32(acc) * s  (mod 2^32)
        self.ins.append(("mullo", d, acc, s))

    def and29(self, d, src):                     # d = src & M29; src: register name or ("lo", acc)
        self.ins.append(("and29", d, src))

    def ashr(self, acc):                         # acc >>= 29 (arithmetic)
        self.ins.append(("ashr", acc))

    def movlo(self, d, acc):                     # d = lo32(acc)
        self.ins.append(("movlo", d, acc))

    def shr29lo(self, d, acc):                   # d = lo32(acc >> 29): ONE v_alignbit_b32 instead of the last column's 64-bit shift + v_mov
        self.ins.append(("shr29lo", d, acc))

    def shl1(self, d, s):
        self.ins.append(("shl1", d, s))

    def neg(self, d, s):
        self.ins.append(("neg", d, s))


def column_range(k):
    return (0 if k < 9 else k - 8), (k if k < 9 else 8)


def reduce_column(q, acc, k, m, r):
    """the Montgomery part of column k on accumulator `acc`: m[i] * p[k-i] terms, then m_k or the output limb, then the shift
    (column 16: the shifted value IS output limb 8 -- one v_alignbit_b32 of the accumulator's halves)"""
    lo, hi = column_range(k)
    for i in range(lo, hi + 1):
        if i < k or k >= 9:
            q.mad(acc, m[i], "p%d" % (k - i))
    if k < 9:
        q.mullo(m[k], acc, "inv")
        q.and29(m[k], m[k])
        q.mad(acc, m[k], "p0")
    else:
        q.and29(r[k - 9], ("lo", acc))
    if k == 16:
        q.shr29lo(r[8], acc)
    else:
        q.ashr(acc)


def gen_mul(kind):
    """kind: 'mul' a*b | 'sqr' a*a | 'sum' a*b + c*d.  Outputs r0..r8 (they hold m_0..m_8 until their column has passed)."""
    q = Seq()
    r = ["r%d" % i for i in range(9)]
    if kind == "sqr":
        for i in range(8):
            q.shl1("t%d" % i, "a%d" % i)
    first = True
    for k in range(17):
        lo, hi = column_range(k)
        for i in range(lo, hi + 1):
            j = k - i
            if kind == "sqr":
                if i < j:
                    q.mad(0, "t%d" % i, "a%d" % j, first)
                elif i == j:
                    q.mad(0, "a%d" % i, "a%d" % i, first)
                else:
                    continue
            else:
                q.mad(0, "a%d" % i, "b%d" % j, first)
            first = False
        if kind == "sum":
            for i in range(lo, hi + 1):
                q.mad(0, "c%d" % i, "d%d" % (k - i))
        reduce_column(q, 0, k, r, r)
    return q


def gen_fp2():
    """(a + b i)(c + d i): r = a c + b (-d), s = a d + b c; two column chains side by side.  t0..t8 = -d."""
    q = Seq()
    r = ["r%d" % i for i in range(9)]
    s = ["s%d" % i for i in range(9)]
    for i in range(9):
        q.neg("t%d" % i, "d%d" % i)
    first = True
    for k in range(17):
        lo, hi = column_range(k)
        for i in range(lo, hi + 1):
            j = k - i
            q.mad(0, "a%d" % i, "c%d" % j, first)
            q.mad(1, "a%d" % i, "d%d" % j, first)
            first = False
            q.mad(0, "b%d" % i, "t%d" % j)
            q.mad(1, "b%d" % i, "c%d" % j)
        # the two reductions interleaved instruction by instruction
        q0, q1 = Seq(), Seq()
        reduce_column(q0, 0, k, r, r)
        reduce_column(q1, 1, k, s, s)
        for x, y in zip(q0.ins, q1.ins):
            q.ins.append(x)
            q.ins.append(y)
    return q


# ---- simulation ------------------------------------------------------------------------------------
def simulate(q, regs, P29, INV29):
    regs = dict(regs)
    for i in range(9):
        regs["p%d" % i] = P29[i]
    regs["inv"] = INV29
    acc = [None, None]
    for ins in q.ins:
        op = ins[0]
        if op == "mad":
            _, a, x, y, fresh = ins
            base = 0 if fresh else acc[a]
            acc[a] = s64(s32(regs[x]) * s32(regs[y]) + base)
            # the design bound: no column may leave the signed 64-bit range
            assert abs(s32(regs[x]) * s32(regs[y]) + base) < (1 << 63), "column overflow"
        elif op == "mullo":
            _, d, a, s = ins
            regs[d] = ((acc[a] & 0xffffffff) * (regs[s] & 0xffffffff)) & 0xffffffff
        elif op == "and29":
            _, d, src = ins
            v = (acc[src[1]] & 0xffffffff) if isinstance(src, tuple) else regs[src]
            regs[d] = v & M29
        elif op == "ashr":
            acc[ins[1]] = acc[ins[1]] >> 29
        elif op == "movlo":
            regs[ins[1]] = s32(acc[ins[2]])
        elif op == "shr29lo":
            regs[ins[1]] = s32(acc[ins[2]] >> 29)
        elif op == "shl1":
            regs[ins[1]] = s32(regs[ins[2]] << 1)
        elif op == "neg":
            regs[ins[1]] = s32(-regs[ins[2]])
        else:
            raise ValueError(op)
    return regs


def model(prods, P29, INV29):
    """lazy29.cuh FpR::mont / mont_diff as written there: prods = list of (x[9], y[9]) whose products are summed"""
    m = [0] * 9
    r = [0] * 9
    carry = 0
    for k in range(17):
        lo, hi = column_range(k)
        acc = carry
        for x, y in prods:
            for i in range(lo, hi + 1):
                acc += x[i] * y[k - i]
        for i in range(lo, hi + 1):
            if i < k or k >= 9:
                acc += m[i] * P29[k - i]
        if k < 9:
            m[k] = ((acc & 0xffffffff) * INV29) & M29
            acc += m[k] * P29[0]
        else:
            r[k - 9] = acc & M29
        carry = acc >> 29
    r[8] = carry
    return r


def value(l):
    return sum(v << (29 * i) for i, v in enumerate(l))


FQ = dict(P29=[0x187cfd47, 0x010460b6, 0x1c72a34f, 0x02d522d0, 0x1585d978, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e], INV29=0x04866389)
FR = dict(P29=[0x10000001, 0x1f0fac9f, 0x0e5c2450, 0x07d090f3, 0x1585d283, 0x02db40c0, 0x00a6e141, 0x0e5c2634, 0x0030644e], INV29=0x0fffffff)


def rand_limbs(rng, bound_bits, signed, top_small=True):
    v = []
    for i in range(9):
        b = bound_bits if (i < 8 or not top_small) else 23
        x = rng.getrandbits(b)
        if rng.random() < 0.05:
            x = (1 << b) - 1          # the extremes
        if signed and rng.random() < 0.5:
            x = -x
        v.append(x)
    return v


def check():
    rng = random.Random(20260930)
    for params in (FQ, FR):
        p = value(params["P29"])
        Rinv = pow(1 << 261, -1, p)
        for trial in range(300):
            a = rand_limbs(rng, 29, True)
            b = rand_limbs(rng, 29, True)
            c = rand_limbs(rng, 29, True)
            d = rand_limbs(rng, 29, True)
            wide = rand_limbs(rng, 30, True)     # one side of a plain product may carry |limb| <= 2^30
            regs = {}
            for n, l in (("a", a), ("b", b), ("c", c), ("d", d)):
                for i in range(9):
                    regs["%s%d" % (n, i)] = l[i]
            # mul (one wide side)
            rw = dict(regs)
            for i in range(9):
                rw["a%d" % i] = wide[i]
            out = simulate(gen_mul("mul"), rw, **params)
            got = [out["r%d" % i] for i in range(9)]
            assert got == model([(wide, b)], **params), "mul"
            assert value(got) % p == value(wide) * value(b) * Rinv % p
            # sqr
            out = simulate(gen_mul("sqr"), regs, **params)
            got = [out["r%d" % i] for i in range(9)]
            assert got == model([(a, a)], **params), "sqr"
            # sum
            out = simulate(gen_mul("sum"), regs, **params)
            got = [out["r%d" % i] for i in range(9)]
            assert got == model([(a, b), (c, d)], **params), "sum"
            assert value(got) % p == (value(a) * value(b) + value(c) * value(d)) * Rinv % p
            # fp2
            out = simulate(gen_fp2(), regs, **params)
            nd = [-x for x in d]
            assert [out["r%d" % i] for i in range(9)] == model([(a, c), (b, nd)], **params), "fp2 re"
            assert [out["s%d" % i] for i in range(9)] == model([(a, d), (b, c)], **params), "fp2 im"
    counts = {k: len(gen_mul(k).ins) for k in ("mul", "sqr", "sum")}
    counts["fp2"] = len(gen_fp2().ins)
    print("schedules verified on the CPU; instructions per call:", counts)


# ---- rendering ----------------------------------------------------------------------------------------
def render(q, name, ins_groups, out_groups, tmp_groups, naccs):
    """ins_groups: [(prefix, c_expr)], each 9 limbs of int32_t; outputs likewise (early clobber); temporaries: [(prefix, count)]"""
    lines = []

    def reg(x):
        return "%[" + x + "]"

    for ins in q.ins:
        op = ins[0]
        if op == "mad":
            _, a, x, y, fresh = ins
            lines.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (ACC[a], reg(x), reg(y), "0" if fresh else ACC[a]))
        elif op == "mullo":
            lines.append("v_mul_lo_u32 %s, %s, %s" % (reg(ins[1]), ACC_LO[ins[2]], reg(ins[3])))
        elif op == "and29":
            src = ACC_LO[ins[2][1]] if isinstance(ins[2], tuple) else reg(ins[2])
            lines.append("v_and_b32 %s, %s, %s" % (reg(ins[1]), reg("mask"), src))
        elif op == "ashr":
            lines.append("v_ashrrev_i64 %s, 29, %s" % (ACC[ins[1]], ACC[ins[1]]))
        elif op == "movlo":
            lines.append("v_mov_b32 %s, %s" % (reg(ins[1]), ACC_LO[ins[2]]))
        elif op == "shr29lo":
            lines.append("v_alignbit_b32 %s, %s, %s, 29" % (reg(ins[1]), ACC_HI[ins[2]], ACC_LO[ins[2]]))
        elif op == "shl1":
            lines.append("v_lshlrev_b32 %s, 1, %s" % (reg(ins[1]), reg(ins[2])))
        elif op == "neg":
            lines.append("v_sub_u32 %s, 0, %s" % (reg(ins[1]), reg(ins[2])))
    args = ", ".join(["int32_t* __restrict__ %s" % p for p, _ in out_groups] + ["const int32_t* __restrict__ %s" % p for p, _ in ins_groups])
    o = []
    o.append("template <class PR>")
    o.append("__device__ __forceinline__ void %s(%s) {" % (name, args))
    for pfx, cnt in tmp_groups:
        o.append("    int32_t %s_[%d];" % (pfx, cnt))
    o.append("    asm(")
    for l in lines:
        o.append('        "%s\\n\\t"' % l)
    outs = []
    for pfx, _ in out_groups:
        outs += ['[%s%d] "=&v"(%s[%d])' % (pfx, i, pfx, i) for i in range(9)]
    for pfx, cnt in tmp_groups:
        outs += ['[%s%d] "=&v"(%s_[%d])' % (pfx, i, pfx, i) for i in range(cnt)]
    inps = []
    for pfx, _ in ins_groups:
        inps += ['[%s%d] "v"(%s[%d])' % (pfx, i, pfx, i) for i in range(9)]
    inps += ['[p%d] "s"((int32_t)PR::P29[%d])' % (i, i) for i in range(9)]
    inps += ['[inv] "s"((int32_t)PR::INV29)', '[mask] "s"(0x1fffffff)']
    o.append("        : " + ", ".join(outs))
    o.append("        : " + ", ".join(inps))
    clob = ["vcc"]
    for a in range(naccs):
        clob += CLOBBER[a]
    o.append("        : " + ", ".join('"%s"' % c for c in clob) + ");")
    o.append("}")
    return "\n".join(o)


HEADER = """// mont_asm.inc -- GENERATED by tools/gen_mont_asm.py; do not edit.
//
// Hand-scheduled Montgomery multipliers in the lazy radix-2^29 form of lazy29.cuh for the bucket accumulation
// (k_msm_accumulate; replaces the field arithmetic of /root/reference/src/groth16/fr.rs:18-71 inside the inner
// products of fr.rs:114-119,191-198).  Same algorithm, same limbs out as FpR::mont / mont_diff / Fp2R::operator*:
// one 64-bit accumulator per column chain, m_k kept in the register of output limb k.  The accumulators are the fixed
// registers v[4:5] (and v[6:7] for the second chain of the Fq2 product).  Schedules are verified on the CPU by the
// generator (wrap-around semantics of every instruction, register reuse, column bounds).
"""


def write(path):
    parts = [HEADER]
    parts.append(render(gen_mul("mul"), "mont_asm_mul", [("a", 9), ("b", 9)], [("r", 9)], [], 1))
    parts.append(render(gen_mul("sqr"), "mont_asm_sqr", [("a", 9)], [("r", 9)], [("t", 8)], 1))
    parts.append(render(gen_mul("sum"), "mont_asm_sum", [("a", 9), ("b", 9), ("c", 9), ("d", 9)], [("r", 9)], [], 1))
    parts.append(render(gen_fp2(), "mont_asm_fp2", [("a", 9), ("b", 9), ("c", 9), ("d", 9)], [("r", 9), ("s", 9)], [("t", 9)], 2))
    open(path, "w").write("\n\n".join(parts) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    check()
    if "--check" not in sys.argv:
        write(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zksnark_rs_amd", "csrc", "mont_asm.inc"))
