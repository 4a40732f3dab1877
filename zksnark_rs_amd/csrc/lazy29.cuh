// lazy29.cuh -- register-resident field arithmetic for the MSM accumulation loop.
//
// The bucket accumulation (k_msm_accumulate) is >85 % of a proof and is bound by VALU
// instruction count.  Outside that loop field elements are 8 x 32-bit, fully reduced (ff.cuh):
// every multiply converts to 9 x 29-bit limbs and back (~60 instructions) and every add/sub is a
// carry chain + conditional correction (~50 instructions).  Inside the loop the accumulator and
// all temporaries stay in the multiplier's own radix instead:
//
//   FpR : 9 signed 32-bit limbs, value = sum v[i] 2^(29 i), a residue mod p (NOT reduced).
//         "normal form" (N): v[0..7] in [0, 2^29), v[8] small and signed; |value| < 16p.
//   add / sub are 9 independent v_add/v_sub (no carries, no reduction); a difference of two normal
//   forms has |limb| < 2^29 and can be multiplied directly; sums / longer combinations are brought
//   back to normal form with `norm` (carry propagation, 24 light instructions).
//   mul / sqr are the same product-scanning Montgomery as ff.cuh on SIGNED 64-bit accumulators
//   (v_mad_i64_i32): inputs with |limb| <= 2^30 on ONE side and < 2^29 on the other keep every
//   column sum below 2^63; the output is in normal form with value in (-p/4, 1.3 p) whatever
//   the (bounded) inputs were -- Montgomery reduction with R = 2^261 >> p contracts.
//
// The caller (msm_impl.hpp) loads points from the 8 x 32 tables, keeps the Jacobian accumulator in
// FpR for the whole bucket and stores the fully reduced 8 x 32 form at the end, so the change is
// invisible outside the kernel.  Bounds are argued at each use in madd_lazy below.
#pragma once
#include "ec.cuh"

// independent 64-bit accumulation chains per product column of the Montgomery multiplier
// (3: even / odd partial products and the reduction terms; 2: products / reduction; 1: one chain)
#ifndef ZK_MONT_CHAINS
#define ZK_MONT_CHAINS 2
#endif

// Hand-scheduled multipliers (mont_asm.inc, generated and checked by tools/gen_mont_asm.py): device code only; the C++ forms
// below stay the definition (host code, -DZK_MONT_ASM=0 builds for A/B) and produce the same limbs.
#ifndef ZK_MONT_ASM
#define ZK_MONT_ASM 1
#endif
#if ZK_MONT_ASM && defined(__HIP_DEVICE_COMPILE__)
#define ZK_MONT_ASM_ON 1
#else
#define ZK_MONT_ASM_ON 0
#endif

namespace zk {

#if ZK_MONT_ASM_ON
#include "mont_asm.inc"
#include "madd_asm.inc"
#endif

template <class PR>
struct FpR {
    typedef Fp<PR> Elem;
    int32_t v[9];
    static constexpr int32_t M29 = 0x1fffffff;

    ZK_HD static FpR load(const Fp<PR>& x) {   // canonical residue [0, p) -> normal form
        uint32_t t[9];
        Fp<PR>::to29(x.l, t);
        FpR r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = (int32_t)t[i];
        return r;
    }
    // limb-wise, no carries: results are NOT in normal form
    ZK_HD FpR operator+(const FpR& b) const {
        FpR r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = v[i] + b.v[i];
        return r;
    }
    ZK_HD FpR operator-(const FpR& b) const {
        FpR r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = v[i] - b.v[i];
        return r;
    }
    ZK_HD FpR neg() const {
        FpR r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.v[i] = -v[i];
        return r;
    }
    // carry propagation -> normal form (value unchanged)
    ZK_HD FpR norm() const {
        FpR r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int32_t t = v[i] + c;
            r.v[i] = t & M29;
            c = t >> 29;   // arithmetic
        }
        r.v[8] = v[8] + c;
        return r;
    }
    // a*b*2^-261 mod p in normal form.  Requires |a limbs| <= 2^30, |b limbs| < 2^29 (or vice versa).
    template <bool SQR>
    ZK_HD static FpR mont(const FpR& a, const FpR& b) {
#if ZK_MONT_ASM_ON
        FpR o;
        if (SQR) mont_asm_sqr<PR>(o.v, a.v); else mont_asm_mul<PR>(o.v, a.v, b.v);
        return o;
#else
        int32_t m[9], a2[9];
        if (SQR) {
#pragma unroll
            for (int i = 0; i < 9; ++i) a2[i] = a.v[i] * 2;
        }
        FpR r;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
#if ZK_MONT_CHAINS == 3
            int64_t acc0 = carry, acc1 = 0, acc2 = 0;
            if (SQR) {
#pragma unroll
                for (int i = lo; i <= hi; ++i) {
                    const int j = k - i;
                    if (i < j) { if (i & 1) acc1 += (int64_t)a2[i] * a.v[j]; else acc0 += (int64_t)a2[i] * a.v[j]; }
                    else if (i == j) acc1 += (int64_t)a.v[i] * a.v[i];
                }
            } else {
#pragma unroll
                for (int i = lo; i <= hi; ++i) { if (i & 1) acc1 += (int64_t)a.v[i] * b.v[k - i]; else acc0 += (int64_t)a.v[i] * b.v[k - i]; }
            }
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (i < k || k >= 9) acc2 += (int64_t)m[i] * (int32_t)PR::P29[k - i];
            int64_t acc = acc0 + acc1 + acc2;
#else
            int64_t acc0 = carry, acc2 = 0;
            if (SQR) {
#pragma unroll
                for (int i = lo; i <= hi; ++i) {
                    const int j = k - i;
                    if (i < j) acc0 += (int64_t)a2[i] * a.v[j];
                    else if (i == j) acc0 += (int64_t)a.v[i] * a.v[i];
                }
            } else {
#pragma unroll
                for (int i = lo; i <= hi; ++i) acc0 += (int64_t)a.v[i] * b.v[k - i];
            }
#if ZK_MONT_CHAINS == 2
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (i < k || k >= 9) acc2 += (int64_t)m[i] * (int32_t)PR::P29[k - i];
            int64_t acc = acc0 + acc2;
#else
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (i < k || k >= 9) acc0 += (int64_t)m[i] * (int32_t)PR::P29[k - i];
            int64_t acc = acc0;
            (void)acc2;
#endif
#endif
            if (k < 9) {
                m[k] = (int32_t)(((uint32_t)acc * PR::INV29) & (uint32_t)M29);
                acc += (int64_t)m[k] * (int32_t)PR::P29[0];
            } else {
                r.v[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)M29);
            }
            carry = acc >> 29;   // arithmetic; exact for k < 9 (low 29 bits are zero)
        }
        r.v[8] = (int32_t)carry;
        ZK_SCHED_FENCE();
        return r;
#endif
    }
    // (a*b - c*d) * 2^-261 mod p with ONE reduction (243 multiply-adds instead of 324).  Requires
    // |limb| < 2^29 on all four operands (normal forms or differences of two normal forms): a column
    // then holds at most 18 products < 2^58 plus 9 reduction terms < 2^58 plus the carry, < 2^63.
    // Output: normal form, |value| < 2 * (8p)^2 / 2^261 + p < 3p.
    ZK_HD static FpR mont_diff(const FpR& a, const FpR& b, const FpR& c, const FpR& d) {
#if ZK_MONT_ASM_ON
        FpR o;
        const FpR nc = c.neg();
        mont_asm_sum<PR>(o.v, a.v, b.v, nc.v, d.v);
        return o;
#else
        int32_t m[9];
        FpR r;
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
            int64_t acc0 = carry, acc1 = 0, acc2 = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) {
                acc0 += (int64_t)a.v[i] * b.v[k - i];
                acc1 -= (int64_t)c.v[i] * d.v[k - i];
            }
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (i < k || k >= 9) acc2 += (int64_t)m[i] * (int32_t)PR::P29[k - i];
            int64_t acc = acc0 + acc1 + acc2;
            if (k < 9) {
                m[k] = (int32_t)(((uint32_t)acc * PR::INV29) & (uint32_t)M29);
                acc += (int64_t)m[k] * (int32_t)PR::P29[0];
            } else {
                r.v[k - 9] = (int32_t)((uint32_t)acc & (uint32_t)M29);
            }
            carry = acc >> 29;
        }
        r.v[8] = (int32_t)carry;
        ZK_SCHED_FENCE();
        return r;
#endif
    }
    // (a*b + c*d) * 2^-261 mod p with one reduction; bounds as mont_diff
    ZK_HD static FpR mont_sum(const FpR& a, const FpR& b, const FpR& c, const FpR& d) {
#if ZK_MONT_ASM_ON
        FpR o;
        mont_asm_sum<PR>(o.v, a.v, b.v, c.v, d.v);
        return o;
#else
        return mont_diff(a, b, c.neg(), d);
#endif
    }
    ZK_HD FpR operator*(const FpR& b) const { return mont<false>(*this, b); }
    ZK_HD FpR sqr() const { return mont<true>(*this, *this); }

    // x == 0 (mod p)?  For x in normal form with |value| < 3p, i.e. any Montgomery output or the
    // double of one: the normal form of an integer is unique, so compare with k*p, k = -2..2.
    // Limb 0 filters (a false positive needs a 2^-29 coincidence) before the full comparison.
    ZK_HD bool is_zero_mod_p() const {
        bool maybe = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) maybe |= v[0] == PR::KP29[k][0];
        if (!maybe) return false;
        bool hit = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int32_t d = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) d |= v[i] ^ PR::KP29[k][i];
            hit |= d == 0;
        }
        return hit;
    }
    // fully reduced 8 x 32 form; accepts any value with |value| < 16p and limbs within int32 range.  (A Montgomery output is
    // within (-p/4, 1.3p), sums of a few of them stay below 8p -- but dbl_lazy's X3 = E^2 - 2D and Y3 = E(D - X3) - 8C reach
    // +-13p, and mul_small_lazy ends in a doubling for every even factor: with the former 8p range the weighted column / row
    // sums of the MSM tail were stored wrongly for such values.)
    ZK_HD Fp<PR> store_exact() const {
        FpR t;
#pragma unroll
        for (int i = 0; i < 9; ++i) t.v[i] = v[i] + PR::POSP29[0][i];   // + 16p: value in (0, 32p)
        t = t.norm();
        // conditional subtraction of 16p, 8p, 4p, 2p, p with borrow propagation in radix 2^29
#pragma unroll
        for (int sh = 0; sh < 5; ++sh) {
            int32_t d[9], br = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int32_t x = t.v[i] - PR::POSP29[sh][i] + br;
                d[i] = x & M29;
                br = x >> 29;
            }
            d[8] = t.v[8] - PR::POSP29[sh][8] + br;
            const bool ge = d[8] >= 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) t.v[i] = ge ? d[i] : t.v[i];
        }
        uint32_t u[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) u[i] = (uint32_t)t.v[i];
        Fp<PR> o;
        Fp<PR>::from29(u, o.l);
        return o;
    }
};

// Fq2 = Fq[i]/(i^2+1) over the lazy form; components of stored values are kept in normal form
template <class PR>
struct Fp2R {
    typedef Fq2 Elem;
    FpR<PR> c0, c1;
    ZK_HD static Fp2R load(const Fq2& x) { return Fp2R{FpR<PR>::load(x.c0), FpR<PR>::load(x.c1)}; }
    ZK_HD Fp2R operator+(const Fp2R& o) const { return Fp2R{c0 + o.c0, c1 + o.c1}; }
    ZK_HD Fp2R operator-(const Fp2R& o) const { return Fp2R{c0 - o.c0, c1 - o.c1}; }
    ZK_HD Fp2R neg() const { return Fp2R{c0.neg(), c1.neg()}; }
    ZK_HD Fp2R norm() const { return Fp2R{c0.norm(), c1.norm()}; }
    // operands: components with |limb| < 2^29 (normal forms or differences of two normal forms)
    // (a0 + a1 i)(b0 + b1 i) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) i with the two coordinates reduced side by
    // side: 4 x 81 product terms accumulate straight into the two column sums and only 2 x 81 reduction
    // terms follow.  Same multiply-add count as Karatsuba's three full multiplications (486), but none
    // of its limb-wise additions, normalisations and per-multiplication column overhead -- which at
    // this size cost as much as the 81 multiply-adds Karatsuba saves.
    // Columns: 18 products < 2^58 + 9 reduction terms < 2^58 + carry < 2^63.  Output in normal form,
    // |value| < 2 (8p)^2 / 2^261 + p < 2p.
    ZK_HD Fp2R operator*(const Fp2R& o) const {
#if ZK_MONT_ASM_ON
        Fp2R w;
        mont_asm_fp2<PR>(w.c0.v, w.c1.v, c0.v, c1.v, o.c0.v, o.c1.v);
        return w;
#else
        typedef FpR<PR> L;
        int32_t m0[9], m1[9], nb1[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) nb1[i] = -o.c1.v[i];
        Fp2R r;
        int64_t carry0 = 0, carry1 = 0;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
            int64_t p0 = carry0, p1 = carry1, q0 = 0, q1 = 0;
#pragma unroll
            for (int i = lo; i <= hi; ++i) {
                p0 += (int64_t)c0.v[i] * o.c0.v[k - i];
                p0 += (int64_t)c1.v[i] * nb1[k - i];
                p1 += (int64_t)c0.v[i] * o.c1.v[k - i];
                p1 += (int64_t)c1.v[i] * o.c0.v[k - i];
            }
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (i < k || k >= 9) {
                    q0 += (int64_t)m0[i] * (int32_t)PR::P29[k - i];
                    q1 += (int64_t)m1[i] * (int32_t)PR::P29[k - i];
                }
            int64_t acc0 = p0 + q0, acc1 = p1 + q1;
            if (k < 9) {
                m0[k] = (int32_t)(((uint32_t)acc0 * PR::INV29) & (uint32_t)L::M29);
                m1[k] = (int32_t)(((uint32_t)acc1 * PR::INV29) & (uint32_t)L::M29);
                acc0 += (int64_t)m0[k] * (int32_t)PR::P29[0];
                acc1 += (int64_t)m1[k] * (int32_t)PR::P29[0];
            } else {
                r.c0.v[k - 9] = (int32_t)((uint32_t)acc0 & (uint32_t)L::M29);
                r.c1.v[k - 9] = (int32_t)((uint32_t)acc1 & (uint32_t)L::M29);
            }
            carry0 = acc0 >> 29;
            carry1 = acc1 >> 29;
        }
        r.c0.v[8] = (int32_t)carry0;
        r.c1.v[8] = (int32_t)carry1;
        ZK_SCHED_FENCE();
        return r;
#endif
    }
    // (a0 + a1)(a0 - a1) + 2 a0 a1 i; the doubled limbs (<= 2^30) sit on one side of the second product
    ZK_HD Fp2R sqr() const {
        FpR<PR> d = (c0 + c1).norm() * (c0 - c1);       // (N) x (|limb| < 2^30)
        return Fp2R{d, (c0 + c0) * c1};
    }
    ZK_HD bool is_zero_mod_p() const { return c0.is_zero_mod_p() && c1.is_zero_mod_p(); }
    ZK_HD Fq2 store_exact() const { return Fq2{c0.store_exact(), c1.store_exact()}; }
};

template <class F> struct LazyOf;
template <> struct LazyOf<Fq> { typedef FpR<FqParams> type; };
template <> struct LazyOf<Fq2> { typedef Fp2R<FqParams> type; };

// Jacobian accumulator in lazy form; `inf` replaces the Z == 0 test
template <class F>
struct JacR {
    typename LazyOf<F>::type X, Y, Z;
    bool inf;
};

// acc += (qx, qy), mixed Jacobian + affine addition without the constant factors of madd-2007-bl
// (the result differs from it by the projective scaling lambda = 2):
//   Z1Z1 = Z1^2, U2 = x2 Z1Z1, S2 = y2 Z1 Z1Z1, H = U2 - X1, R = S2 - Y1,
//   HH = H^2, HHH = H HH, W = X1 HH,
//   X3 = R^2 - HHH - 2W,  Y3 = R (W - X3) - Y1 HHH,  Z3 = Z1 H                     (8M + 3S)
// Bounds: X1, Y1, Z1, x2, y2 in normal form; every product is in normal form; H, R are
// differences of two normal forms (|limb| < 2^29) and are multiplied directly, as is W - X3; X3
// (4 terms) and Y3 (it feeds R = S2 - Y1, which is squared) are normalised.
// Returns false when the caller must take the slow path (P == Q: doubling).
template <class F>
ZK_HD bool madd_lazy(JacR<F>& p, const typename LazyOf<F>::type& qx, const typename LazyOf<F>::type& qy) {
    typedef typename LazyOf<F>::type L;
    if (p.inf) {
        p.X = qx; p.Y = qy.norm(); p.inf = false;   // qy may be a negated point (limbs <= 0): R = S2 - Y1 must stay below 2^29
        // Z = 1 in Montgomery form
        p.Z = L::load(F::one());
        return true;
    }
    L Z1Z1 = p.Z.sqr();
    L U2 = qx * Z1Z1;
    L S2 = (qy * p.Z) * Z1Z1;
    L H = U2 - p.X;
    L R = S2 - p.Y;
    L HH = H.sqr();
    if (HH.is_zero_mod_p()) {               // H == 0 (mod p): same x coordinate
        if (R.sqr().is_zero_mod_p()) return false;   // same point: doubling, slow path
        p.inf = true;                        // P + (-P)
        return true;
    }
    L HHH = H * HH;
    L W = p.X * HH;
    L X3 = (R.sqr() - HHH - (W + W)).norm();
    L Y3 = (R * (W - X3) - p.Y * HHH).norm();   // W - X3: difference of two normal forms, multiplied directly
    p.Z = p.Z * H;
    p.X = X3;
    p.Y = Y3;
    return true;
}

// ---- bucket accumulator of the MSMs: extended Jacobian (XYZZ) coordinates ------------------------
// x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2.  Mixed addition (madd-2008-s) needs no Z1^2 / Z1^3:
//   U2 = x2 ZZ1, S2 = y2 ZZZ1, P = U2 - X1, R = S2 - Y1, PP = P^2, PPP = P PP, Q = X1 PP,
//   X3 = R^2 - PPP - 2Q,  Y3 = R (Q - X3) - Y1 PPP,  ZZ3 = ZZ1 PP,  ZZZ3 = ZZZ1 PPP      (8M + 2S)
// Over Fq, Y3's two products share one Montgomery reduction (mont_diff): 1467 multiply-adds against
// 1674 for the Jacobian form.  Over Fq2 the four coordinates take 72 registers: the kernel sits at the
// 256-register limit with 16 dwords of scratch, and is still 6 % faster than the Jacobian accumulator.
// L = FpR<..> (G1) or Fp2R<..> (G2).
template <class L>
struct XyzzR {
    L X, Y, ZZ, ZZZ;
    bool inf;
};

// R D - Y PPP: one reduction over Fq; over Fq2 the fused form would need 36 products per column (> 2^63)
template <class PR>
ZK_HD FpR<PR> xyzz_ydiff(const FpR<PR>& R, const FpR<PR>& D, const FpR<PR>& Y, const FpR<PR>& PPP) { return FpR<PR>::mont_diff(R, D, Y, PPP); }
template <class PR>
ZK_HD Fp2R<PR> xyzz_ydiff(const Fp2R<PR>& R, const Fp2R<PR>& D, const Fp2R<PR>& Y, const Fp2R<PR>& PPP) { return (R * D - Y * PPP).norm(); }

// Returns false when the caller must take the slow path (P == Q: doubling).
template <class L>
ZK_HD bool madd_xyzz(XyzzR<L>& p, const L& qx, const L& qy) {
    if (p.inf) {
        p.X = qx; p.Y = qy.norm(); p.inf = false;   // qy may be a negated point (limbs <= 0): R = S2 - Y1 must stay below 2^29
        p.ZZ = p.ZZZ = L::load(L::Elem::one());
        return true;
    }
    L U2 = qx * p.ZZ;
    L S2 = qy * p.ZZZ;
    L P = U2 - p.X;                          // differences of two normal forms: |limb| < 2^29
    L R = S2 - p.Y;
    L PP = P.sqr();
    if (PP.is_zero_mod_p()) {                // same x coordinate
        if (R.sqr().is_zero_mod_p()) return false;
        p.inf = true;                        // P + (-P)
        return true;
    }
    L PPP = P * PP;
    L Q = p.X * PP;
    L X3 = (R.sqr() - PPP - (Q + Q)).norm();
    p.Y = xyzz_ydiff(R, Q - X3, p.Y, PPP);   // Q - X3: difference of two normal forms
    p.X = X3;
    p.ZZ = p.ZZ * PP;
    p.ZZZ = p.ZZZ * PPP;
    return true;
}

// The same addition for an accumulator that is known NOT to be infinity (the hot loop of k_msm_accumulate: a run starts from its
// first point, so the test never has to sit inside the loop).  Returns 0 when the sum was formed, 1 when P == Q (the caller doubles),
// 2 when P == -Q (the sum is infinity; the accumulator is left as it was).
template <class L>
ZK_HD int madd_xyzz_nz(XyzzR<L>& p, const L& qx, const L& qy) {
    L U2 = qx * p.ZZ;
    L S2 = qy * p.ZZZ;
    L P = U2 - p.X;                          // differences of two normal forms: |limb| < 2^29
    L R = S2 - p.Y;
    L PP = P.sqr();
    if (PP.is_zero_mod_p()) return R.sqr().is_zero_mod_p() ? 1 : 2;
    L PPP = P * PP;
    L Q = p.X * PP;
    L X3 = (R.sqr() - PPP - (Q + Q)).norm();
    p.Y = xyzz_ydiff(R, Q - X3, p.Y, PPP);   // Q - X3: difference of two normal forms
    p.X = X3;
    p.ZZ = p.ZZ * PP;
    p.ZZZ = p.ZZZ * PPP;
    return 0;
}

// The SECOND point of a run: the accumulator still is its first point (ZZ = ZZZ = 1), so the sum is an affine + affine addition --
// U2 = x2, S2 = y2, ZZ3 = PP, ZZZ3 = PPP: four multiplications of ten less, once per bucket (26 entries per bucket in the A and B
// products of a 2^20-gate proof).  Called from the PEELED second trip of k_msm_accumulate only: as a branch inside the hot loop the
// same saving was paid back in register moves at the joins (profiles/r5_experiments.txt item 7).
// Returns 0 when the sum was formed, 1 when P == Q (the caller doubles), 2 when P == -Q (the sum is infinity; coordinates untouched).
// qy may be a negated point (limbs <= 0): normalised before R = y2 - Y1 (|limb| < 2^29 is what the squaring and mont_diff need).
template <class L>
ZK_HD int madd_xyzz_second(XyzzR<L>& p, const L& qx, const L& qy) {
    L P = qx - p.X;
    L R = qy.norm() - p.Y;
    L PP = P.sqr();
    if (PP.is_zero_mod_p()) return R.sqr().is_zero_mod_p() ? 1 : 2;
    L PPP = P * PP;
    L Q = p.X * PP;
    L X3 = (R.sqr() - PPP - (Q + Q)).norm();
    p.Y = xyzz_ydiff(R, Q - X3, p.Y, PPP);
    p.X = X3;
    p.ZZ = PP;
    p.ZZZ = PPP;
    return 0;
}

// Jacobian point with the same affine image: (X ZZ^2, Y ZZZ^2, ZZZ)
template <class L>
ZK_HD Jac<typename L::Elem> xyzz_store(const XyzzR<L>& p) {
    typedef typename L::Elem F;
    if (p.inf) return Jac<F>::infinity();
    L z2 = p.ZZ.sqr(), z3 = p.ZZZ.sqr();
    return Jac<F>{(p.X * z2).store_exact(), (p.Y * z3).store_exact(), p.ZZZ.store_exact()};
}
template <class L>
ZK_HD XyzzR<L> xyzz_from_jac(const Jac<typename L::Elem>& j) {
    XyzzR<L> r;
    r.inf = j.is_inf();
    L z = L::load(j.Z);
    r.X = L::load(j.X); r.Y = L::load(j.Y);
    r.ZZ = z.sqr();
    r.ZZZ = r.ZZ * z;
    return r;
}

// p + q, both XYZZ (add-2008-s, 12M + 2S); operands hold normal forms (accumulator images)
template <class L>
ZK_HD XyzzR<L> add_xyzz(const XyzzR<L>& p, const XyzzR<L>& q) {
    typedef typename L::Elem F;
    if (p.inf) return q;
    if (q.inf) return p;
    L U1 = p.X * q.ZZ, U2 = q.X * p.ZZ;
    L S1 = p.Y * q.ZZZ, S2 = q.Y * p.ZZZ;
    L P = U2 - U1, R = S2 - S1;
    L PP = P.sqr();
    XyzzR<L> r;
    if (PP.is_zero_mod_p()) {
        if (R.sqr().is_zero_mod_p()) {       // same point: double through the Jacobian formulas (rare)
            JacR<F> j = dbl_lazy(jacr_load(xyzz_store(p)));
            return xyzz_from_jac<L>(jacr_store(j));
        }
        r = p;
        r.inf = true;
        return r;
    }
    L PPP = P * PP;
    L Q = U1 * PP;
    L X3 = (R.sqr() - PPP - (Q + Q)).norm();
    r.inf = false;
    r.Y = xyzz_ydiff(R, Q - X3, S1, PPP);
    r.X = X3;
    r.ZZ = (p.ZZ * q.ZZ) * PP;
    r.ZZZ = (p.ZZZ * q.ZZZ) * PPP;
    return r;
}

// p = 2 p in place (dbl-2008-s-1 with a = 0: U = 2Y, V = U^2, W = U V, S = X V, M = 3 X^2, X3 = M^2 - 2S, Y3 = M (S - X3) - W Y,
// ZZ3 = V ZZ, ZZZ3 = W ZZZ).  U and M are sums of normal forms: normalised before they are multiplied (|limb| < 2^29).
template <class L>
ZK_HD void dbl_xyzz(XyzzR<L>& p) {
    if (p.inf) return;
    const L U = (p.Y + p.Y).norm();
    const L V = U.sqr();
    const L W = U * V;
    const L S = p.X * V;
    const L X2 = p.X.sqr();
    const L M = (X2 + X2 + X2).norm();
    const L X3 = (M.sqr() - (S + S)).norm();
    p.Y = xyzz_ydiff(M, S - X3, W, p.Y);
    p.X = X3;
    p.ZZ = V * p.ZZ;
    p.ZZZ = W * p.ZZZ;
}

// p += *q with the operand's coordinates fetched one at a time, each right before the products that use it (the fences keep the
// compiler from hoisting the loads): over Fq2 the by-value form holds both operands (144 registers) under its temporaries and, capped at
// the tail kernels' 168 registers, spilled 700 bytes per lane -- 1.9 GB per proof of scratch traffic in the row / column sums of the G2
// product.  Same formulas and bounds as add_xyzz.  (The shipped G2 fold goes one step further and keeps three of the sum's four
// coordinates in LDS: add_xyzz_from_parked, msm_impl.hpp; this form is its ZK_FOLD_PARK=0 alternative.)
template <class L>
__device__ __forceinline__ void add_xyzz_from(XyzzR<L>& p, const XyzzR<L>* q) {
    if (q->inf) return;
    if (p.inf) { p = *q; return; }
    // every value dies as early as the formulas allow (X1 after U1, U1 after Q, ...): at most six coordinates are alive at once
    L U1, P;
    {
        const L qzz = q->ZZ;
        U1 = p.X * qzz;
    }
    asm volatile("" ::: "memory");
    {
        const L qx = q->X;
        P = qx * p.ZZ - U1;
    }
    asm volatile("" ::: "memory");
    const L PP = P.sqr();
    if (PP.is_zero_mod_p()) {
        // same x (rare): opposite points, or the same point -- then 2 q is the sum, doubled in place (X1 is gone by now, and the
        // by-value form here is what made the kernel spill)
        const L qy = q->Y;
        const bool same = (qy * p.ZZZ - p.Y * q->ZZZ).sqr().is_zero_mod_p();
        if (same) { p = *q; dbl_xyzz(p); } else p.inf = true;
        return;
    }
    {
        const L qzz = q->ZZ;                 // fetched again (the line is in L1 / L2) rather than kept across the test
        p.ZZ = (p.ZZ * qzz) * PP;
    }
    asm volatile("" ::: "memory");
    const L Q = U1 * PP;
    const L PPP = P * PP;
    L S1, R;
    {
        const L qzzz = q->ZZZ;
        S1 = p.Y * qzzz;
        R = p.ZZZ;                           // Z1^3, needed once more for S2
        p.ZZZ = (R * qzzz) * PPP;
    }
    asm volatile("" ::: "memory");
    {
        const L qy = q->Y;
        R = qy * R - S1;
    }
    asm volatile("" ::: "memory");
    const L X3 = (R.sqr() - PPP - (Q + Q)).norm();
    p.Y = xyzz_ydiff(R, Q - X3, S1, PPP);
    p.X = X3;
}

// accumulator interface used by k_msm_accumulate / k_msm_merge
template <class F> struct AccOf { typedef XyzzR<typename LazyOf<F>::type> type; };
template <class L> ZK_HD void acc_clear(XyzzR<L>& a) { a.inf = true; a.X = a.Y = a.ZZ = a.ZZZ = L::load(L::Elem::zero()); }
template <class L> ZK_HD bool acc_madd(XyzzR<L>& a, const L& x, const L& y) { return madd_xyzz(a, x, y); }
template <class L> ZK_HD Jac<typename L::Elem> acc_store(const XyzzR<L>& a) { return xyzz_store(a); }
template <class L> ZK_HD void acc_load(XyzzR<L>& a, const Jac<typename L::Elem>& j) { a = xyzz_from_jac<L>(j); }
template <class L> ZK_HD XyzzR<L> acc_add(const XyzzR<L>& a, const XyzzR<L>& b) { return add_xyzz(a, b); }

// ---- general Jacobian addition / doubling in lazy form (reduction tail of the MSM) -------------
template <class F>
ZK_HD JacR<F> jacr_load(const Jac<F>& p) {
    typedef typename LazyOf<F>::type L;
    JacR<F> r;
    r.inf = p.is_inf();
    r.X = L::load(p.X); r.Y = L::load(p.Y); r.Z = L::load(p.Z);
    return r;
}
template <class F>
ZK_HD Jac<F> jacr_store(const JacR<F>& p) {
    return p.inf ? Jac<F>::infinity() : Jac<F>{p.X.store_exact(), p.Y.store_exact(), p.Z.store_exact()};
}

// dbl-2009-l (a = 0) with every intermediate kept multipliable (see the bounds in the header)
template <class F>
ZK_HD JacR<F> dbl_lazy(const JacR<F>& p) {
    typedef typename LazyOf<F>::type L;
    if (p.inf) return p;
    L A = p.X.sqr(), B = p.Y.sqr(), C = B.sqr();
    L t = (p.X + B).norm().sqr();
    L D = (t - A - C);
    D = (D + D).norm();
    L E = (A + A + A).norm();
    L X3 = (E.sqr() - D - D).norm();
    L c2 = (C + C).norm(), c4 = (c2 + c2).norm();
    L Y3 = (E * (D - X3) - (c4 + c4)).norm();
    L yz = p.Y * p.Z;
    JacR<F> r;
    r.inf = false;
    r.X = X3; r.Y = Y3; r.Z = (yz + yz).norm();
    return r;
}

// p + q, both Jacobian (12M + 4S, no constant factors):
//   U1 = X1 Z2^2, U2 = X2 Z1^2, S1 = Y1 Z2^3, S2 = Y2 Z1^3, H = U2 - U1, R = S2 - S1,
//   X3 = R^2 - H^3 - 2 U1 H^2,  Y3 = R (U1 H^2 - X3) - S1 H^3,  Z3 = Z1 Z2 H
template <class F>
ZK_HD JacR<F> add_lazy(const JacR<F>& p, const JacR<F>& q) {
    typedef typename LazyOf<F>::type L;
    if (p.inf) return q;
    if (q.inf) return p;
    L Z1Z1 = p.Z.sqr(), Z2Z2 = q.Z.sqr();
    L U1 = p.X * Z2Z2, U2 = q.X * Z1Z1;
    L S1 = (p.Y * q.Z) * Z2Z2, S2 = (q.Y * p.Z) * Z1Z1;
    L H = U2 - U1, R = S2 - S1;
    L HH = H.sqr();
    if (HH.is_zero_mod_p()) {
        if (R.sqr().is_zero_mod_p()) return dbl_lazy(p);
        JacR<F> r = p;
        r.inf = true;
        return r;
    }
    L HHH = H * HH;
    L V = U1 * HH;
    L X3 = (R.sqr() - HHH - (V + V)).norm();
    L Y3 = (R * (V - X3) - S1 * HHH).norm();
    JacR<F> r;
    r.inf = false;
    r.X = X3; r.Y = Y3; r.Z = (p.Z * q.Z) * H;
    return r;
}

// k * P for a small non-negative k
template <class F>
ZK_HD JacR<F> mul_small_lazy(const JacR<F>& p, uint32_t k) {
    JacR<F> acc = p;
    acc.inf = true;
    for (int i = 31 - __builtin_clz(k | 1); i >= 0; --i) {
        acc = dbl_lazy(acc);
        if ((k >> i) & 1) acc = add_lazy(acc, p);
    }
    return acc;
}

}  // namespace zk

