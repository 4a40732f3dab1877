// ec.cuh -- BN254 G1 / G2 group law for gfx950 (one point per lane, limbs in VGPRs).
//
// Replaces bn's G1/G2 `+`, `-`, `* Fr` reached through G1Local/G2Local and
// EllipticEncryptable::exp_encrypted_g1/g2 (/root/reference/src/groth16/fr.rs:114-119,175-223).
// y^2 = x^3 + b with a = 0; Jacobian coordinates (X:Y:Z), x = X/Z^2, y = Y/Z^3.
// The same templates serve G1 (F = Fq) and G2 (F = Fq2).
//
// Every formula is COMPLETE for this library's purposes: infinity, P+P and P+(-P) are handled
// explicitly, because proofs must be bit-identical to the CPU path on tiny circuits whose CRS
// legitimately contains infinity or repeated points (SURVEY.md 7 "Hard parts").
#pragma once
#include "ff.cuh"

namespace zk {

template <class F>
struct Aff {
    F x, y;  // infinity <=> x == 0 && y == 0 (never on the curve since b != 0)
    ZK_HD static Aff infinity() { return Aff{F::zero(), F::zero()}; }
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    ZK_HD Aff neg() const { return Aff{x, -y}; }
};

template <class F>
struct Jac {
    F X, Y, Z;
    ZK_HD static Jac infinity() { return Jac{F::zero(), F::one(), F::zero()}; }
    ZK_HD static Jac from_affine(const Aff<F>& a) { return a.is_inf() ? infinity() : Jac{a.x, a.y, F::one()}; }
    ZK_HD bool is_inf() const { return Z.is_zero(); }
    ZK_HD Jac neg() const { return Jac{X, -Y, Z}; }
};

// dbl-2009-l (a = 0): 2M + 5S
template <class F>
ZK_HD Jac<F> jac_dbl(const Jac<F>& p) {
    if (p.is_inf()) return p;
    F A = p.X.sqr(), B = p.Y.sqr(), C = B.sqr();
    F D = ((p.X + B).sqr() - A - C).dbl();
    F E = A.dbl() + A;
    F X3 = E.sqr() - D.dbl();
    F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
    F Z3 = (p.Y * p.Z).dbl();
    return Jac<F>{X3, Y3, Z3};
}

#define ZK_NI __host__ __device__ __attribute__((noinline))
// Out-of-line copies for latency-bound tail kernels and rare branches: one body per
// translation unit instead of one per call site (compile time and code size).
template <class F>
ZK_NI Jac<F> jac_dbl_ni(const Jac<F>& p) { return jac_dbl(p); }

// add-2007-bl: 11M + 5S
template <class F>
ZK_HD Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    F Z1Z1 = p.Z.sqr(), Z2Z2 = q.Z.sqr();
    F U1 = p.X * Z2Z2, U2 = q.X * Z1Z1;
    F S1 = p.Y * q.Z * Z2Z2, S2 = q.Y * p.Z * Z1Z1;
    if (U1 == U2) {
        if (S1 == S2) return jac_dbl_ni(p);
        return Jac<F>::infinity();
    }
    F H = U2 - U1;
    F I = H.dbl().sqr();
    F J = H * I;
    F rr = (S2 - S1).dbl();
    F V = U1 * I;
    F X3 = rr.sqr() - J - V.dbl();
    F Y3 = rr * (V - X3) - (S1 * J).dbl();
    F Z3 = ((p.Z + q.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return Jac<F>{X3, Y3, Z3};
}

// madd-2007-bl (Z2 = 1): 7M + 4S
template <class F>
ZK_HD Jac<F> jac_madd(const Jac<F>& p, const Aff<F>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return Jac<F>{q.x, q.y, F::one()};
    F Z1Z1 = p.Z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * p.Z * Z1Z1;
    if (U2 == p.X) {
        if (S2 == p.Y) return jac_dbl_ni(p);
        return Jac<F>::infinity();
    }
    F H = U2 - p.X;
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F J = H * I;
    F rr = (S2 - p.Y).dbl();
    F V = p.X * I;
    F X3 = rr.sqr() - J - V.dbl();
    F Y3 = rr * (V - X3) - (p.Y * J).dbl();
    F Z3 = (p.Z + H).sqr() - Z1Z1 - HH;
    return Jac<F>{X3, Y3, Z3};
}

// Same as jac_madd but with the (rare) doubling branch inlined: a hot loop that contains a call
// keeps its Jacobian accumulator in scratch memory so that it can be passed by reference, which
// costs a 96/192 B store per iteration (measured: 4-6 GB of WRITE_SIZE per accumulate launch).
template <class F>
ZK_HD Jac<F> jac_madd_nocall(const Jac<F>& p, const Aff<F>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return Jac<F>{q.x, q.y, F::one()};
    F Z1Z1 = p.Z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * p.Z * Z1Z1;
    if (U2 == p.X) {
        if (S2 == p.Y) return jac_dbl(p);
        return Jac<F>::infinity();
    }
    F H = U2 - p.X;
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F J = H * I;
    F rr = (S2 - p.Y).dbl();
    F V = p.X * I;
    F X3 = rr.sqr() - J - V.dbl();
    F Y3 = rr * (V - X3) - (p.Y * J).dbl();
    F Z3 = (p.Z + H).sqr() - Z1Z1 - HH;
    return Jac<F>{X3, Y3, Z3};
}

template <class F>
ZK_NI Jac<F> jac_add_ni(const Jac<F>& p, const Jac<F>& q) { return jac_add(p, q); }
template <class F>
ZK_NI Jac<F> jac_madd_ni(const Jac<F>& p, const Aff<F>& q) { return jac_madd(p, q); }

template <class F>
ZK_NI Aff<F> jac_to_affine(const Jac<F>& p) {
    if (p.is_inf()) return Aff<F>::infinity();
    F zi = p.Z.inv();
    F zi2 = zi.sqr();
    return Aff<F>{p.X * zi2, p.Y * zi2 * zi};
}

// The same with the data-dependent inversion (ff.cuh inv_vartime = division steps in batches of 30), for single-lane callers.  Z depends on the witness (it is a
// product of the differences met along the addition chain) and cannot be recovered from the affine point, so it is NOT public: what is
// inverted is Z * lambda for a fresh uniformly random lambda != 0 the caller supplies, whose distribution -- and with it the trip
// count of the loop -- is independent of Z; 1 / Z = lambda / (Z lambda).
template <class F>
ZK_NI Aff<F> jac_to_affine_vartime(const Jac<F>& p, const F& lambda) {
    if (p.is_inf()) return Aff<F>::infinity();
    F zi = (p.Z * lambda).inv_vartime() * lambda;
    F zi2 = zi.sqr();
    return Aff<F>{p.X * zi2, p.Y * zi2 * zi};
}

// k * P for a canonical 256-bit scalar given as 8 little-endian words; MSB-first
// double-and-add (the algorithm bn's Mul<Fr> uses [recollection]; the group element does not
// depend on it).
template <class F>
ZK_NI Jac<F> jac_mul_words(const Jac<F>& p, const uint32_t* k) {
    Jac<F> acc = Jac<F>::infinity();
    bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = jac_dbl_ni(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) {
            acc = jac_add_ni(acc, p);
            started = true;
        }
    }
    return acc;
}

// k * P for a small non-negative integer k
template <class F>
ZK_NI Jac<F> jac_mul_small(const Jac<F>& p, uint32_t k) {
    Jac<F> acc = Jac<F>::infinity();
    for (int i = 31 - __builtin_clz(k | 1); i >= 0; --i) {
        acc = jac_dbl_ni(acc);
        if ((k >> i) & 1) acc = jac_add_ni(acc, p);
    }
    return acc;
}

typedef Aff<Fq> G1A;
typedef Jac<Fq> G1J;
typedef Aff<Fq2> G2A;
typedef Jac<Fq2> G2J;

// Montgomery <-> canonical for whole points (ABI boundary)
ZK_HD G1A g1a_from_canonical(const G1A& p) { return G1A{Fq::from_canonical(p.x), Fq::from_canonical(p.y)}; }
ZK_HD G1A g1a_to_canonical(const G1A& p) { return G1A{p.x.to_canonical(), p.y.to_canonical()}; }
ZK_HD G2A g2a_from_canonical(const G2A& p) { return G2A{Fq2::from_canonical(p.x), Fq2::from_canonical(p.y)}; }
ZK_HD G2A g2a_to_canonical(const G2A& p) { return G2A{p.x.to_canonical(), p.y.to_canonical()}; }
ZK_HD G1A pt_from_canonical(const G1A& p) { return g1a_from_canonical(p); }
ZK_HD G2A pt_from_canonical(const G2A& p) { return g2a_from_canonical(p); }
ZK_HD G1A pt_to_canonical(const G1A& p) { return g1a_to_canonical(p); }
ZK_HD G2A pt_to_canonical(const G2A& p) { return g2a_to_canonical(p); }

}  // namespace zk
