// verify.hip -- groth16::verify (/root/reference/src/groth16/mod.rs:299-320) and the pairing it
// needs (EllipticEncryptable::pairing = bn::pairing, /root/reference/src/groth16/fr.rs:120-122;
// GtLocal "+" = Gt multiplication, fr.rs:225-231).
//
// verify is not on the accelerated path: it is one call per proof with l+1 scalar multiplications
// and four pairings, CPU code in the reference and host code here (the same ff.cuh / ec.cuh
// arithmetic compiled for the host).  Optimal ate pairing on BN254: tower Fq2 = Fq[i]/(i^2+1),
// Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + i, Fq12 = Fq6[w]/(w^2 - v); D-type twist
// psi(x', y') = (x' w^2, y' w^3); affine line functions; final exponentiation by square-and-multiply
// with (q^12 - 1)/r.  The four pairings of the check share one final exponentiation:
//   e(alpha,beta) e(S,gamma) e(C,delta) == e(A,B)  <=>  FE(ml(alpha,beta) ml(S,gamma) ml(C,delta) ml(-A,B)) == 1.
#include "pipeline.hpp"
#include "pairing_consts.hpp"

namespace zk {

static Fq fq_small(uint32_t v) { return Fq::from_u32(v); }
static Fq2 fq2_mul_xi(const Fq2& a) {   // (9 + i) * a
    Fq n0 = a.c0.dbl().dbl().dbl() + a.c0, n1 = a.c1.dbl().dbl().dbl() + a.c1;
    return Fq2{n0 - a.c1, n1 + a.c0};
}
static Fq2 fq2_conj(const Fq2& a) { return Fq2{a.c0, -a.c1}; }
static Fq2 fq2_from_words(const uint32_t* w) {
    Fq2 x;
    for (int i = 0; i < 8; ++i) { x.c0.l[i] = w[i]; x.c1.l[i] = w[8 + i]; }
    return Fq2::from_canonical(x);
}

struct Fq6 {
    Fq2 a0, a1, a2;
    static Fq6 zero() { return Fq6{Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
    static Fq6 one() { return Fq6{Fq2::one(), Fq2::zero(), Fq2::zero()}; }
    Fq6 operator+(const Fq6& o) const { return Fq6{a0 + o.a0, a1 + o.a1, a2 + o.a2}; }
    Fq6 operator-(const Fq6& o) const { return Fq6{a0 - o.a0, a1 - o.a1, a2 - o.a2}; }
    Fq6 operator-() const { return Fq6{-a0, -a1, -a2}; }
    Fq6 operator*(const Fq6& o) const {
        Fq2 c0 = a0 * o.a0 + fq2_mul_xi(a1 * o.a2 + a2 * o.a1);
        Fq2 c1 = a0 * o.a1 + a1 * o.a0 + fq2_mul_xi(a2 * o.a2);
        Fq2 c2 = a0 * o.a2 + a1 * o.a1 + a2 * o.a0;
        return Fq6{c0, c1, c2};
    }
    Fq6 mul_v() const { return Fq6{fq2_mul_xi(a2), a0, a1}; }
    Fq6 inv() const {
        Fq2 t0 = a0.sqr() - fq2_mul_xi(a1 * a2);
        Fq2 t1 = fq2_mul_xi(a2.sqr()) - a0 * a1;
        Fq2 t2 = a1.sqr() - a0 * a2;
        Fq2 d = (a0 * t0 + fq2_mul_xi(a2 * t1 + a1 * t2)).inv();
        return Fq6{t0 * d, t1 * d, t2 * d};
    }
    bool operator==(const Fq6& o) const { return a0 == o.a0 && a1 == o.a1 && a2 == o.a2; }
};
struct Fq12 {
    Fq6 c0, c1;
    static Fq12 one() { return Fq12{Fq6::one(), Fq6::zero()}; }
    Fq12 operator*(const Fq12& o) const {
        Fq6 t0 = c0 * o.c0, t1 = c1 * o.c1;
        return Fq12{t0 + t1.mul_v(), c0 * o.c1 + c1 * o.c0};
    }
    Fq12 sqr() const { return *this * *this; }
    bool operator==(const Fq12& o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq12 pow_words(const uint32_t* e, int nwords) const {
        Fq12 acc = one();
        bool started = false;
        for (int i = nwords * 32 - 1; i >= 0; --i) {
            if (started) acc = acc.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1) { acc = acc * *this; started = true; }
        }
        return acc;
    }
};

// line through T and Q2 on the twist (tangent when `dbl`), evaluated at P; T <- T + Q2
static Fq12 line_and_add(G2A& T, const G2A& Q2, bool dbl, const G1A& P) {
    Fq2 lam;
    if (dbl) {
        Fq2 x2 = T.x.sqr();
        lam = (x2.dbl() + x2) * T.y.dbl().inv();
    } else {
        lam = (Q2.y - T.y) * (Q2.x - T.x).inv();
    }
    Fq2 x3 = lam.sqr() - T.x - Q2.x;
    Fq2 y3 = lam * (T.x - x3) - T.y;
    // l = yP - lam xP w + (lam xT - yT) v w
    Fq12 l;
    l.c0 = Fq6{Fq2{P.y, Fq::zero()}, Fq2::zero(), Fq2::zero()};
    l.c1 = Fq6{-(lam * Fq2{P.x, Fq::zero()}), lam * T.x - T.y, Fq2::zero()};
    T = G2A{x3, y3};
    return l;
}

static Fq12 miller_loop(const G1A& P, const G2A& Q) {
    if (P.is_inf() || Q.is_inf()) return Fq12::one();
    Fq12 f = Fq12::one();
    G2A T = Q;
    for (int i = ATE_LOOP_BITS - 2; i >= 0; --i) {
        Fq12 l = line_and_add(T, T, true, P);
        f = f.sqr() * l;
        if ((ATE_LOOP[i >> 5] >> (i & 31)) & 1) {
            l = line_and_add(T, Q, false, P);
            f = f * l;
        }
    }
    const Fq2 gx = fq2_from_words(GAMMA_X), gy = fq2_from_words(GAMMA_Y);
    G2A Q1{fq2_conj(Q.x) * gx, fq2_conj(Q.y) * gy};
    G2A Q2{fq2_conj(Q1.x) * gx, -(fq2_conj(Q1.y) * gy)};   // -pi^2(Q)
    f = f * line_and_add(T, Q1, false, P);
    f = f * line_and_add(T, Q2, false, P);
    return f;
}
static Fq12 final_exponentiation(const Fq12& f) { return f.pow_words(FINAL_EXP, FINAL_EXP_WORDS); }

static Fq fq_from_u64x4(const uint64_t* w) {
    Fq x;
    for (int i = 0; i < 4; ++i) { x.l[2 * i] = (uint32_t)w[i]; x.l[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return x;
}
static bool rd_g1(const uint64_t* w, G1A& out) {
    Fq x = fq_from_u64x4(w), y = fq_from_u64x4(w + 4);
    if (!x.raw_in_range() || !y.raw_in_range()) return false;
    out = G1A{Fq::from_canonical(x), Fq::from_canonical(y)};
    if (out.is_inf()) return true;
    return out.y.sqr() == out.x.sqr() * out.x + fq_small(3);
}
static bool rd_g2(const uint64_t* w, G2A& out) {
    Fq2 x{fq_from_u64x4(w), fq_from_u64x4(w + 4)}, y{fq_from_u64x4(w + 8), fq_from_u64x4(w + 12)};
    if (!x.raw_in_range() || !y.raw_in_range()) return false;
    out = G2A{Fq2::from_canonical(x), Fq2::from_canonical(y)};
    if (out.is_inf()) return true;
    Fq2 b2 = Fq2{fq_small(3), Fq::zero()} * Fq2{fq_small(9), fq_small(1)}.inv();   // 3 / xi
    if (!(out.y.sqr() == out.x.sqr() * out.x + b2)) return false;
    // r-torsion: the twist E'(Fq2) has order r * (2q - r) and the cofactor has small factors; the ate Miller loop is
    // bilinear only on the order-r subgroup G2, so a twist point outside it is rejected ([r]Q must be infinity).
    // G1 needs no such test: E(Fq) has prime order r.
    return jac_mul_words(G2J::from_affine(out), FrParams::P).is_inf();
}
static void be_to_words(const uint8_t* be, uint64_t* w) {
    for (int i = 0; i < 4; ++i) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | be[(3 - i) * 8 + b];
        w[i] = v;
    }
}
// decode the canonical 65 / 129 byte blocks of a proof
// The encoding is canonical, so that a proof has exactly one byte string: infinity is tag 0x00 followed by zeros ONLY,
// a finite point is tag 0x04 with coordinates < q that satisfy the curve equation ((0, 0), the in-memory
// image of infinity, is not on either curve and is rejected under tag 0x04).
static bool all_zero(const uint8_t* p, size_t n) {
    uint8_t acc = 0;
    for (size_t i = 0; i < n; ++i) acc |= p[i];
    return acc == 0;
}
static bool dec_g1(const uint8_t* p, G1A& out) {
    if (p[0] == 0) { out = G1A::infinity(); return all_zero(p + 1, 64); }
    if (p[0] != 4) return false;
    uint64_t w[8];
    be_to_words(p + 1, w); be_to_words(p + 33, w + 4);
    return rd_g1(w, out) && !out.is_inf();
}
static bool dec_g2(const uint8_t* p, G2A& out) {
    if (p[0] == 0) { out = G2A::infinity(); return all_zero(p + 1, 128); }
    if (p[0] != 4) return false;
    uint64_t w[16];
    be_to_words(p + 1, w + 4); be_to_words(p + 33, w);          // x.c1 | x.c0
    be_to_words(p + 65, w + 12); be_to_words(p + 97, w + 8);    // y.c1 | y.c0
    return rd_g2(w, out) && !out.is_inf();
}

static void fq12_to_words(const Fq12& f, uint64_t* out) {
    const Fq2* parts[6] = {&f.c0.a0, &f.c0.a1, &f.c0.a2, &f.c1.a0, &f.c1.a1, &f.c1.a2};
    for (int k = 0; k < 6; ++k) {
        Fq c[2] = {parts[k]->c0.to_canonical(), parts[k]->c1.to_canonical()};
        for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 4; ++i) out[(2 * k + h) * 4 + i] = (uint64_t)c[h].l[2 * i] | ((uint64_t)c[h].l[2 * i + 1] << 32);
    }
}

}  // namespace zk

using namespace zk;

extern "C" {

int zk_pairing(const uint64_t g1[ZK_G1_WORDS], const uint64_t g2[ZK_G2_WORDS], uint64_t out[48]) {
    if (!g1 || !g2 || !out) return ZK_ERR_ARG;
    G1A P;
    G2A Q;
    if (!rd_g1(g1, P) || !rd_g2(g2, Q)) return ZK_ERR_RANGE;
    fq12_to_words(final_exponentiation(miller_loop(P, Q)), out);
    return ZK_OK;
}

int zk_verify(zk_ctx* ctx, const zk_crs* crs, const uint64_t* inputs, size_t n_inputs, const uint8_t proof[ZK_PROOF_BYTES], int* ok) {
    if (!ctx || !crs || !proof || !ok || (n_inputs && !inputs)) return ZK_ERR_ARG;
    *ok = 0;
    return guarded(ctx, [&] {
        const size_t l = crs->input;
        // host copies of the handful of CRS points verify reads
        std::vector<uint64_t> sg((l + 1) * 8), a1(8), b2(16), g2(16), d2(16);
        zk_crs_out o{};
        o.alpha_g1 = a1.data(); o.sum_gamma_g1 = sg.data(); o.beta_g2 = b2.data(); o.gamma_g2 = g2.data(); o.delta_g2 = d2.data();
        crs_download(ctx, *crs, o);
        G1A alpha, A, C;
        G2A beta, gamma, delta, B;
        ZK_REQUIRE(rd_g1(a1.data(), alpha) && rd_g2(b2.data(), beta) && rd_g2(g2.data(), gamma) && rd_g2(d2.data(), delta), ZK_ERR_ARG, "verify: CRS point not on the curve or outside G2");
        if (!dec_g1(proof, A) || !dec_g2(proof + 65, B) || !dec_g1(proof + 194, C)) return;   // malformed / off-curve proof: rejected
        // sum_term = sum_{i<=l} (1, inputs...)_i * sum_gamma_i  (zip truncates, mod.rs:308-314)
        G1J sum = G1J::infinity();
        for (size_t i = 0; i <= l && i < n_inputs + 1; ++i) {
            G1A g;
            ZK_REQUIRE(rd_g1(sg.data() + 8 * i, g), ZK_ERR_ARG, "verify: CRS point not on the curve");
            Fr k;
            if (i == 0) { k = Fr::zero(); k.l[0] = 1; }
            else {
                k = Fr::zero();
                for (int w = 0; w < 4; ++w) { k.l[2 * w] = (uint32_t)inputs[4 * (i - 1) + w]; k.l[2 * w + 1] = (uint32_t)(inputs[4 * (i - 1) + w] >> 32); }
                ZK_REQUIRE(k.raw_in_range(), ZK_ERR_RANGE, "verify: input >= r");
            }
            sum = jac_add(sum, jac_mul_words(G1J::from_affine(g), k.l));
        }
        G1A S = jac_to_affine(sum);
        Fq12 f = miller_loop(alpha, beta) * miller_loop(S, gamma) * miller_loop(C, delta) * miller_loop(A.neg(), B);
        *ok = final_exponentiation(f) == Fq12::one() ? 1 : 0;
    });
}

}  // extern "C"
