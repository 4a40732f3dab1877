// oracle/groth16.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// Faithful restatement of the reference protocol layer, generic over an "engine" E that plays
// the role of `T: EllipticEncryptable + Field` (/root/reference/src/groth16/mod.rs:23-56):
//   QAP<P>                        /root/reference/src/groth16/mod.rs:60-67
//   QAP::from(RootRepresentation) /root/reference/src/groth16/mod.rs:69-102, src/groth16/fr.rs:140-173
//   SigmaG1 / SigmaG2 / Proof     /root/reference/src/groth16/mod.rs:105-128
//   setup                         /root/reference/src/groth16/mod.rs:134-197
//   prove                         /root/reference/src/groth16/mod.rs:213-296
//   verify                        /root/reference/src/groth16/mod.rs:299-320
// The reference draws (alpha,beta,gamma,delta,x) and (r,s) from thread_rng inside setup/prove
// (mod.rs:139-145,231); here they are injected so results are reproducible.
//
// Engine concept:
//   typedef T, G1, G2, GT;
//   static G1 encrypt_g1(T); static G2 encrypt_g2(T);
//   static G1 exp_g1(T, G1); static G2 exp_g2(T, G2);      // exp_encrypted_g1/g2
//   static GT pairing(G1, G2);  static GT gt_add(GT, GT);   // GtLocal "+" is Gt multiplication
//   static G1 g1_zero(); static G2 g2_zero();               // Sum seeds
//   G1: + and -, G2: +
#pragma once
#include "poly.hpp"
#include "z251.hpp"
#include "bn254.hpp"

namespace orc {

// circuit::dummy_rep::DummyRep (/root/reference/src/groth16/circuit/dummy_rep.rs:6-13)
template <class F>
struct DummyRep {
    std::vector<std::vector<std::pair<F, F>>> u, v, w;  // per wire: (root, value)
    std::vector<F> roots;
    size_t input = 0;
    bool operator==(const DummyRep& o) const {
        return u == o.u && v == o.v && w == o.w && roots == o.roots && input == o.input;
    }
};

template <class T>
struct QAP {
    std::vector<Coeffs<T>> u, v, w;
    Coeffs<T> t;
    size_t input = 0;
    size_t degree = 0;
};

template <class T>
QAP<T> qap_from_root_rep(const DummyRep<T>& rr) {
    QAP<T> q;
    for (const auto& pts : rr.u) q.u.push_back(poly_from_points(rr.roots, pts));
    for (const auto& pts : rr.v) q.v.push_back(poly_from_points(rr.roots, pts));
    for (const auto& pts : rr.w) q.w.push_back(poly_from_points(rr.roots, pts));
    if (q.u.size() != q.v.size() || q.u.size() != q.w.size()) throw std::logic_error("assert_eq!(u.len(), v.len())");
    q.t = root_poly(rr.roots);
    q.input = rr.input;
    q.degree = orc::degree(q.t);
    return q;
}

template <class G1T>
struct SigmaG1 {
    G1T alpha, beta, delta;
    std::vector<G1T> xi, sum_gamma, sum_delta, xi_t;
};
template <class G2T>
struct SigmaG2 {
    G2T beta, gamma, delta;
    std::vector<G2T> xi;
};
template <class G1T, class G2T>
struct Proof {
    G1T a;
    G2T b;
    G1T c;
};

template <class T>
struct Trapdoor {
    T alpha, beta, gamma, delta, x;
};

// mod.rs:134-197
template <class E>
std::pair<SigmaG1<typename E::G1>, SigmaG2<typename E::G2>> setup_with_trapdoor(
    const QAP<typename E::T>& qap, const Trapdoor<typename E::T>& td) {
    typedef typename E::T T;
    const T &alpha = td.alpha, &beta = td.beta, &gamma = td.gamma, &delta = td.delta, &x = td.x;
    Coeffs<T> xi = powers(x, qap.degree);
    SigmaG1<typename E::G1> s1;
    SigmaG2<typename E::G2> s2;
    size_t m = std::min(qap.u.size(), std::min(qap.v.size(), qap.w.size()));
    for (size_t i = 0; i < m; ++i) {
        T comb = beta * evaluate(qap.u[i], x) + alpha * evaluate(qap.v[i], x) + evaluate(qap.w[i], x);
        if (i < qap.input + 1) s1.sum_gamma.push_back(E::encrypt_g1(comb / gamma));
        else s1.sum_delta.push_back(E::encrypt_g1(comb / delta));
    }
    if (!xi.empty()) {
        for (size_t i = 0; i + 1 < xi.size(); ++i)
            s1.xi_t.push_back(E::encrypt_g1((xi[i] * evaluate(qap.t, x)) / delta));
    }
    s1.alpha = E::encrypt_g1(alpha);
    s1.beta = E::encrypt_g1(beta);
    s1.delta = E::encrypt_g1(delta);
    for (const auto& e : xi) s1.xi.push_back(E::encrypt_g1(e));
    s2.beta = E::encrypt_g2(beta);
    s2.gamma = E::encrypt_g2(gamma);
    s2.delta = E::encrypt_g2(delta);
    for (const auto& e : xi) s2.xi.push_back(E::encrypt_g2(e));
    return {s1, s2};
}

// The weighted sums of mod.rs:233-253: sum_i qap.u[i] * weights[i] (zip truncates)
template <class T>
Coeffs<T> weighted_sum(const std::vector<Coeffs<T>>& polys, const std::vector<T>& weights) {
    Coeffs<T> acc{T::from_usize(0)};
    size_t m = std::min(polys.size(), weights.size());
    for (size_t i = 0; i < m; ++i) acc = poly_add(acc, poly_scale(polys[i], weights[i]));
    return acc;
}

// mod.rs:213-296
template <class E>
Proof<typename E::G1, typename E::G2> prove_with_rs(
    const QAP<typename E::T>& qap, const SigmaG1<typename E::G1>& s1, const SigmaG2<typename E::G2>& s2,
    const std::vector<typename E::T>& weights, const typename E::T& r, const typename E::T& s) {
    typedef typename E::T T;
    typedef typename E::G1 G1T;
    typedef typename E::G2 G2T;
    Coeffs<T> u_sum = weighted_sum(qap.u, weights);
    Coeffs<T> v_sum = weighted_sum(qap.v, weights);
    Coeffs<T> w_sum = weighted_sum(qap.w, weights);

    G1T a_g1 = E::g1_zero(), b_g1 = E::g1_zero();
    G2T b_g2 = E::g2_zero();
    for (size_t i = 0; i < std::min(u_sum.size(), s1.xi.size()); ++i) a_g1 = a_g1 + E::exp_g1(u_sum[i], s1.xi[i]);
    for (size_t i = 0; i < std::min(v_sum.size(), s1.xi.size()); ++i) b_g1 = b_g1 + E::exp_g1(v_sum[i], s1.xi[i]);
    for (size_t i = 0; i < std::min(v_sum.size(), s2.xi.size()); ++i) b_g2 = b_g2 + E::exp_g2(v_sum[i], s2.xi[i]);

    G1T a = a_g1 + s1.alpha + E::exp_g1(r, s1.delta);
    G2T b = b_g2 + s2.beta + E::exp_g2(s, s2.delta);

    Coeffs<T> h = poly_div(poly_sub(poly_mul(u_sum, v_sum), w_sum), qap.t);

    G1T c_h = E::g1_zero();
    for (size_t i = 0; i < std::min(h.size(), s1.xi_t.size()); ++i) c_h = c_h + E::exp_g1(h[i], s1.xi_t[i]);
    G1T c_l = E::g1_zero();
    for (size_t i = qap.input + 1, k = 0; i < weights.size() && k < s1.sum_delta.size(); ++i, ++k)
        c_l = c_l + E::exp_g1(weights[i], s1.sum_delta[k]);
    G1T c = c_h + c_l + E::exp_g1(s, a) + E::exp_g1(r, s1.beta + b_g1 + E::exp_g1(s, s1.delta)) -
            E::exp_g1(r * s, s1.delta);
    return Proof<G1T, G2T>{a, b, c};
}

// mod.rs:299-320
template <class E>
bool verify(const SigmaG1<typename E::G1>& s1, const SigmaG2<typename E::G2>& s2,
            const std::vector<typename E::T>& inputs, const Proof<typename E::G1, typename E::G2>& proof) {
    typedef typename E::T T;
    typename E::G1 sum_term = E::g1_zero();
    for (size_t i = 0; i < s1.sum_gamma.size() && i < inputs.size() + 1; ++i) {
        T a = i == 0 ? T::one() : inputs[i - 1];
        sum_term = sum_term + E::exp_g1(a, s1.sum_gamma[i]);
    }
    return E::gt_add(E::gt_add(E::pairing(s1.alpha, s2.beta), E::pairing(sum_term, s2.gamma)),
                     E::pairing(proof.c, s2.delta)) == E::pairing(proof.a, proof.b);
}

// ---- Engines ---------------------------------------------------------------------------
// Z251 as its own pairing group: /root/reference/src/groth16/mod.rs:339-359
struct Z251Engine {
    typedef Z251 T;
    typedef Z251 G1;
    typedef Z251 G2;
    typedef Z251 GT;
    static G1 encrypt_g1(T a) { return a * Z251::from_usize(69); }
    static G2 encrypt_g2(T a) { return a * Z251::from_usize(69); }
    static G1 exp_g1(T a, G1 g) { return a * g; }
    static G2 exp_g2(T a, G2 g) { return a * g; }
    static GT pairing(G1 a, G2 b) { return a * b; }
    static GT gt_add(GT a, GT b) { return a + b; }
    static G1 g1_zero() { return Z251::from_usize(0); }
    static G2 g2_zero() { return Z251::from_usize(0); }
};

// FrLocal over bn: /root/reference/src/groth16/fr.rs:101-123 (pairing is provided by pairing.hpp)
struct BnEngine {
    typedef Fr T;
    typedef orc::G1 G1;
    typedef orc::G2 G2;
    static G1 encrypt_g1(const T& a) { return enc_base_g1().mul(a); }
    static G2 encrypt_g2(const T& a) { return enc_base_g2().mul(a); }
    static G1 exp_g1(const T& a, const G1& g) { return g.mul(a); }
    static G2 exp_g2(const T& a, const G2& g) { return g.mul(a); }
    static G1 g1_zero() { return G1::zero(); }
    static G2 g2_zero() { return G2::zero(); }
};

// Pairing-free validity oracle: the proof an honest prover must output given
// the trapdoor, as discrete logs w.r.t. the encryption bases.  Uses only Fr arithmetic and
// three scalar multiplications; independent of MSM algorithm, NTT and summation order.
// `hx_tx` = h(x) * t(x) where h is the reference's QUOTIENT (remainder dropped).
inline Proof<G1, G2> trapdoor_proof(const Trapdoor<Fr>& td, size_t input,
                                    const std::vector<Fr>& ux, const std::vector<Fr>& vx, const std::vector<Fr>& wx,
                                    const Fr& hx_tx, const std::vector<Fr>& weights, const Fr& r, const Fr& s) {
    Fr U = Fr::zero(), V = Fr::zero(), L = Fr::zero();
    size_t m = std::min(weights.size(), ux.size());
    for (size_t i = 0; i < m; ++i) {
        U = U + weights[i] * ux[i];
        V = V + weights[i] * vx[i];
        if (i > input) L = L + weights[i] * (td.beta * ux[i] + td.alpha * vx[i] + wx[i]);
    }
    Fr a_log = td.alpha + U + r * td.delta;
    Fr b_log = td.beta + V + s * td.delta;
    Fr c_log = (L + hx_tx) / td.delta + s * a_log + r * b_log - r * s * td.delta;
    return Proof<G1, G2>{enc_base_g1().mul(a_log), enc_base_g2().mul(b_log), enc_base_g1().mul(c_log)};
}

}  // namespace orc
