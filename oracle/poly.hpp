// oracle/poly.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// Faithful restatement of the reference's generic polynomial algebra, generic over a field
// type T exactly as the Rust is (instantiated for Z251 and Fr):
//   Polynomial::{coefficients,degree,evaluate,remove_leading_zeros}  /root/reference/src/field/mod.rs:231-356
//   polynomial_division                                              /root/reference/src/field/mod.rs:428-469
//   powers, dft, idft                                                /root/reference/src/field/mod.rs:493-537
//   CoefficientPoly Add/Neg/Sub/Sum/Mul/Mul<T>/Div                   /root/reference/src/groth16/coefficient_poly.rs:24-157
//   From<(roots,points)>, lagrange_basis, root_poly                  /root/reference/src/groth16/coefficient_poly.rs:159-200
// Pinned by the reference's own known-answer tests (all over Z251), see tests/test_oracle_kats.py.
//
// Field concept: T::zero(), T::one(), T::from_usize(n), + - * / unary-, ==, T::inv()
// ("/" and inv() throw std::domain_error on zero, mirroring the reference's panics).
#pragma once
#include <algorithm>
#include <cstddef>
#include <stdexcept>
#include <utility>
#include <vector>

namespace orc {

template <class T>
using Coeffs = std::vector<T>;

// field/mod.rs:291-297: highest exponent ignoring high zero coefficients; 0 for empty / all-zero.
template <class T>
size_t degree(const Coeffs<T>& c) {
    size_t n = c.size();
    while (n > 0 && c[n - 1] == T::zero()) --n;
    return n == 0 ? 0 : n - 1;
}

// field/mod.rs:344-355
template <class T>
void remove_leading_zeros(Coeffs<T>& c) {
    size_t n = c.size();
    while (n > 0 && c[n - 1] == T::zero()) --n;
    c.resize(n, T::zero());
}

// field/mod.rs:338-343 (Horner)
template <class T>
T evaluate(const Coeffs<T>& c, const T& x) {
    T acc = T::zero();
    for (size_t i = c.size(); i-- > 0;) acc = (acc * x) + c[i];
    return acc;
}

// field/mod.rs:493-504
template <class T>
Coeffs<T> powers(const T& x, size_t n) {
    Coeffs<T> out;
    out.reserve(n);
    T s = T::one();
    for (size_t i = 0; i < n; ++i) { out.push_back(s); s = s * x; }
    return out;
}

// field/mod.rs:508-520: out[k] = sum_j seq[j] * (root^k)^j, natural order in and out.
template <class T>
Coeffs<T> dft(const Coeffs<T>& seq, const T& root) {
    Coeffs<T> out;
    Coeffs<T> ri = powers(root, seq.size());
    for (size_t k = 0; k < seq.size(); ++k) {
        T acc = T::zero();
        T r = T::one();
        for (size_t j = 0; j < seq.size(); ++j) { acc = acc + seq[j] * r; r = r * ri[k]; }
        out.push_back(acc);
    }
    return out;
}

// field/mod.rs:524-537
template <class T>
Coeffs<T> idft(const Coeffs<T>& seq, const T& root) {
    Coeffs<T> out = dft(seq, root.inv());
    T ninv = T::from_usize(seq.size()).inv();
    for (auto& v : out) v = v * ninv;
    return out;
}

// coefficient_poly.rs:24-49: length = max(len a, len b); no stripping of high zeros.
template <class T>
Coeffs<T> poly_add(const Coeffs<T>& a, const Coeffs<T>& b) {
    const Coeffs<T>& s = a.size() < b.size() ? a : b;
    const Coeffs<T>& l = a.size() < b.size() ? b : a;
    Coeffs<T> out;
    out.reserve(l.size());
    for (size_t i = 0; i < l.size(); ++i) out.push_back((i < s.size() ? s[i] : T::from_usize(0)) + l[i]);
    return out;
}

// coefficient_poly.rs:51-62
template <class T>
Coeffs<T> poly_neg(Coeffs<T> a) {
    for (auto& c : a) c = -c;
    return a;
}

// coefficient_poly.rs:64-73
template <class T>
Coeffs<T> poly_sub(const Coeffs<T>& a, const Coeffs<T>& b) { return poly_add(a, poly_neg(b)); }

// coefficient_poly.rs:75-91: fold seeded with [0]
template <class T>
Coeffs<T> poly_sum(const std::vector<Coeffs<T>>& polys) {
    Coeffs<T> acc{T::from_usize(0)};
    for (const auto& p : polys) acc = poly_add(acc, p);
    return acc;
}

// coefficient_poly.rs:132-146
template <class T>
Coeffs<T> poly_scale(Coeffs<T> a, const T& s) {
    for (auto& c : a) c = c * s;
    return a;
}

// coefficient_poly.rs:93-130: schoolbook product after stripping high zeros; output has
// deg(a)+deg(b)+1 coefficients (an empty/zero operand yields that many zeros, never []).
template <class T>
Coeffs<T> poly_mul(Coeffs<T> a, Coeffs<T> b) {
    remove_leading_zeros(a);
    remove_leading_zeros(b);
    size_t da = degree(a), db = degree(b);
    size_t d = da + db + 1;
    Coeffs<T> out(d, T::from_usize(0));
    for (size_t i = 0; i < d; ++i) {
        // pairs (k, i-k), k over valid indices of a (ascending index of b as the reference zips)
        T acc = T::from_usize(0);
        size_t jlo = i > da ? i - da : 0;
        for (size_t j = jlo; j <= i && j < b.size(); ++j) {
            size_t k = i - j;
            if (k < a.size()) acc = acc + a[k] * b[j];
        }
        out[i] = acc;
    }
    return out;
}

// field/mod.rs:428-469.  Returns (q, r).  Throws on an all-zero divisor ("Dividend must be
// non-zero" panic, :440); returns ([0],[0]) when deg(divisor) > deg(poly) (:443-445).
// One field division per quotient term (:456), as the reference does.
template <class T>
std::pair<Coeffs<T>, Coeffs<T>> polynomial_division(Coeffs<T> poly, Coeffs<T> dividend) {
    {
        bool all_zero = true;
        for (const auto& c : dividend) if (!(c == T::zero())) { all_zero = false; break; }
        if (all_zero) throw std::domain_error("Dividend must be non-zero");
    }
    if (degree(dividend) > degree(poly)) return {Coeffs<T>{T::zero()}, Coeffs<T>{T::zero()}};
    remove_leading_zeros(poly);
    remove_leading_zeros(dividend);
    size_t d = degree(dividend);
    Coeffs<T> q(degree(poly) + 1 - d, T::zero());
    Coeffs<T> r = poly;
    T c = dividend[d];
    while (r.size() != 0 && degree(r) >= d) {
        size_t dr = degree(r);
        T s = r[dr] / c;
        q[dr - d] = s;
        for (size_t k = 0; k <= d; ++k) r[dr - d + k] = r[dr - d + k] - dividend[k] * s;
        remove_leading_zeros(r);
    }
    return {q, r};
}

// coefficient_poly.rs:148-157: Div keeps only the quotient
template <class T>
Coeffs<T> poly_div(const Coeffs<T>& a, const Coeffs<T>& b) { return polynomial_division(a, b).first; }

// coefficient_poly.rs:173-190
template <class T>
Coeffs<T> lagrange_basis(const Coeffs<T>& roots, const T& x) {
    Coeffs<T> acc{T::from_usize(1)};
    for (const auto& m : roots) {
        if (m == x) continue;
        Coeffs<T> lin{-m, T::from_usize(1)};
        acc = poly_mul(poly_scale(lin, T::from_usize(1) / (x - m)), acc);
    }
    return acc;
}

// coefficient_poly.rs:159-171
template <class T>
Coeffs<T> poly_from_points(const Coeffs<T>& roots, const std::vector<std::pair<T, T>>& points) {
    std::vector<Coeffs<T>> terms;
    for (const auto& pt : points) terms.push_back(poly_scale(lagrange_basis(roots, pt.first), pt.second));
    return poly_sum(terms);
}

// coefficient_poly.rs:192-200
template <class T>
Coeffs<T> root_poly(const Coeffs<T>& roots) {
    Coeffs<T> acc{T::from_usize(1)};
    for (const auto& r : roots) acc = poly_mul(acc, Coeffs<T>{-r, T::from_usize(1)});
    return acc;
}

}  // namespace orc
