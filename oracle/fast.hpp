// oracle/fast.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// Same-result, asymptotically fast CPU formulations of the reference's hot path, used (a) to
// extend parity checks beyond the sizes the faithful restatement (groth16.hpp) can reach
// (the reference is O(m*n)+O(n^2)) and (b) as the "same-algorithm CPU" baseline
// B2 of BASELINE.md.  Every function here is pinned against the faithful path at small n in
// tests/test_oracle_fast.py; nothing here is an independent source of truth.
//
//   fr_ntt            == field::dft / idft semantics  (/root/reference/src/field/mod.rs:508-537)
//   msm_pippenger     == sum_i exp_encrypted(a_i, P_i) (/root/reference/src/groth16/mod.rs:255-272)
//   SparseQap::*      == QAP::from(root_rep) + setup/prove with roots = omega^j
//                        (/root/reference/src/groth16/fr.rs:140-173, mod.rs:134-296)
#pragma once
#include <atomic>
#include <functional>
#include <thread>
#include <algorithm>
#include "groth16.hpp"

namespace orc {

static inline uint32_t bitrev32(uint32_t x, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

// In-place radix-2 NTT, natural order in and out: out[k] = sum_j in[j] * w^(jk), w = root of
// unity of order 2^log_n (inverse: w^-1 and scaling by n^-1, as field::idft).
// tasks 0..count-1 pulled from an atomic counter by `threads` workers (threads <= 1: inline)
inline void parallel_for(size_t count, unsigned threads, const std::function<void(size_t)>& fn) {
    if (threads <= 1 || count <= 1) { for (size_t i = 0; i < count; ++i) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    unsigned nt = (unsigned)std::min<size_t>(threads, count);
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < count;) fn(i); });
    for (auto& th : pool) th.join();
}

inline void fr_ntt(std::vector<Fr>& a, unsigned log_n, bool inverse, unsigned threads = 1) {
    size_t n = (size_t)1 << log_n;
    Fr w = fr_root_of_unity((int)log_n);
    if (inverse) w = w.inv();
    for (size_t i = 0; i < n; ++i) { size_t j = bitrev32((uint32_t)i, log_n); if (i < j) std::swap(a[i], a[j]); }
    for (unsigned s = 1; s <= log_n; ++s) {
        size_t half = (size_t)1 << (s - 1);
        Fr wm = w;
        for (unsigned k = s; k < log_n; ++k) wm = wm.sqr();
        std::vector<Fr> tw(half);
        tw[0] = Fr::one();
        for (size_t k = 1; k < half; ++k) tw[k] = tw[k - 1] * wm;
        // n/2 butterflies per stage, split into contiguous blocks of butterfly indices
        const size_t total = n >> 1, blocks = threads > 1 ? std::min<size_t>(total, (size_t)threads * 4) : 1;
        parallel_for(blocks, threads, [&](size_t blk) {
            for (size_t b = total * blk / blocks, e = total * (blk + 1) / blocks; b < e; ++b) {
                size_t k = b & (half - 1), base = (b >> (s - 1)) << s;
                Fr t = tw[k] * a[base + k + half];
                Fr u = a[base + k];
                a[base + k] = u + t;
                a[base + k + half] = u - t;
            }
        });
    }
    if (inverse) {
        Fr ninv = Fr::from_u64(n).inv();
        for (auto& v : a) v = v * ninv;
    }
}

// Mixed addition Jacobian += affine (complete: handles inf, doubling, inverse)
template <class F>
Jac<F> madd(const Jac<F>& p, const Affine<F>& q) {
    if (q.inf) return p;
    if (p.is_zero()) return Jac<F>::from_affine(q);
    F Z1Z1 = p.Z.sqr();
    F U2 = q.x * Z1Z1;
    F S2 = q.y * p.Z * Z1Z1;
    if (U2 == p.X) {
        if (S2 == p.Y) return p.dbl();
        return Jac<F>::zero();
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

// Pippenger bucket MSM, unsigned c-bit windows.
template <class F>
Jac<F> msm_pippenger(const std::vector<Affine<F>>& pts, const std::vector<U256>& sc, unsigned c, unsigned threads = 1) {
    size_t n = std::min(pts.size(), sc.size());
    unsigned windows = (256 + c - 1) / c;
    // windows are independent (one task each); the Horner combination over windows is serial
    std::vector<Jac<F>> wsum(windows, Jac<F>::zero());
    parallel_for(windows, threads, [&](size_t w) {
        std::vector<Jac<F>> buckets((size_t)1 << c, Jac<F>::zero());
        unsigned lo = (unsigned)w * c;
        for (size_t i = 0; i < n; ++i) {
            uint32_t d = 0;
            for (unsigned k = 0; k < c && lo + k < 256; ++k) d |= (uint32_t)sc[i].bit((int)(lo + k)) << k;
            if (d) buckets[d] = madd(buckets[d], pts[i]);
        }
        Jac<F> run = Jac<F>::zero(), acc = Jac<F>::zero();
        for (size_t b = buckets.size() - 1; b >= 1; --b) { run = run + buckets[b]; acc = acc + run; }
        wsum[w] = acc;
    });
    Jac<F> total = Jac<F>::zero();
    for (int w = (int)windows - 1; w >= 0; --w) {
        for (unsigned k = 0; k < c; ++k) total = total.dbl();
        total = total + wsum[w];
    }
    return total;
}

// Fixed-base scalar multiplication table (8-bit windows) for fast CRS generation.
template <class F>
struct FixedBase {
    std::vector<Affine<F>> tab;  // [32][256]
    explicit FixedBase(const Jac<F>& base) {
        tab.resize(32 * 256);
        Jac<F> wbase = base;
        for (int w = 0; w < 32; ++w) {
            Jac<F> acc = Jac<F>::zero();
            for (int d = 0; d < 256; ++d) { tab[w * 256 + d] = acc.to_affine(); acc = acc + wbase; }
            wbase = acc;  // 256 * wbase
        }
    }
    Jac<F> mul(const Fr& k) const {
        U256 e = k.to_u256();
        Jac<F> acc = Jac<F>::zero();
        for (int w = 0; w < 32; ++w) {
            unsigned d = (unsigned)(e.l[w / 8] >> (8 * (w % 8))) & 0xff;
            if (d) acc = madd(acc, tab[w * 256 + d]);
        }
        return acc;
    }
};

// Montgomery batch inversion (zeros stay zero)
inline void batch_inverse(std::vector<Fr>& v) {
    std::vector<Fr> pre(v.size());
    Fr acc = Fr::one();
    for (size_t i = 0; i < v.size(); ++i) { pre[i] = acc; if (!v[i].is_zero()) acc = acc * v[i]; }
    Fr inv = acc.inv();
    for (size_t i = v.size(); i-- > 0;) {
        if (v[i].is_zero()) continue;
        Fr t = inv * pre[i];
        inv = inv * v[i];
        v[i] = t;
    }
}

// Sparse evaluation-form QAP on the domain roots[j] = omega^j, n = 2^log_n gates.
// CSR by WIRE for each of u, v, w: entries (gate index, value) -- the DummyRep rows with the
// root replaced by its index.
struct SparseMat {
    std::vector<size_t> ptr;      // m+1
    std::vector<uint32_t> gate;   // nnz
    std::vector<Fr> val;          // nnz
};
struct SparseQap {
    unsigned log_n = 0;
    size_t m = 0, input = 0;
    SparseMat u, v, w;
    size_t n_ap = 0;   // > 0: the domain is the integers 1..n_ap (ASTParser's roots, circuit/mod.rs:517) instead of the roots of unity
    size_t n() const { return n_ap ? n_ap : (size_t)1 << log_n; }

    // evaluation vector over the domain: E[j] = sum_i weights[i] * M_i(omega^j)
    std::vector<Fr> eval_vec(const SparseMat& M, const std::vector<Fr>& weights) const {
        std::vector<Fr> e(n(), Fr::zero());
        size_t mm = std::min(m, weights.size());
        for (size_t i = 0; i < mm; ++i)
            for (size_t k = M.ptr[i]; k < M.ptr[i + 1]; ++k) e[M.gate[k]] = e[M.gate[k]] + weights[i] * M.val[k];
        return e;
    }
    // Lagrange basis values L_j(x) for the domain omega^j
    std::vector<Fr> lagrange_at(const Fr& x) const {
        size_t N = n();
        Fr w = fr_root_of_unity((int)log_n);
        std::vector<Fr> wj = powers(w, N), den(N);
        for (size_t j = 0; j < N; ++j) den[j] = x - wj[j];
        Fr xn = x;
        for (unsigned k = 0; k < log_n; ++k) xn = xn.sqr();
        Fr c = (xn - Fr::one()) * Fr::from_u64(N).inv();
        std::vector<Fr> L(N);
        bool hit = false;
        for (size_t j = 0; j < N; ++j) if (den[j].is_zero()) { hit = true; for (auto& l : L) l = Fr::zero(); L[j] = Fr::one(); }
        if (hit) return L;
        batch_inverse(den);
        for (size_t j = 0; j < N; ++j) L[j] = c * wj[j] * den[j];
        return L;
    }
    // Lagrange basis values L_k(x), k = 1..n, for the domain {1, .., n}: L_k(x) = t(x) w_k / (x - k), t(x) = prod (x - j),
    // w_k = 1 / prod_{j != k} (k - j) = (-1)^(n-k) / ((k-1)! (n-k)!)   (coefficient_poly.rs:173-190 evaluated at x)
    std::vector<Fr> lagrange_at_integers(const Fr& x, Fr* t_at_x = nullptr) const {
        const size_t N = n();
        std::vector<Fr> fact(N + 1), den(N), L(N);
        fact[0] = Fr::one();
        for (size_t j = 1; j <= N; ++j) fact[j] = fact[j - 1] * Fr::from_u64(j);
        Fr tx = Fr::one();
        for (size_t k = 1; k <= N; ++k) { den[k - 1] = x - Fr::from_u64(k); tx = tx * den[k - 1]; }
        if (t_at_x) *t_at_x = tx;
        for (size_t k = 1; k <= N; ++k) {
            if (den[k - 1].is_zero()) { for (auto& l : L) l = Fr::zero(); L[k - 1] = Fr::one(); return L; }
            den[k - 1] = den[k - 1] * fact[k - 1] * fact[N - k];
            if ((N - k) & 1) den[k - 1] = -den[k - 1];
        }
        batch_inverse(den);
        for (size_t k = 0; k < N; ++k) L[k] = tx * den[k];
        return L;
    }
    std::vector<Fr> wire_evals(const SparseMat& M, const std::vector<Fr>& L) const {
        std::vector<Fr> out(m, Fr::zero());
        for (size_t i = 0; i < m; ++i)
            for (size_t k = M.ptr[i]; k < M.ptr[i + 1]; ++k) out[i] = out[i] + M.val[k] * L[M.gate[k]];
        return out;
    }
};

struct BnCrs {
    SigmaG1<G1> s1;
    SigmaG2<G2> s2;
};

// setup for a SparseQap (same CRS as the faithful setup on QAP::from(root_rep) with roots omega^j)
inline BnCrs fast_setup(const SparseQap& q, const Trapdoor<Fr>& td) {
    static FixedBase<Fq> fb1(enc_base_g1());
    static FixedBase<Fq2> fb2(enc_base_g2());
    size_t n = q.n();
    std::vector<Fr> L = q.n_ap ? q.lagrange_at_integers(td.x) : q.lagrange_at(td.x);
    std::vector<Fr> ux = q.wire_evals(q.u, L), vx = q.wire_evals(q.v, L), wx = q.wire_evals(q.w, L);
    Fr tx = td.x;
    for (unsigned k = 0; k < q.log_n; ++k) tx = tx.sqr();
    tx = tx - Fr::one();
    Fr ginv = td.gamma.inv(), dinv = td.delta.inv();
    BnCrs crs;
    std::vector<Fr> xi = powers(td.x, n);
    for (size_t i = 0; i < q.m; ++i) {
        Fr comb = td.beta * ux[i] + td.alpha * vx[i] + wx[i];
        if (i < q.input + 1) crs.s1.sum_gamma.push_back(fb1.mul(comb * ginv));
        else crs.s1.sum_delta.push_back(fb1.mul(comb * dinv));
    }
    for (size_t i = 0; i + 1 < n; ++i) crs.s1.xi_t.push_back(fb1.mul(xi[i] * tx * dinv));
    crs.s1.alpha = fb1.mul(td.alpha); crs.s1.beta = fb1.mul(td.beta); crs.s1.delta = fb1.mul(td.delta);
    for (size_t i = 0; i < n; ++i) { crs.s1.xi.push_back(fb1.mul(xi[i])); crs.s2.xi.push_back(fb2.mul(xi[i])); }
    crs.s2.beta = fb2.mul(td.beta); crs.s2.gamma = fb2.mul(td.gamma); crs.s2.delta = fb2.mul(td.delta);
    return crs;
}

template <class F>
std::vector<Affine<F>> to_affine_vec(const std::vector<Jac<F>>& v, size_t count) {
    std::vector<Affine<F>> out;
    for (size_t i = 0; i < std::min(count, v.size()); ++i)
        out.push_back(!v[i].is_zero() && v[i].Z == F::one() ? Affine<F>{v[i].X, v[i].Y, false} : v[i].to_affine());   // Z = 1: no inversion
    return out;
}

struct FastProveStats { double t_eval = 0, t_ntt = 0, t_msm = 0; };

// One window of a Pippenger MSM: sum_b b * (sum of the points whose c-bit digit at bit `lo` is b), unsigned digits.
template <class F>
Jac<F> msm_window(const std::vector<Affine<F>>& pts, const std::vector<U256>& sc, unsigned c, unsigned lo) {
    const size_t n = std::min(pts.size(), sc.size());
    std::vector<Jac<F>> buckets((size_t)1 << c, Jac<F>::zero());
    const uint64_t mask = ((uint64_t)1 << c) - 1;
    for (size_t i = 0; i < n; ++i) {
        const unsigned word = lo >> 6, off = lo & 63;
        uint64_t d = word < 4 ? sc[i].l[word] >> off : 0;
        if (off + c > 64 && word + 1 < 4) d |= sc[i].l[word + 1] << (64 - off);
        d &= mask;
        if (d) buckets[d] = madd(buckets[d], pts[i]);
    }
    Jac<F> run = Jac<F>::zero(), acc = Jac<F>::zero();
    for (size_t b = buckets.size() - 1; b >= 1; --b) { run = run + buckets[b]; acc = acc + run; }
    return acc;
}
template <class F>
Jac<F> msm_combine(const std::vector<Jac<F>>& wsum, unsigned c) {
    Jac<F> total = Jac<F>::zero();
    for (int w = (int)wsum.size() - 1; w >= 0; --w) {
        for (unsigned k = 0; k < c; ++k) total = total.dbl();
        total = total + wsum[w];
    }
    return total;
}

// The CRS as prove() reads it, in affine form (what the CPU path needs for mixed additions).  Built once per CRS, outside any
// timed region: the reference, too, holds its CRS ready-made (it clones vectors of points, groth16/mod.rs:282,288).
struct FastCrs {
    std::vector<Affine<Fq>> xi1, xit, sdl;
    std::vector<Affine<Fq2>> xi2;
    const BnCrs* crs = nullptr;
    FastCrs(const BnCrs& c, size_t n) : crs(&c) {
        xi1 = to_affine_vec(c.s1.xi, n); xi2 = to_affine_vec(c.s2.xi, n);
        xit = to_affine_vec(c.s1.xi_t, n); sdl = to_affine_vec(c.s1.sum_delta, (size_t)-1);
    }
};

// prove for a SparseQap: NTT for interpolation / product, exact division by t = x^n - 1,
// Pippenger for the five inner products.  Same group elements as prove_with_rs.
// threads > 1: the NTT stages are split over the workers and the (inner product, window) pairs of the five Pippenger sums are
// ONE task list pulled from an atomic counter, G2 windows first (they cost ~3x a G1 window) -- 5 x ceil(256 / c) tasks.
inline Proof<G1, G2> fast_prove(const SparseQap& q, const FastCrs& fc, const std::vector<Fr>& weights,
                                const Fr& r, const Fr& s, unsigned c = 0, unsigned threads = 1) {
    const BnCrs& crs = *fc.crs;
    size_t n = q.n();
    if (c == 0) c = q.log_n <= 8 ? 4 : (q.log_n <= 14 ? q.log_n - 4 : std::min(16u, q.log_n - 3));
    std::vector<Fr> U = q.eval_vec(q.u, weights), V = q.eval_vec(q.v, weights), W = q.eval_vec(q.w, weights);
    fr_ntt(U, q.log_n, true, threads); fr_ntt(V, q.log_n, true, threads); fr_ntt(W, q.log_n, true, threads);   // coefficient form
    // product on a domain of size 2n
    std::vector<Fr> A = U, B = V;
    A.resize(2 * n, Fr::zero()); B.resize(2 * n, Fr::zero());
    fr_ntt(A, q.log_n + 1, false, threads); fr_ntt(B, q.log_n + 1, false, threads);
    parallel_for(threads > 1 ? 64 : 1, threads, [&](size_t blk) {
        const size_t per = (2 * n + 63) / 64, lo = threads > 1 ? blk * per : 0, hi = threads > 1 ? std::min(2 * n, lo + per) : 2 * n;
        for (size_t i = lo; i < hi; ++i) A[i] = A[i] * B[i];
    });
    fr_ntt(A, q.log_n + 1, true, threads);
    for (size_t i = 0; i < n; ++i) A[i] = A[i] - W[i];           // P = U*V - W, deg <= 2n-2
    // long division by x^n - 1: q_k = r_{k+n}, r_k += r_{k+n}, top down
    std::vector<Fr> h(n > 0 ? n - 1 : 0, Fr::zero());
    for (size_t k = 2 * n - 1; k >= n; --k) { if (k - n < h.size()) h[k - n] = A[k]; A[k - n] = A[k - n] + A[k]; }
    auto scal = [](const std::vector<Fr>& v, size_t cnt) {
        std::vector<U256> out; for (size_t i = 0; i < std::min(cnt, v.size()); ++i) out.push_back(v[i].to_u256()); return out; };
    std::vector<Fr> wl(weights.begin() + std::min(weights.size(), q.input + 1), weights.end());
    auto su = scal(U, n), sv = scal(V, n), sh = scal(h, n), sl = scal(wl, (size_t)-1);
    const unsigned windows = (256 + c - 1) / c;
    std::vector<Jac<Fq>> wa(windows, Jac<Fq>::zero()), wb1 = wa, wh = wa, wl_ = wa;
    std::vector<Jac<Fq2>> wb2(windows, Jac<Fq2>::zero());
    parallel_for((size_t)5 * windows, threads, [&](size_t t) {
        const unsigned k = (unsigned)(t / windows), w = (unsigned)(t % windows), lo = w * c;
        if (k == 0) wb2[w] = msm_window(fc.xi2, sv, c, lo);
        if (k == 1) wl_[w] = msm_window(fc.sdl, sl, c, lo);
        if (k == 2) wa[w] = msm_window(fc.xi1, su, c, lo);
        if (k == 3) wb1[w] = msm_window(fc.xi1, sv, c, lo);
        if (k == 4) wh[w] = msm_window(fc.xit, sh, c, lo);
    });
    G1 a_g1 = msm_combine(wa, c), b_g1 = msm_combine(wb1, c), c_h = msm_combine(wh, c), c_l = msm_combine(wl_, c);
    G2 b_g2 = msm_combine(wb2, c);
    G1 a = a_g1 + crs.s1.alpha + crs.s1.delta.mul(r);
    G2 b = b_g2 + crs.s2.beta + crs.s2.delta.mul(s);
    G1 cc = c_h + c_l + a.mul(s) + (crs.s1.beta + b_g1 + crs.s1.delta.mul(s)).mul(r) - crs.s1.delta.mul(r * s);
    return Proof<G1, G2>{a, b, cc};
}
inline Proof<G1, G2> fast_prove(const SparseQap& q, const BnCrs& crs, const std::vector<Fr>& weights,
                                const Fr& r, const Fr& s, unsigned c = 0, unsigned threads = 1) {
    FastCrs fc(crs, q.n());
    return fast_prove(q, fc, weights, r, s, c, threads);
}

// Closed-form honest proof for a SparseQap (O(n) field work; usable at n = 2^20).
inline Proof<G1, G2> fast_trapdoor_proof(const SparseQap& q, const Trapdoor<Fr>& td, const std::vector<Fr>& weights,
                                         const Fr& r, const Fr& s) {
    std::vector<Fr> L = q.n_ap ? q.lagrange_at_integers(td.x) : q.lagrange_at(td.x);
    std::vector<Fr> ux = q.wire_evals(q.u, L), vx = q.wire_evals(q.v, L), wx = q.wire_evals(q.w, L);
    std::vector<Fr> Ue = q.eval_vec(q.u, weights), Ve = q.eval_vec(q.v, weights), We = q.eval_vec(q.w, weights);
    Fr U = Fr::zero(), V = Fr::zero(), W = Fr::zero(), rem = Fr::zero();
    size_t mm = std::min(q.m, weights.size());
    for (size_t i = 0; i < mm; ++i) { U = U + weights[i] * ux[i]; V = V + weights[i] * vx[i]; W = W + weights[i] * wx[i]; }
    // remainder of (UV - W) mod (x^n - 1) has evaluations Ue*Ve - We on the domain
    for (size_t j = 0; j < q.n(); ++j) rem = rem + (Ue[j] * Ve[j] - We[j]) * L[j];
    Fr hx_tx = U * V - W - rem;
    return trapdoor_proof(td, q.input, ux, vx, wx, hx_tx, weights, r, s);
}

}  // namespace orc
