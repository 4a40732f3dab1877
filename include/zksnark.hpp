// zksnark.hpp -- C++17 host side above the C ABI (zkgpu.h), mirroring the reference crate's API for the
// accelerated path: same names, argument order and meaning, and error behaviour, so that code and tests
// written against republicprotocol/zksnark-rs read the same here.  Header-only; link with -lzkgpu.
//
//   reference (Rust)                                              here
//   ------------------------------------------------------------  -------------------------------------------
//   FrLocal, From<usize>, FromStr, + - * / neg   (fr.rs:18-99)     zksnark::FrLocal (canonical 4 x u64; arithmetic
//                                                                  through zk_fr_batch on the GPU)
//   ASTParser::try_parse(code) -> RootRepresentation              zksnark::ASTParser::try_parse(code) -> Circuit
//                                 (circuit/mod.rs:224-527)
//   groth16::weights(code, assignments)  (circuit/mod.rs:529-637)  zksnark::groth16::weights(code, assignments)
//   QAP<CoefficientPoly<FrLocal>>: From<root_rep> (fr.rs:140-173)  zksnark::QAP::from(ctx, circuit)
//   QAP { u, v, w, t, input, degree }    (groth16/mod.rs:60-67)    zksnark::QAP::from_dense(ctx, u, v, w, t, input)
//   groth16::setup(&qap) -> (SigmaG1, SigmaG2)  (mod.rs:134-197)   zksnark::groth16::setup(ctx, qap) -> Sigma
//   groth16::prove(&qap, (&s1, &s2), &weights)  (mod.rs:213-296)   zksnark::groth16::prove(ctx, qap, sigma, weights)
//   groth16::verify((s1, s2), &inputs, proof)   (mod.rs:299-320)   zksnark::groth16::verify(ctx, sigma, inputs, proof)
//
// Where the reference panics (division by zero fr.rs:54,69; "Dividend must be non-zero" field/mod.rs:440;
// unwrap() of a ParseErr) this API throws zksnark::Error carrying the ABI status and message.  The
// randomness the reference draws inside setup / prove (thread_rng, mod.rs:139-145,231) is drawn here from
// std::random_device; the *_with variants take it as arguments (tests, reproducible runs).
#pragma once
#include <array>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zkgpu.h"

namespace zksnark {

class Error : public std::runtime_error {
   public:
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// One device context; not shareable between threads without external locking (as the ABI).
class Context {
   public:
    explicit Context(int device = 0) {
        int rc = zk_ctx_create(device, &ctx_);
        if (rc != ZK_OK) throw Error(rc, std::string("zk_ctx_create: ") + zk_strerror(rc));   // no GPU: no CPU fallback
    }
    ~Context() { zk_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    zk_ctx* get() const { return ctx_; }
    void check(int rc, const char* where) const {
        if (rc != ZK_OK) throw Error(rc, std::string(where) + ": " + zk_strerror(rc) + " (" + zk_last_error(ctx_) + ")");
    }

   private:
    zk_ctx* ctx_ = nullptr;
};

// ---- FrLocal (fr.rs:9-99): an element of the BN254 scalar field, canonical integer --------------
struct FrLocal {
    std::array<uint64_t, 4> w{};   // little-endian words
    FrLocal() = default;
    FrLocal(uint64_t v) { w[0] = v; }                                   // From<usize> (fr.rs:73-77)
    bool operator==(const FrLocal& o) const { return w == o.w; }
    bool operator!=(const FrLocal& o) const { return !(*this == o); }
    static constexpr std::array<uint64_t, 4> MODULUS = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    bool in_range() const {
        for (int i = 3; i >= 0; --i)
            if (w[i] != MODULUS[i]) return w[i] < MODULUS[i];
        return false;
    }
    // Random::random_elem (fr.rs:89-99): uniform, rejection sampled
    static FrLocal random_elem() {
        static thread_local std::random_device rd;
        FrLocal x;
        do {
            for (auto& l : x.w) l = ((uint64_t)rd() << 32) | rd();
            x.w[3] &= (1ull << 62) - 1;
        } while (!x.in_range());
        return x;
    }
};

// field arithmetic needs a device; bind one Context to get operators with the reference's spelling
class Field {
   public:
    explicit Field(const Context& c) : c_(c) {}
    FrLocal add(const FrLocal& a, const FrLocal& b) const { return op(0, a, b); }
    FrLocal sub(const FrLocal& a, const FrLocal& b) const { return op(1, a, b); }
    FrLocal mul(const FrLocal& a, const FrLocal& b) const { return op(2, a, b); }
    FrLocal inv(const FrLocal& a) const { return op(3, a, a); }           // panics on zero in the reference (fr.rs:54,69)
    FrLocal div(const FrLocal& a, const FrLocal& b) const { return mul(a, inv(b)); }
    FrLocal neg(const FrLocal& a) const { return sub(FrLocal(0), a); }

   private:
    FrLocal op(int code, const FrLocal& a, const FrLocal& b) const {
        FrLocal r;
        c_.check(zk_fr_batch(c_.get(), code, a.w.data(), b.w.data(), r.w.data(), 1), "FrLocal arithmetic");
        return r;
    }
    const Context& c_;
};

// ---- front end -----------------------------------------------------------------------------------
// RootRepresentation produced by ASTParser::try_parse (circuit/mod.rs:224-527)
class Circuit {
   public:
    Circuit(Circuit&& o) noexcept : c_(std::exchange(o.c_, nullptr)) {}
    Circuit(const Circuit&) = delete;
    ~Circuit() { if (c_) zk_circuit_free(c_); }
    const zk_circuit* get() const { return c_; }
    size_t wires() const { return dims()[0]; }
    size_t gates() const { return dims()[1]; }
    size_t input() const { return dims()[2]; }
    size_t assignments() const { return dims()[3]; }
    // circuit::weights: [1] ++ every wire value, inputs in `in` order
    std::vector<FrLocal> weights(const std::vector<FrLocal>& assignments_in) const {
        std::vector<FrLocal> out(wires());
        int rc = zk_circuit_weights(c_, assignments_in.empty() ? nullptr : assignments_in[0].w.data(), assignments_in.size(), out[0].w.data(), out.size());
        if (rc != ZK_OK) throw Error(rc, std::string("weights: ") + zk_circuit_last_error(c_));
        return out;
    }

   private:
    friend struct ASTParser;
    explicit Circuit(zk_circuit* c) : c_(c) {}
    std::array<size_t, 4> dims() const {
        std::array<size_t, 4> d{};
        zk_circuit_dims(c_, &d[0], &d[1], &d[2], &d[3]);
        return d;
    }
    zk_circuit* c_;
};

struct ASTParser {
    // Result<_, ParseErr>: the Err arm is thrown, its text is the ParseErr's Debug form
    static Circuit try_parse(const std::string& code) {
        zk_circuit* c = nullptr;
        char err[512] = {0};
        int rc = zk_circuit_parse(code.c_str(), &c, err, sizeof(err));
        if (rc != ZK_OK) throw Error(rc, std::string("ParseErr: ") + err);
        return Circuit(c);
    }
};

// ---- QAP<CoefficientPoly<FrLocal>> (groth16/mod.rs:60-67), device resident ------------------------
class QAP {
   public:
    QAP(QAP&& o) noexcept : q_(std::exchange(o.q_, nullptr)) {}
    QAP(const QAP&) = delete;
    ~QAP() { if (q_) zk_qap_free(q_); }
    // From<RootRepresentation> (fr.rs:140-173): Lagrange interpolation over the circuit's roots 1..n
    static QAP from(const Context& c, const Circuit& circuit) {
        zk_qap* q = nullptr;
        c.check(zk_circuit_qap(c.get(), circuit.get(), &q), "QAP::from");
        return QAP(q);
    }
    // the struct literal QAP { u, v, w, t, input, degree }: u, v, w = m polynomials of `degree` coefficients
    // (shorter ones are zero padded), t = degree + 1 coefficients
    static QAP from_dense(const Context& c, const std::vector<std::vector<FrLocal>>& u, const std::vector<std::vector<FrLocal>>& v,
                          const std::vector<std::vector<FrLocal>>& w, const std::vector<FrLocal>& t, size_t input) {
        const size_t m = u.size(), n = t.size() - 1;
        auto flat = [&](const std::vector<std::vector<FrLocal>>& p) {
            std::vector<uint64_t> out(m * n * 4, 0);
            for (size_t i = 0; i < m; ++i)
                for (size_t k = 0; k < p[i].size() && k < n; ++k)
                    for (int l = 0; l < 4; ++l) out[(i * n + k) * 4 + l] = p[i][k].w[l];
            return out;
        };
        if (v.size() != m || w.size() != m) throw Error(ZK_ERR_ARG, "QAP: u, v, w must have one polynomial per wire");
        auto fu = flat(u), fv = flat(v), fw = flat(w);
        zk_qap* q = nullptr;
        c.check(zk_qap_upload_dense(c.get(), fu.data(), fv.data(), fw.data(), t[0].w.data(), m, n, input, &q), "QAP");
        return QAP(q);
    }
    // From<RootRepresentation> for ANY root representation (circuit/mod.rs:201-214; fr.rs:140-173): u, v, w = one row per wire of
    // (root, value) pairs, roots = the representation's distinct roots in gate order, `input` as RootRepresentation::input().  The
    // wire polynomials are never interpolated (zk_qap_upload_sparse_roots): any size up to 2^22 gates, the reference's proof bytes.
    typedef std::vector<std::vector<std::pair<FrLocal, FrLocal>>> Rows;
    static QAP from_root_rep(const Context& c, const std::vector<FrLocal>& roots, const Rows& u, const Rows& v, const Rows& w, size_t input) {
        const size_t n = roots.size(), m = u.size();
        if (v.size() != m || w.size() != m) throw Error(ZK_ERR_ARG, "QAP: u, v, w must have one row per wire");
        auto index_of = [&](const FrLocal& r) {
            for (size_t j = 0; j < n; ++j) if (roots[j].w == r.w) return (uint32_t)j;
            throw Error(ZK_ERR_ARG, "QAP: a row names a root that roots() does not list");
        };
        struct Csr { std::vector<uint64_t> ptr, val; std::vector<uint32_t> gate; };
        auto pack = [&](const Rows& rows) {
            Csr x;
            x.ptr.push_back(0);
            for (const auto& row : rows) {
                for (const auto& e : row) {
                    x.gate.push_back(index_of(e.first));
                    x.val.insert(x.val.end(), e.second.w.begin(), e.second.w.end());
                }
                x.ptr.push_back(x.gate.size());
            }
            if (x.gate.empty()) { x.gate.push_back(0); x.val.resize(4); }   // non-null pointers for empty rows
            return x;
        };
        Csr cu = pack(u), cv = pack(v), cw = pack(w);
        std::vector<uint64_t> rw(4 * n);
        for (size_t j = 0; j < n; ++j) std::copy(roots[j].w.begin(), roots[j].w.end(), rw.begin() + 4 * j);
        zk_qap_sparse_desc d{};
        d.log_n = 0; d.m = m; d.input = input;
        d.u = zk_sparse_rows{cu.ptr.data(), cu.gate.data(), cu.val.data()};
        d.v = zk_sparse_rows{cv.ptr.data(), cv.gate.data(), cv.val.data()};
        d.w = zk_sparse_rows{cw.ptr.data(), cw.gate.data(), cw.val.data()};
        zk_qap* q = nullptr;
        c.check(zk_qap_upload_sparse_roots(c.get(), &d, rw.data(), n, &q), "QAP::from_root_rep");
        return QAP(q);
    }
    const zk_qap* get() const { return q_; }
    size_t degree() const { size_t n, m, l; int d; zk_qap_dims(q_, &n, &m, &l, &d); return n; }
    size_t wires() const { size_t n, m, l; int d; zk_qap_dims(q_, &n, &m, &l, &d); return m; }
    size_t input() const { size_t n, m, l; int d; zk_qap_dims(q_, &n, &m, &l, &d); return l; }

   private:
    explicit QAP(zk_qap* q) : q_(q) {}
    zk_qap* q_;
};

namespace groth16 {

// (SigmaG1, SigmaG2) (groth16/mod.rs:105-121), device resident
class Sigma {
   public:
    Sigma(Sigma&& o) noexcept : s_(std::exchange(o.s_, nullptr)) {}
    Sigma(const Sigma&) = delete;
    ~Sigma() { if (s_) zk_crs_free(s_); }
    const zk_crs* get() const { return s_; }
    void save(const Context& c, const std::string& path) const { c.check(zk_crs_save(c.get(), s_, path.c_str()), "Sigma::save"); }
    static Sigma load(const Context& c, const std::string& path) {
        zk_crs* s = nullptr;
        c.check(zk_crs_load(c.get(), path.c_str(), &s), "Sigma::load");
        return Sigma(s);
    }

   private:
    friend Sigma setup_with(const Context&, const QAP&, const std::array<FrLocal, 5>&);
    explicit Sigma(zk_crs* s) : s_(s) {}
    zk_crs* s_;
};

// Proof { a, b, c } (groth16/mod.rs:124-128) in the canonical 259-byte encoding
struct Proof {
    std::array<uint8_t, ZK_PROOF_BYTES> bytes{};
    bool operator==(const Proof& o) const { return bytes == o.bytes; }
};

// weights(code, assignments) (circuit/mod.rs:529-637)
inline std::vector<FrLocal> weights(const std::string& code, const std::vector<FrLocal>& assignments) {
    return ASTParser::try_parse(code).weights(assignments);
}

// setup with the trapdoor (alpha, beta, gamma, delta, x) given
inline Sigma setup_with(const Context& c, const QAP& qap, const std::array<FrLocal, 5>& trapdoor) {
    uint64_t td[20];
    for (int k = 0; k < 5; ++k)
        for (int l = 0; l < 4; ++l) td[4 * k + l] = trapdoor[k].w[l];
    zk_crs* s = nullptr;
    c.check(zk_setup(c.get(), qap.get(), td, &s), "groth16::setup");
    return Sigma(s);
}
// groth16::setup(&qap) (mod.rs:134-197): five non-zero random draws (mod.rs:139-145)
inline Sigma setup(const Context& c, const QAP& qap) {
    std::array<FrLocal, 5> td;
    for (auto& t : td) do { t = FrLocal::random_elem(); } while (t == FrLocal(0));
    return setup_with(c, qap, td);
}

// prove with the blinding scalars given
inline Proof prove_with(const Context& c, const QAP& qap, const Sigma& sigma, const std::vector<FrLocal>& weights_, const FrLocal& r, const FrLocal& s) {
    Proof p;
    c.check(zk_prove(c.get(), sigma.get(), qap.get(), weights_.empty() ? nullptr : weights_[0].w.data(), weights_.size(), r.w.data(), s.w.data(), p.bytes.data()),
            "groth16::prove");
    return p;
}
// groth16::prove(&qap, (&sigma_g1, &sigma_g2), &weights) (mod.rs:213-296): r, s drawn inside (mod.rs:231)
inline Proof prove(const Context& c, const QAP& qap, const Sigma& sigma, const std::vector<FrLocal>& weights_) {
    return prove_with(c, qap, sigma, weights_, FrLocal::random_elem(), FrLocal::random_elem());
}

// groth16::verify((sigma_g1, sigma_g2), &inputs, proof) (mod.rs:299-320)
inline bool verify(const Context& c, const Sigma& sigma, const std::vector<FrLocal>& inputs, const Proof& proof) {
    int ok = 0;
    c.check(zk_verify(c.get(), sigma.get(), inputs.empty() ? nullptr : inputs[0].w.data(), inputs.size(), proof.bytes.data(), &ok), "groth16::verify");
    return ok != 0;
}

}  // namespace groth16
}  // namespace zksnark
