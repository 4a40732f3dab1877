// oracle/bn254.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in
// oracle/.  The product (zksnark_rs_amd/) never includes, links or calls this code.
//
// What this restates: the arithmetic the reference obtains from the third-party crate
// `bn = "0.4.3"` (zcash-hackworks alt_bn128; /root/reference/Cargo.toml:14), which is NOT
// vendored under /root/reference and cannot be built here (no Rust toolchain, no network).
// Reference call sites that define what is needed:
//   Fr  +,-,*,neg,inverse,from_str,random        /root/reference/src/groth16/fr.rs:18-99
//   G1::one(), G2::one(), G*Fr, G+G, G-G, zero   /root/reference/src/groth16/fr.rs:106-119,175-223
// Published algorithm restated: 254-bit prime fields in Montgomery form (R = 2^256),
// Fq2 = Fq[i]/(i^2+1), short-Weierstrass y^2 = x^3 + b (a = 0) in Jacobian coordinates,
// scalar multiplication MSB-first double-and-add (what bn's Mul<Fr> does [recollection]).
//
// PARITY STATUS: "parity unpinned" at the bn byte boundary -- the reference holds no
// known-answer vector for any Fr/G1/G2 value.  Group elements are pinned by
// mathematics instead: published curve constants (checked in tests: primality-independent
// identities, generators on curve, r*G = inf) and canonical AFFINE coordinates, which do not
// depend on the coordinate system or summation order.  Cross-checked against the independent
// big-int twin oracle/pyref.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <stdexcept>
#include <vector>

namespace orc {

typedef unsigned __int128 u128;

struct U256 {
    uint64_t l[4];
    bool operator==(const U256& o) const { return l[0] == o.l[0] && l[1] == o.l[1] && l[2] == o.l[2] && l[3] == o.l[3]; }
    bool operator!=(const U256& o) const { return !(*this == o); }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool bit(int i) const { return (l[i >> 6] >> (i & 63)) & 1; }
};

static inline int u256_cmp(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] < b.l[i]) return -1;
        if (a.l[i] > b.l[i]) return 1;
    }
    return 0;
}
// a += b, returns carry
static inline uint64_t u256_add(U256& a, const U256& b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; a.l[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
// a -= b, returns borrow
static inline uint64_t u256_sub(U256& a, const U256& b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - br;
        a.l[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}

// ---------------------------------------------------------------------------------------
// Prime field in Montgomery form, 4 x 64-bit limbs, R = 2^256.
// ---------------------------------------------------------------------------------------
struct FrParams {
    static constexpr U256 P   = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
    static constexpr U256 R1  = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
    static constexpr U256 R2  = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;  // -P^-1 mod 2^64
};
struct FqParams {
    static constexpr U256 P   = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
    static constexpr U256 R1  = {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}};
    static constexpr U256 R2  = {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}};
    static constexpr uint64_t INV = 0x87d20782e4866389ULL;
};

template <class PR>
struct Fp {
    U256 v;  // Montgomery representation, always fully reduced (< P)

    static Fp zero() { Fp r; r.v = U256{{0, 0, 0, 0}}; return r; }
    static Fp one() { Fp r; r.v = PR::R1; return r; }
    static Fp from_u256(const U256& x) {  // canonical integer (reduced mod P first if needed)
        Fp r; r.v = x;
        while (u256_cmp(r.v, PR::P) >= 0) u256_sub(r.v, PR::P);
        Fp r2; r2.v = PR::R2;
        return r * r2;
    }
    static Fp from_u64(uint64_t x) { return from_u256(U256{{x, 0, 0, 0}}); }
    U256 to_u256() const {  // canonical integer
        Fp o; o.v = U256{{1, 0, 0, 0}};
        return (*this * o).v;
    }
    bool is_zero() const { return v.is_zero(); }
    bool operator==(const Fp& o) const { return v == o.v; }
    bool operator!=(const Fp& o) const { return !(v == o.v); }

    Fp operator+(const Fp& o) const {
        Fp r = *this;
        uint64_t c = u256_add(r.v, o.v);
        if (c || u256_cmp(r.v, PR::P) >= 0) u256_sub(r.v, PR::P);
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r = *this;
        if (u256_sub(r.v, o.v)) u256_add(r.v, PR::P);
        return r;
    }
    Fp operator-() const {
        if (is_zero()) return *this;
        Fp r; r.v = PR::P; u256_sub(r.v, v); return r;
    }
    Fp dbl() const { return *this + *this; }
    // CIOS Montgomery multiplication
    Fp operator*(const Fp& o) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128)v.l[j] * o.v.l[i] + t[j];
                t[j] = (uint64_t)c; c >>= 64;
            }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * PR::INV;
            c = (u128)m * PR::P.l[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; ++j) {
                c += (u128)m * PR::P.l[j] + t[j];
                t[j - 1] = (uint64_t)c; c >>= 64;
            }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fp r; r.v = U256{{t[0], t[1], t[2], t[3]}};
        if (t[4] || u256_cmp(r.v, PR::P) >= 0) u256_sub(r.v, PR::P);
        return r;
    }
    Fp sqr() const { return *this * *this; }
    Fp pow(const U256& e) const {
        Fp acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc.sqr();
            if (e.bit(i)) acc = acc * *this;
        }
        return acc;
    }
    // Fr::inverse() returns None on zero; FrLocal `/` and mul_inv() panic
    // (/root/reference/src/groth16/fr.rs:50-71).  Here: throws.
    Fp inv() const {
        if (is_zero()) throw std::domain_error("Tried to divide by zero");
        U256 e = PR::P; U256 two = {{2, 0, 0, 0}}; u256_sub(e, two);
        return pow(e);
    }
    Fp operator/(const Fp& o) const { return *this * o.inv(); }

    // bn's Fr::from_str [recollection]: decimal digits only, accumulates res = res*10 + d in the
    // field (so values >= P wrap); any other char -> None.  Empty string -> Some(0).
    static bool from_str(const std::string& s, Fp& out) {
        Fp ten = from_u64(10), res = zero();
        for (char ch : s) {
            if (ch < '0' || ch > '9') return false;
            res = res * ten + from_u64((uint64_t)(ch - '0'));
        }
        out = res;
        return true;
    }
    static Fp from_usize(size_t n) { return from_u64((uint64_t)n); }  // fr.rs:73-77 (decimal string round trip)
};

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

// ---------------------------------------------------------------------------------------
// Fq2 = Fq[i] / (i^2 + 1)
// ---------------------------------------------------------------------------------------
struct Fq2 {
    Fq c0, c1;
    static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fq2& o) const { return !(*this == o); }
    Fq2 operator+(const Fq2& o) const { return Fq2{c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2& o) const { return Fq2{c0 - o.c0, c1 - o.c1}; }
    Fq2 operator-() const { return Fq2{-c0, -c1}; }
    Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
    Fq2 operator*(const Fq2& o) const {
        Fq aa = c0 * o.c0, bb = c1 * o.c1;
        Fq s = (c0 + c1) * (o.c0 + o.c1);
        return Fq2{aa - bb, s - aa - bb};
    }
    Fq2 sqr() const {
        Fq ab = c0 * c1;
        return Fq2{(c0 + c1) * (c0 - c1), ab.dbl()};
    }
    Fq2 inv() const {
        Fq d = (c0.sqr() + c1.sqr()).inv();
        return Fq2{c0 * d, -(c1 * d)};
    }
};

// ---------------------------------------------------------------------------------------
// y^2 = x^3 + b, a = 0.  Jacobian (X:Y:Z), x = X/Z^2, y = Y/Z^3, infinity <=> Z = 0.
// ---------------------------------------------------------------------------------------
template <class F>
struct Affine {
    F x, y;
    bool inf;
    static Affine infinity() { return Affine{F::zero(), F::zero(), true}; }
    bool operator==(const Affine& o) const { return inf == o.inf && (inf || (x == o.x && y == o.y)); }
    bool operator!=(const Affine& o) const { return !(*this == o); }
};

template <class F>
struct Jac {
    F X, Y, Z;
    static Jac zero() { return Jac{F::zero(), F::one(), F::zero()}; }  // G::zero()
    static Jac from_affine(const Affine<F>& a) { return a.inf ? zero() : Jac{a.x, a.y, F::one()}; }
    bool is_zero() const { return Z.is_zero(); }

    Jac dbl() const {
        if (is_zero()) return *this;
        F A = X.sqr(), B = Y.sqr(), C = B.sqr();
        F D = ((X + B).sqr() - A - C).dbl();
        F E = A.dbl() + A;
        F Fv = E.sqr();
        F X3 = Fv - D.dbl();
        F Y3 = E * (D - X3) - C.dbl().dbl().dbl();
        F Z3 = (Y * Z).dbl();
        return Jac{X3, Y3, Z3};
    }
    Jac operator+(const Jac& o) const {
        if (is_zero()) return o;
        if (o.is_zero()) return *this;
        F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        F U1 = X * Z2Z2, U2 = o.X * Z1Z1;
        F S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
        if (U1 == U2) {
            if (S1 == S2) return dbl();
            return zero();
        }
        F H = U2 - U1;
        F I = H.dbl().sqr();
        F J = H * I;
        F rr = (S2 - S1).dbl();
        F V = U1 * I;
        F X3 = rr.sqr() - J - V.dbl();
        F Y3 = rr * (V - X3) - (S1 * J).dbl();
        F Z3 = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
        return Jac{X3, Y3, Z3};
    }
    Jac operator-() const { return Jac{X, -Y, Z}; }
    Jac operator-(const Jac& o) const { return *this + (-o); }
    // point * scalar: MSB-first double-and-add over the canonical 256-bit scalar.
    Jac mul(const U256& k) const {
        Jac acc = zero();
        bool started = false;
        for (int i = 255; i >= 0; --i) {
            if (started) acc = acc.dbl();
            if (k.bit(i)) { acc = acc + *this; started = true; }
        }
        return acc;
    }
    Jac mul(const Fr& k) const { return mul(k.to_u256()); }
    Affine<F> to_affine() const {
        if (is_zero()) return Affine<F>::infinity();
        F zi = Z.inv();
        F zi2 = zi.sqr();
        return Affine<F>{X * zi2, Y * zi2 * zi, false};
    }
    // group-element equality (PartialEq on bn's G1/G2 compares the represented points)
    bool operator==(const Jac& o) const { return to_affine() == o.to_affine(); }
    bool operator!=(const Jac& o) const { return !(*this == o); }
};

typedef Jac<Fq> G1;
typedef Jac<Fq2> G2;
typedef Affine<Fq> G1A;
typedef Affine<Fq2> G2A;

static inline U256 u256_from_dec(const char* s) {
    // small helper for constants given in decimal
    U256 r = {{0, 0, 0, 0}};
    for (; *s; ++s) {
        u128 c = (u128)(*s - '0');
        for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] * 10; r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
}

static inline G1 g1_generator() { return G1{Fq::from_u64(1), Fq::from_u64(2), Fq::one()}; }
static inline G2 g2_generator() {
    Fq2 x{Fq::from_u256(u256_from_dec("10857046999023057135944570762232829481370756359578518086990519993285655852781")),
          Fq::from_u256(u256_from_dec("11559732032986387107991004021392285783925812861821192530917403151452391805634"))};
    Fq2 y{Fq::from_u256(u256_from_dec("8495653923123431417604973247489272438418190587263600148770280649306958101930")),
          Fq::from_u256(u256_from_dec("4082367875863433681332203403145435568316851327593401208105741076214120093531"))};
    return G2{x, y, Fq2::one()};
}
static inline Fq g1_b() { return Fq::from_u64(3); }
static inline Fq2 g2_b() { return Fq2{Fq::from_u64(3), Fq::zero()} * Fq2{Fq::from_u64(9), Fq::from_u64(1)}.inv(); }
static inline bool on_curve(const G1A& p) { return p.inf || p.y.sqr() == p.x.sqr() * p.x + g1_b(); }
static inline bool on_curve(const G2A& p) { return p.inf || p.y.sqr() == p.x.sqr() * p.x + g2_b(); }

// Encryption bases, /root/reference/src/groth16/fr.rs:106-113
static inline const G1& enc_base_g1() { static G1 g = g1_generator().mul(U256{{69, 0, 0, 0}}); return g; }
static inline const G2& enc_base_g2() { static G2 g = g2_generator().mul(U256{{96, 0, 0, 0}}); return g; }

// 2^28-th primitive root of unity of Fr: 5^((r-1)/2^28)
static inline Fr fr_w28_compute() {
    U256 e = FrParams::P;
    e.l[0] -= 1;  // r-1
    U256 s = {{(e.l[0] >> 28) | (e.l[1] << 36), (e.l[1] >> 28) | (e.l[2] << 36), (e.l[2] >> 28) | (e.l[3] << 36), e.l[3] >> 28}};
    return Fr::from_u64(5).pow(s);
}
static inline Fr fr_root_of_unity(int log_n) {
    // canonical value of 5^((r-1)/2^28); bn_constants KAT re-derives it with fr_w28_compute()
    Fr w = Fr::from_u256(U256{{0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}});
    for (int i = log_n; i < 28; ++i) w = w.sqr();
    return w;
}

// ---------------------------------------------------------------------------------------
// Canonical encodings (build-defined; documented in DESIGN.md)
// ---------------------------------------------------------------------------------------
static inline void be32(const U256& x, uint8_t* out) {
    for (int i = 0; i < 4; ++i)
        for (int b = 0; b < 8; ++b) out[31 - (i * 8 + b)] = (uint8_t)(x.l[i] >> (8 * b));
}
static inline void encode_g1(const G1& p, uint8_t out[65]) {
    G1A a = p.to_affine();
    std::memset(out, 0, 65);
    if (a.inf) return;
    out[0] = 4; be32(a.x.to_u256(), out + 1); be32(a.y.to_u256(), out + 33);
}
static inline void encode_g2(const G2& p, uint8_t out[129]) {
    G2A a = p.to_affine();
    std::memset(out, 0, 129);
    if (a.inf) return;
    out[0] = 4;
    be32(a.x.c1.to_u256(), out + 1); be32(a.x.c0.to_u256(), out + 33);
    be32(a.y.c1.to_u256(), out + 65); be32(a.y.c0.to_u256(), out + 97);
}

// SplitMix64 stream shared with pyref.py / the product / bench.py
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    // uniform non-zero Fr (Random for FrLocal rejects zero, fr.rs:90-99)
    Fr fr() {
        for (;;) {
            U256 x;
            for (int i = 0; i < 4; ++i) x.l[i] = next();
            x.l[3] &= (1ULL << 62) - 1;
            if (!x.is_zero() && u256_cmp(x, FrParams::P) < 0) return Fr::from_u256(x);
        }
    }
};

}  // namespace orc
