// ff.cuh -- BN254 prime-field arithmetic for gfx950 (and the host side of this library).
//
// Replaces what the reference gets from crate bn 0.4.3 through FrLocal's operators
// (/root/reference/src/groth16/fr.rs:18-71): 254-bit Montgomery arithmetic.  Layout is
// MI355X-first: 8 x 32-bit limbs (the VALU's native multiply is v_mad_u64_u32: 32x32+64),
// one element per lane, 32 B per element in HBM so a lane moves an element with two
// global_load_dwordx4.
//
// Values are kept fully reduced in [0, p) and in Montgomery form (R = 2^256) on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZK_HD __host__ __device__ __forceinline__

namespace zk {

struct FrParams {
    // r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    static constexpr uint32_t P[8]  = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R1[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t INV = 0xefffffffu;  // -p^-1 mod 2^32
};
struct FqParams {
    // q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    static constexpr uint32_t P[8]  = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R1[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t INV = 0xe4866389u;
};

template <class PR>
struct alignas(16) Fp {
    uint32_t l[8];

    ZK_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = 0;
        return r;
    }
    ZK_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = PR::R1[i];
        return r;
    }
    ZK_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = PR::R2[i];
        return r;
    }
    ZK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i];
        return o == 0;
    }
    ZK_HD bool operator==(const Fp& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= l[i] ^ b.l[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Fp& b) const { return !(*this == b); }

    // r = a - p if a >= p (a < 2p assumed), given the carry-out of the addition that made a
    ZK_HD static Fp reduce_once(const Fp& a, uint32_t carry) {
        Fp d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)a.l[i] - PR::P[i] - br;
            d.l[i] = (uint32_t)t;
            br = (uint32_t)(t >> 63);
        }
        bool use_d = carry | (br == 0);
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = use_d ? d.l[i] : a.l[i];
        return r;
    }
    ZK_HD Fp operator+(const Fp& b) const {
        Fp s;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)l[i] + b.l[i];
            s.l[i] = (uint32_t)c;
            c >>= 32;
        }
        return reduce_once(s, (uint32_t)c);
    }
    ZK_HD Fp operator-(const Fp& b) const {
        Fp d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)l[i] - b.l[i] - br;
            d.l[i] = (uint32_t)t;
            br = (uint32_t)(t >> 63);
        }
        // add p back when the subtraction borrowed
        uint32_t mask = 0u - br;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)d.l[i] + (PR::P[i] & mask);
            d.l[i] = (uint32_t)c;
            c >>= 32;
        }
        return d;
    }
    ZK_HD Fp operator-() const { return zero() - *this; }
    ZK_HD Fp dbl() const { return *this + *this; }

    // CIOS Montgomery product a*b*R^-1 mod p; 8x8 limb products as 64-bit multiply-adds
    // (v_mad_u64_u32 on gfx950).
    ZK_HD Fp operator*(const Fp& b) const {
        uint32_t t[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t c = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                c = (uint64_t)l[j] * b.l[i] + t[j] + c;
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            c += t[8];
            t[8] = (uint32_t)c;
            t[9] = (uint32_t)(c >> 32);
            uint32_t m = t[0] * PR::INV;
            c = (uint64_t)m * PR::P[0] + t[0];
            c >>= 32;
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                c = (uint64_t)m * PR::P[j] + t[j] + c;
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            c += t[8];
            t[7] = (uint32_t)c;
            t[8] = t[9] + (uint32_t)(c >> 32);
        }
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = t[i];
        return reduce_once(r, t[8]);
    }
    ZK_HD Fp sqr() const { return *this * *this; }

    // canonical integer <-> Montgomery
    ZK_HD static Fp from_canonical(const Fp& x) { return x * r2(); }
    ZK_HD Fp to_canonical() const {
        Fp o = zero();
        o.l[0] = 1;
        return *this * o;
    }
    ZK_HD static Fp from_u32(uint32_t v) {
        Fp x = zero();
        x.l[0] = v;
        return from_canonical(x);
    }
    // true iff the raw limbs encode an integer < p
    ZK_HD bool raw_in_range() const {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t t = (uint64_t)l[i] - PR::P[i] - br;
            br = (uint32_t)(t >> 63);
        }
        return br != 0;
    }
    // x^e, e given as 8 little-endian words (not secret-dependent timing: this is a prover)
    ZK_HD Fp pow_words(const uint32_t* e) const {
        Fp acc = one();
        for (int i = 255; i >= 0; --i) {
            acc = acc.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1) acc = acc * *this;
        }
        return acc;
    }
    // Fermat inverse x^(p-2); inverse of zero is zero here, callers that mirror the reference's
    // panic on division by zero (fr.rs:54,69) test is_zero() first.
    ZK_HD Fp inv() const {
        uint32_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = PR::P[i];
        e[0] -= 2;  // p is odd and P[0] >= 2 for both moduli
        return pow_words(e);
    }
};

typedef Fp<FrParams> Fr;
typedef Fp<FqParams> Fq;

// ---- Fq2 = Fq[i]/(i^2+1) -----------------------------------------------------------------
struct Fq2 {
    Fq c0, c1;
    ZK_HD static Fq2 zero() { return Fq2{Fq::zero(), Fq::zero()}; }
    ZK_HD static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool operator==(const Fq2& o) const { return c0 == o.c0 && c1 == o.c1; }
    ZK_HD bool operator!=(const Fq2& o) const { return !(*this == o); }
    ZK_HD Fq2 operator+(const Fq2& o) const { return Fq2{c0 + o.c0, c1 + o.c1}; }
    ZK_HD Fq2 operator-(const Fq2& o) const { return Fq2{c0 - o.c0, c1 - o.c1}; }
    ZK_HD Fq2 operator-() const { return Fq2{-c0, -c1}; }
    ZK_HD Fq2 dbl() const { return Fq2{c0.dbl(), c1.dbl()}; }
    ZK_HD Fq2 operator*(const Fq2& o) const {
        Fq aa = c0 * o.c0, bb = c1 * o.c1;
        Fq s = (c0 + c1) * (o.c0 + o.c1);
        return Fq2{aa - bb, s - aa - bb};
    }
    ZK_HD Fq2 sqr() const {
        Fq ab = c0 * c1;
        return Fq2{(c0 + c1) * (c0 - c1), ab.dbl()};
    }
    ZK_HD Fq2 inv() const {
        Fq d = (c0.sqr() + c1.sqr()).inv();
        return Fq2{c0 * d, -(c1 * d)};
    }
    ZK_HD static Fq2 from_canonical(const Fq2& x) { return Fq2{Fq::from_canonical(x.c0), Fq::from_canonical(x.c1)}; }
    ZK_HD Fq2 to_canonical() const { return Fq2{c0.to_canonical(), c1.to_canonical()}; }
    ZK_HD bool raw_in_range() const { return c0.raw_in_range() && c1.raw_in_range(); }
};

}  // namespace zk
