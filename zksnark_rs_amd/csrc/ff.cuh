// ff.cuh -- BN254 prime-field arithmetic for gfx950 (and the host side of this library).
//
// Replaces what the reference gets from crate bn 0.4.3 through FrLocal's operators
// (/root/reference/src/groth16/fr.rs:18-71): 254-bit Montgomery arithmetic.  Layout is
// MI355X-first: 8 x 32-bit limbs (the VALU's native multiply is v_mad_u64_u32: 32x32+64),
// one element per lane, 32 B per element in HBM so a lane moves an element with two
// global_load_dwordx4.
//
// Values are kept fully reduced in [0, p) and in Montgomery form (R = 2^261) on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZK_HD __host__ __device__ __forceinline__
// Keeps the instruction scheduler from interleaving independent field multiplications: each one
// is ~230 VALU instructions with plenty of internal ILP, and interleaving several only inflates
// the live register set of the curve formulas (occupancy 1-2 waves/SIMD, spills in G2).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_SCHED_FENCE)
#define ZK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ZK_SCHED_FENCE() ((void)0)
#endif

// Wave priority of the short, dependent kernels of a proof (SpMV / NTT stage, the sort, the reduction tails, assembly).  The
// bucket accumulations fill every SIMD with waves that live ~0.5 ms, and the SIMD arbitrates VALU issue by priority first,
// age second: at equal priority a freshly launched wave of a small kernel gets the slots the older accumulation waves leave
// over, so each of the ~100 dependent launches of a proof crawled while an accumulation ran (the timeline showed the scalars
// of an inner product arriving 1-2 ms after the chip went idle).  Raised priority makes them finish in their stand-alone time;
// the accumulation loses only the issue slots those few waves actually use.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_PRIO)
#define ZK_LATENCY_KERNEL() __builtin_amdgcn_s_setprio(3)
#else
#define ZK_LATENCY_KERNEL() ((void)0)
#endif

namespace zk {

struct FrParams {
    // r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    static constexpr uint32_t P[8]  = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    // Montgomery radix R = 2^261 = (2^29)^9, see Fp::operator*
    static constexpr uint32_t R1[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
    static constexpr uint32_t R2[8] = {0x45b69bd4u, 0x38c2e14bu, 0x85883377u, 0x0ffedb18u, 0xabc6e54du, 0x7840f9f0u, 0x848b0f05u, 0x0a054a3eu};
    static constexpr uint32_t P29[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV29 = 0x0fffffffu;  // -p^-1 mod 2^29
    // k*r for k = -2..2 and 16r, 8r, 4r, 2r, r in "normal form" (see FqParams): lazy29.cuh
    static constexpr int32_t KP29[5][9] = {{536870910, 31499968, 55031646, 274652697, 351558393, 441024126, 514997629, 55030679, -6342813}, {268435455, 15749984, 295951279, 405761804, 175779196, 488947519, 525934270, 295950795, -3171407}, {0, 0, 0, 0, 0, 0, 0, 0, 0}, {268435457, 521120927, 240919632, 131109107, 361091715, 47923392, 10936641, 240920116, 3171406}, {2, 505370943, 481839265, 262218214, 185312518, 95846785, 21873282, 481840232, 6342812}};
    static constexpr int32_t POSP29[5][9] = {{16, 284871160, 96617743, 487132983, 408758323, 229903370, 174986257, 96625472, 50742503}, {8, 410871036, 316744327, 512001947, 204379161, 383387141, 87493128, 316748192, 25371251}, {4, 473870974, 426807619, 524436429, 370625036, 191693570, 43746564, 426809552, 12685625}, {2, 505370943, 481839265, 262218214, 185312518, 95846785, 21873282, 481840232, 6342812}, {268435457, 521120927, 240919632, 131109107, 361091715, 47923392, 10936641, 240920116, 3171406}};
};
struct FqParams {
    // q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    static constexpr uint32_t P[8]  = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R1[8] = {0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u, 0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};
    static constexpr uint32_t R2[8] = {0x659bac10u, 0xe1a2a074u, 0x5406005au, 0x63985586u, 0x2d3e2632u, 0xff54c580u, 0x34ea65a6u, 0x2a11a68cu};
    static constexpr uint32_t P29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV29 = 0x04866389u;
    // k*q for k = -2..2 in "normal form" (limbs 0..7 in [0, 2^29), signed top limb): lazy29.cuh
    static constexpr int32_t KP29[5][9] = {{252052850, 502742674, 119191905, 441825886, 351554831, 441024126, 514997629, 55030679, -6342813}, {126026425, 519806793, 59595952, 489348399, 175777415, 488947519, 525934270, 295950795, -3171407}, {0, 0, 0, 0, 0, 0, 0, 0, 0}, {410844487, 17064118, 477274959, 47522512, 361093496, 47923392, 10936641, 240920116, 3171406}, {284818062, 34128237, 417679006, 95045025, 185316080, 95846785, 21873282, 481840232, 6342812}};
    // 16p, 8p, 4p, 2p, p in the same form (store_exact's conditional subtractions)
    static constexpr int32_t POSP29[5][9] = {{131060848, 273025900, 120206576, 223489294, 408786817, 229903370, 174986257, 96625472, 50742503}, {65530424, 136512950, 60103288, 380180103, 204393408, 383387141, 87493128, 316748192, 25371251}, {32765212, 68256475, 298487100, 190090051, 370632160, 191693570, 43746564, 426809552, 12685625}, {284818062, 34128237, 417679006, 95045025, 185316080, 95846785, 21873282, 481840232, 6342812}, {410844487, 17064118, 477274959, 47522512, 361093496, 47923392, 10936641, 240920116, 3171406}};
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

    // ---- Montgomery multiplication, radix 2^29 product scanning -----------------------------
    // On gfx950 v_mad_u64_u32 (32x32+64 -> 64) issues at the same rate as a 64-bit add or an
    // add-with-carry (tools/ubench_valu.hip), so the cost of a 254-bit multiply is the NUMBER of
    // VALU instructions, not the number of multiplies.  With 9 limbs of 29 bits a whole column of
    // the schoolbook product plus the Montgomery correction (18 products < 2^58) fits one 64-bit
    // accumulator: every limb product is exactly one v_mad_u64_u32 accumulating in place, with no
    // carry handling inside a column -- 162 multiply-adds + ~60 shifts/masks instead of the
    // 136 + ~430 of the 32-bit CIOS form (1.5x-1.9x faster, tools/ubench_field.hip).
    static constexpr uint32_t M29 = 0x1fffffffu;
    ZK_HD static void to29(const uint32_t* x, uint32_t* o) {   // o[k] = bits [29k, 29k+29)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int bit = 29 * k, w = bit >> 5, s = bit & 31;
            uint32_t lo = x[w] >> s;
            if (s > 3 && w + 1 < 8) lo |= x[w + 1] << (32 - s);
            o[k] = lo & M29;
        }
    }
    ZK_HD static void from29(const uint32_t* a, uint32_t* x) {   // limbs < 2^29 -> 8 x 32
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int bit = 32 * w, k = bit / 29, s = bit - 29 * k;
            uint32_t v = a[k] >> s;
            const int have = 29 - s;
            if (have < 32 && k + 1 < 9) v |= a[k + 1] << have;
            if (have + 29 < 32 && k + 2 < 9) v |= a[k + 2] << (have + 29);
            x[w] = v;
        }
    }
    // r = a*b*2^-261 mod p (r < 2p), a, b, r as 9 x 29-bit limbs.  SQR: a == b, symmetric terms
    // are formed once with a doubled limb (45 instead of 81 products).
    template <bool SQR>
    ZK_HD static void mont29(const uint32_t* a, const uint32_t* b, uint32_t* r) {
        uint32_t m[9], a2[9];
        if (SQR) {
#pragma unroll
            for (int i = 0; i < 9; ++i) a2[i] = a[i] << 1;
        }
        // Independent accumulator chains per column (products of a*b split in two, m*p in a
        // third): a single running sum makes every v_mad_u64_u32 wait for the previous one, which is
        // latency-bound at the 2-4 waves/SIMD the curve kernels reach.
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const int lo = k < 9 ? 0 : k - 8, hi = k < 9 ? k : 8;
            uint64_t acc0 = carry, acc1 = 0, acc2 = 0;
            if (SQR) {
#pragma unroll
                for (int i = lo; i <= hi; ++i) {
                    const int j = k - i;
                    if (i < j) { if (i & 1) acc1 += (uint64_t)a2[i] * a[j]; else acc0 += (uint64_t)a2[i] * a[j]; }
                    else if (i == j) acc1 += (uint64_t)a[i] * a[i];
                }
            } else {
#pragma unroll
                for (int i = lo; i <= hi; ++i) { if (i & 1) acc1 += (uint64_t)a[i] * b[k - i]; else acc0 += (uint64_t)a[i] * b[k - i]; }
            }
#pragma unroll
            for (int i = lo; i <= hi; ++i)
                if (i < k || k >= 9) acc2 += (uint64_t)m[i] * PR::P29[k - i];
            uint64_t acc = acc0 + acc1 + acc2;
            if (k < 9) {
                m[k] = ((uint32_t)acc * PR::INV29) & M29;
                acc += (uint64_t)m[k] * PR::P29[0];
            } else {
                r[k - 9] = (uint32_t)acc & M29;
            }
            carry = acc >> 29;
        }
        uint64_t acc = carry;
        r[8] = (uint32_t)acc;
    }
    ZK_HD static Fp mul_inline(const Fp& a, const Fp& b) {
        uint32_t x[9], y[9], r[9];
        to29(a.l, x);
        to29(b.l, y);
        mont29<false>(x, y, r);
        Fp o;
        from29(r, o.l);
        o = reduce_once(o, 0);
        ZK_SCHED_FENCE();
        return o;
    }
    ZK_HD static Fp sqr_inline(const Fp& a) {
        uint32_t x[9], r[9];
        to29(a.l, x);
        mont29<true>(x, x, r);
        Fp o;
        from29(r, o.l);
        o = reduce_once(o, 0);
        ZK_SCHED_FENCE();
        return o;
    }
#if defined(ZK_MUL_OUTLINE) && defined(__HIP_DEVICE_COMPILE__)
    // one out-of-line body per translation unit: keeps the G2 loops inside the instruction cache
    __device__ __attribute__((noinline)) static Fp mul_outline(Fp a, Fp b) { return mul_inline(a, b); }
    __device__ __attribute__((noinline)) static Fp sqr_outline(Fp a) { return sqr_inline(a); }
    ZK_HD Fp operator*(const Fp& b) const { return mul_outline(*this, b); }
    ZK_HD Fp sqr() const { return sqr_outline(*this); }
#else
    ZK_HD Fp operator*(const Fp& b) const { return mul_inline(*this, b); }
    ZK_HD Fp sqr() const { return sqr_inline(*this); }
#endif

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
    // The same inverse by the binary extended Euclid for an odd modulus (Handbook of Applied Cryptography 14.61; kept as a second opinion for the tests): about
    // 1.4 x 254 rounds of shifts and subtractions on eight words, ~25 k instructions where Fermat's 255 squarings + 127
    // multiplications take ~130 k.  The trip count depends on the VALUE, so a caller with a secret operand blinds it first
    // (ec.cuh jac_to_affine_vartime: the Z coordinates that close a proof are multiplied by a fresh random factor), and this is
    // for single-lane uses -- the three inversions that close a proof -- not for whole waves, whose lanes would all wait for the
    // slowest one.
    ZK_HD Fp inv_euclid() const {
        if (is_zero()) return zero();
        uint32_t u[8], v[8], x1[8], x2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { u[i] = l[i]; v[i] = PR::P[i]; x1[i] = 0; x2[i] = 0; }
        x1[0] = 1;
        auto is_one = [](const uint32_t* a) {
            uint32_t r = a[0] ^ 1u;
#pragma unroll
            for (int i = 1; i < 8; ++i) r |= a[i];
            return r == 0;
        };
        auto halve = [](uint32_t* a, uint32_t* x) {   // a /= 2 (a even);  x /= 2 mod p  (x + p < 2^255 fits)
#pragma unroll
            for (int i = 0; i < 7; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
            a[7] >>= 1;
            if (x[0] & 1) {
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint64_t t = (uint64_t)x[i] + PR::P[i] + c;
                    x[i] = (uint32_t)t;
                    c = (uint32_t)(t >> 32);
                }
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
            x[7] >>= 1;
        };
        auto sub = [](uint32_t* a, const uint32_t* b) {   // a -= b, returns the borrow
            uint32_t br = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint64_t t = (uint64_t)a[i] - b[i] - br;
                a[i] = (uint32_t)t;
                br = (uint32_t)(t >> 63);
            }
            return br;
        };
        auto sub_mod = [&](uint32_t* a, const uint32_t* b) {   // a = a - b mod p
            if (sub(a, b)) {
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    uint64_t t = (uint64_t)a[i] + PR::P[i] + c;
                    a[i] = (uint32_t)t;
                    c = (uint32_t)(t >> 32);
                }
            }
        };
        auto geq = [](const uint32_t* a, const uint32_t* b) {
            for (int i = 7; i >= 0; --i)
                if (a[i] != b[i]) return a[i] > b[i];
            return true;
        };
        while (!is_one(u) && !is_one(v)) {
            while (!(u[0] & 1)) halve(u, x1);
            while (!(v[0] & 1)) halve(v, x2);
            if (geq(u, v)) { sub(u, v); sub_mod(x1, x2); }
            else { sub(v, u); sub_mod(x2, x1); }
        }
        // y = (a R)^-1 as an integer; a^-1 R = y R^2 = montmul(montmul(y, R^2), R^2)
        Fp y;
        const bool first = is_one(u);
#pragma unroll
        for (int i = 0; i < 8; ++i) y.l[i] = first ? x1[i] : x2[i];
        return y * r2() * r2();
    }

    // The same inverse by Bernstein-Yang division steps in batches of 30 (the variable-time form libsecp256k1 calls modinv32_var,
    // restated): the low 30 bits of (f, g) = (p, a) decide 30 steps at a time as a 2 x 2 matrix of 31-bit integers, which is then
    // applied to the 9 x 30-bit signed limbs of f, g (exact division by 2^30) and of the Bezout coefficients d, e (division mod p).
    // ~19 batches of ~500 instructions where the bit-by-bit Euclid above runs ~35 k: the inversion that closes a proof takes
    // 0.03 ms instead of 0.13 on one lane (tools/ubench_assemble.hip).  Variable time like inv_euclid: same blinding rule.
    ZK_HD Fp inv_vartime() const { return inv_divsteps(); }
    ZK_HD static constexpr uint32_t p_inv30() {   // p^-1 mod 2^30 (Newton's iteration from p mod 2^32, p odd)
        uint32_t x = PR::P[0];
        for (int i = 0; i < 5; ++i) x *= 2u - PR::P[0] * x;
        return x & 0x3fffffffu;
    }
    ZK_HD Fp inv_divsteps() const {
        if (is_zero()) return zero();
        constexpr int32_t M30 = 0x3fffffff;
        constexpr uint32_t PINV = p_inv30();
        int32_t f[9], g[9], d[9], e[9], md[9];
        {   // 8 x 32 -> 9 x 30
            auto split = [](const uint32_t* w, int32_t* o) {
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const int bit = 30 * i, k = bit >> 5, sh = bit & 31;
                    uint64_t v = (uint64_t)w[k] >> sh;
                    if (sh > 2 && k + 1 < 8) v |= (uint64_t)w[k + 1] << (32 - sh);
                    o[i] = (int32_t)(v & (uint64_t)M30);
                }
            };
            uint32_t pw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pw[i] = PR::P[i];
            split(pw, md);
            split(l, g);
#pragma unroll
            for (int i = 0; i < 9; ++i) { f[i] = md[i]; d[i] = 0; e[i] = 0; }
            e[0] = 1;
        }
        int32_t eta = -1;
        for (int it = 0; it < 26; ++it) {
            // 30 division steps on the low words
            uint32_t u = 1, v = 0, q = 0, r = 1;
            uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
            int i = 30;
            for (;;) {
                const uint32_t gz = g0 | (0xffffffffu << i);       // sentinel: at most i zeros
                const int zeros = __builtin_ctz(gz);
                g0 >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
                if (i == 0) break;
                if (eta < 0) {
                    eta = -eta;
                    uint32_t t = f0; f0 = g0; g0 = 0u - t;
                    t = u; u = q; q = 0u - t;
                    t = v; v = r; r = 0u - t;
                }
                g0 += f0; q += u; r += v;                          // g odd, f odd: the sum is even
            }
            const int64_t U = (int32_t)u, V = (int32_t)v, Q = (int32_t)q, Rr = (int32_t)r;
            {   // (d, e) <- (U d + V e, Q d + R e) / 2^30 mod p, kept in (-2p, p)
                const int32_t sd = d[8] >> 31, se = e[8] >> 31;
                int32_t m_d = ((int32_t)U & sd) + ((int32_t)V & se), m_e = ((int32_t)Q & sd) + ((int32_t)Rr & se);
                int64_t cd = U * d[0] + V * e[0], ce = Q * d[0] + Rr * e[0];
                m_d -= (int32_t)((PINV * (uint32_t)cd + (uint32_t)m_d) & (uint32_t)M30);
                m_e -= (int32_t)((PINV * (uint32_t)ce + (uint32_t)m_e) & (uint32_t)M30);
                cd += (int64_t)md[0] * m_d; ce += (int64_t)md[0] * m_e;
                cd >>= 30; ce >>= 30;
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    cd += U * d[k] + V * e[k] + (int64_t)md[k] * m_d;
                    ce += Q * d[k] + Rr * e[k] + (int64_t)md[k] * m_e;
                    d[k - 1] = (int32_t)cd & M30; e[k - 1] = (int32_t)ce & M30;
                    cd >>= 30; ce >>= 30;
                }
                d[8] = (int32_t)cd; e[8] = (int32_t)ce;
            }
            {   // (f, g) <- (U f + V g, Q f + R g) / 2^30, exactly
                int64_t cf = U * f[0] + V * g[0], cg = Q * f[0] + Rr * g[0];
                cf >>= 30; cg >>= 30;
#pragma unroll
                for (int k = 1; k < 9; ++k) {
                    cf += U * f[k] + V * g[k];
                    cg += Q * f[k] + Rr * g[k];
                    f[k - 1] = (int32_t)cf & M30; g[k - 1] = (int32_t)cg & M30;
                    cf >>= 30; cg >>= 30;
                }
                f[8] = (int32_t)cf; g[8] = (int32_t)cg;
            }
            int32_t nz = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) nz |= g[k];
            if (nz == 0) break;
        }
        // f = +-1; the inverse of the stored integer is sign(f) d, brought to [0, p)
        const int32_t fneg = f[8] >> 31;
        auto add_p_if = [&](int32_t mask) {
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] += md[k] & mask;
        };
        auto carry = [&]() {
            int32_t c = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int32_t t = d[k] + c; d[k] = t & M30; c = t >> 30; }
            d[8] += c;
        };
        add_p_if(d[8] >> 31);                       // (-2p, p) -> (-p, p)
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = (d[k] ^ fneg) - fneg;
        carry();
        add_p_if(d[8] >> 31);                       // -> [0, p)
        carry();
        Fp y;
#pragma unroll
        for (int i = 0; i < 8; ++i) {               // 9 x 30 -> 8 x 32
            const int bit = 32 * i, k = bit / 30, sh = bit % 30;
            uint64_t v = (uint64_t)(uint32_t)d[k] >> sh;
            v |= (uint64_t)(uint32_t)d[k + 1] << (30 - sh);
            if (k + 2 < 9) v |= (uint64_t)(uint32_t)d[k + 2] << (60 - sh);
            y.l[i] = (uint32_t)v;
        }
        // y = (a R)^-1 as an integer, as in inv_euclid
        return y * r2() * r2();
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
    ZK_HD Fq2 inv_vartime() const {
        Fq d = (c0.sqr() + c1.sqr()).inv_divsteps();
        return Fq2{c0 * d, -(c1 * d)};
    }
    ZK_HD static Fq2 from_canonical(const Fq2& x) { return Fq2{Fq::from_canonical(x.c0), Fq::from_canonical(x.c1)}; }
    ZK_HD Fq2 to_canonical() const { return Fq2{c0.to_canonical(), c1.to_canonical()}; }
    ZK_HD bool raw_in_range() const { return c0.raw_in_range() && c1.raw_in_range(); }
};

}  // namespace zk
