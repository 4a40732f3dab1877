// oracle/z251.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// The reference's toy field Z/251 (/root/reference/src/field/z251.rs:1-97) and its use as its
// own "pairing group" test double (/root/reference/src/groth16/mod.rs:329-374,
// /root/reference/src/encryption.rs:20-35).  Kept in the oracle only: every polynomial
// known-answer test of the reference is stated over Z251.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

namespace orc {

struct Z251 {
    uint8_t inner;

    static Z251 zero() { return Z251{0}; }
    static Z251 one() { return Z251{1}; }
    // z251.rs:77-82: From<usize> asserts n < 251
    static Z251 from_usize(size_t n) {
        if (n >= 251) throw std::out_of_range("assertion failed: n < 251");
        return Z251{(uint8_t)n};
    }
    // z251.rs:90-96: usize::from_str then From<usize>
    static bool from_str(const std::string& s, Z251& out) {
        if (s.empty()) return false;
        size_t v = 0;
        for (char ch : s) {
            if (ch < '0' || ch > '9') return false;
            v = v * 10 + (size_t)(ch - '0');
            if (v > 1000000) return false;
        }
        out = from_usize(v);
        return true;
    }
    bool operator==(const Z251& o) const { return inner == o.inner; }  // derived PartialEq on `inner`
    bool operator!=(const Z251& o) const { return inner != o.inner; }
    Z251 operator+(const Z251& o) const { return Z251{(uint8_t)(((uint16_t)inner + o.inner) % 251)}; }
    // z251.rs:20-28: `251 - inner` is stored unreduced, so -0 == Z251{251}; every Add/Mul
    // reduces again, so the quirk is only observable through `==`.
    Z251 operator-() const { return Z251{(uint8_t)(251 - inner)}; }
    Z251 operator-(const Z251& o) const { return *this + (-o); }
    Z251 operator*(const Z251& o) const { return Z251{(uint8_t)(((uint16_t)inner * o.inner) % 251)}; }
    Z251 inv() const {
        // z251.rs:50-61 via ext_euc_alg on isize; ext_euc_alg(0,251) returns inverse 0 there, so
        // x/0 == 0 over Z251 in the reference (no panic).  Mirror that.
        int a = inner % 251, r0 = a, r1 = 251, s0 = 1, s1 = 0;
        while (r1 != 0) {
            int q = r0 / r1;
            int r = r0 - q * r1, s = s0 - q * s1;
            r0 = r1; r1 = r; s0 = s1; s1 = s;
        }
        int iv = s0;
        while (iv < 0) iv += 251;
        return Z251{(uint8_t)(iv % 251)};
    }
    Z251 operator/(const Z251& o) const { return *this * o.inv(); }
};

}  // namespace orc
