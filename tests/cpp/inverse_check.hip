// tests/cpp/inverse_check.hip -- the three inversions of ff.cuh against each other on the HOST (the field code is __host__ __device__):
// Fermat (inv), binary extended Euclid (inv_euclid), division steps in batches of 30 (inv_divsteps = inv_vartime, which closes every
// proof), plus x * x^-1 == 1; small values, powers of two, both Montgomery and raw bit patterns, 20000 random values per field.
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -I zksnark_rs_amd/csrc tests/cpp/inverse_check.hip -o inverse_check && ./inverse_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include "ff.cuh"
using namespace zk;
template <class F> int check(const char* name) {
    std::mt19937_64 rng(7);
    int bad = 0, n = 0;
    auto one = [&](F x) {
        F a = x.inv_euclid(), b = x.inv_divsteps(), c = x.inv();
        bool ok = (a == b) && (b == c);
        if (!x.is_zero()) ok = ok && ((x * b) == F::one());
        if (!ok) { if (bad < 5) { printf("%s mismatch at case %d\n", name, n); } ++bad; }
        ++n;
    };
    F z = F::zero(); one(z); one(F::one()); one(-F::one());
    for (uint32_t k = 2; k < 200; ++k) { F x = F::zero(); x.l[0] = k; one(x); one(F::from_canonical(x)); one(-F::from_canonical(x)); }
    for (int b = 1; b < 254; ++b) { F x = F::zero(); x.l[b >> 5] = 1u << (b & 31); one(x); }
    for (int i = 0; i < 20000; ++i) { F x; for (int k = 0; k < 8; ++k) x.l[k] = (uint32_t)rng(); x.l[7] &= 0x0fffffff; one(x); }
    printf("%s: %d cases, %d bad\n", name, n, bad);
    return bad;
}
int main() { return check<Fr>("Fr") | check<Fq>("Fq"); }
