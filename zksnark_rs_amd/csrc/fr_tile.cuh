// fr_tile.cuh -- reductions of field values held in the multiplier's own radix (lazy29.cuh): shared by the transform tiles
// (ntt.hip), the interpolation tree (interp.hip) and the affine pair sums of the G2 inner product (g2_affine.cuh).
#pragma once
#include "lazy29.cuh"

namespace zk {

typedef FpR<FrParams> FrL;

// value - q p with q ~ floor(value / p) estimated from the top limb: any limbs within int32 and
// |value| < 2^9 p in, normal form with value in (-p - eps, 2p) out (eps = 2^-12 p: the low limbs of q p).
template <class PR>
__device__ __forceinline__ FpR<PR> lazy_reduce(const FpR<PR>& a) {
    typedef FpR<PR> FrL;
    typedef PR FrParams;
    const int32_t q = (int32_t)floorf((float)a.v[8] * (1.0f / (float)FrParams::P29[8]));
    FrL r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int64_t t = (int64_t)a.v[i] - (int64_t)q * (int32_t)FrParams::P29[i] + c;
        r.v[i] = (int32_t)((uint32_t)t & (uint32_t)FrL::M29);
        c = t >> 29;
    }
    r.v[8] = a.v[8] - q * (int32_t)FrParams::P29[8] + (int32_t)c;
    return r;
}
__device__ __forceinline__ FrL fr_reduce(const FrL& a) { return lazy_reduce<FrParams>(a); }
// canonical 8 x 32 form of any tile value
template <class PR>
__device__ __forceinline__ Fp<PR> lazy_store_exact(const FpR<PR>& a) {
    typedef FpR<PR> FrL;
    typedef PR FrParams;
    typedef Fp<PR> Fr;
    const FrL t = lazy_reduce<PR>(a);   // (-p - eps, 2p): one of t + p, t, t - p is the residue
    FrL d, s;
    int32_t bd = 0, cs = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int32_t x = t.v[i] - (int32_t)FrParams::P29[i] + bd;
        d.v[i] = x & FrL::M29;
        bd = x >> 29;
        int32_t y = t.v[i] + (int32_t)FrParams::P29[i] + cs;
        s.v[i] = y & FrL::M29;
        cs = y >> 29;
    }
    d.v[8] = t.v[8] - (int32_t)FrParams::P29[8] + bd;
    s.v[8] = t.v[8] + (int32_t)FrParams::P29[8] + cs;
    const bool neg = t.v[8] < 0, ge = d.v[8] >= 0;   // normal form: the sign of the value is the sign of the top limb
    uint32_t u[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) u[i] = (uint32_t)(neg ? s.v[i] : (ge ? d.v[i] : t.v[i]));
    Fr o;
    Fr::from29(u, o.l);
    return o;
}
__device__ __forceinline__ Fr fr_store_exact(const FrL& a) { return lazy_store_exact<FrParams>(a); }

}  // namespace zk
