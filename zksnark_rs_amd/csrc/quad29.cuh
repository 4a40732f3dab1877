// quad29.cuh -- point additions and doublings shared by the FOUR lanes of a quad (the reduction tail of a small MSM).
//
// The tail of an MSM (merge, folds, weighted sum; msm_impl.hpp) is a chain of dependent point additions, each executed by lanes that
// have nothing else to do: a lone proof of 2^16 gates spends 1.5 ms of its 2.55 ms in the ~40 dependent Fq2 additions behind the G2
// accumulation, one wave per SIMD issuing one instruction every ~5.6 cycles (profiles/r3_timeline_lone_2p16.txt).  An XYZZ addition is
// 12M + 2S, but its dependency depth is 4 multiplications: here the four lanes of a quad hold the SAME two points, each multiplies a
// different pair of coordinates, and the products travel between the lanes by DPP quad permutes (v_mov_b32 quad_perm: no LDS, no
// barrier).  4 multiplications + ~650 moves / selects per addition instead of 14 multiplications: ~3x shorter chains for 4x the lanes,
// which a tail has to spare.  Every lane ends with the complete sum, so kernels are written as if one lane did the work; lane
// `role == 0` of a quad stores.  Used when an MSM has few buckets (msm_run: `quad`); large products keep one lane per addition
// (their tails are 5 % of a proof's instructions and would grow by a quarter).
//
// Bounds follow lazy29.cuh: stored coordinates are normal forms, products take normal forms or differences of two of them.
#pragma once
#include "lazy29.cuh"

namespace zk {

// the value lane K of the quad holds
template <int K>
__device__ __forceinline__ int32_t quad_get(int32_t v) {
    int32_t r = __builtin_amdgcn_update_dpp(v, v, K * 0x55, 0xf, 0xf, true);   // v_mov_b32 quad_perm:[K,K,K,K]
    // The move stays a move.  Left to itself the compiler folds it into the instruction that uses the result (GCNDPPCombine), and the
    // commuted form it builds for `get<0>(t) - get<1>(t)` -- v_subrev_u32_dpp -- gave every lane its OWN t as the subtrahend on
    // gfx950 / ROCm 7.2 (tools/quad_check.hip: Y3 right in lane 1 only; right everywhere with -mllvm -amdgpu-dpp-combine=false).
    asm volatile("" : "+v"(r));
    return r;
}
template <int K, class PR>
__device__ __forceinline__ FpR<PR> quad_get(const FpR<PR>& x) {
    FpR<PR> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v[i] = quad_get<K>(x.v[i]);
    return r;
}
template <int K, class PR>
__device__ __forceinline__ Fp2R<PR> quad_get(const Fp2R<PR>& x) { return Fp2R<PR>{quad_get<K>(x.c0), quad_get<K>(x.c1)}; }

// operand of this lane: a / b / c / d for role 0 / 1 / 2 / 3
template <class PR>
__device__ __forceinline__ FpR<PR> quad_sel(int role, const FpR<PR>& a, const FpR<PR>& b, const FpR<PR>& c, const FpR<PR>& d) {
    // the four candidates are read BEFORE the selection: written as a conditional expression over the members, the loads are sunk
    // behind a select of ADDRESSES, which keeps both points in scratch
    const bool r0 = role == 0, r1 = role == 1, r2 = role == 2;
    FpR<PR> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t av = a.v[i], bv = b.v[i], cv = c.v[i], dv = d.v[i];
        const int32_t hi = r2 ? cv : dv, lo = r0 ? av : bv;
        r.v[i] = (r0 | r1) ? lo : hi;
    }
    return r;
}
template <class PR>
__device__ __forceinline__ Fp2R<PR> quad_sel(int role, const Fp2R<PR>& a, const Fp2R<PR>& b, const Fp2R<PR>& c, const Fp2R<PR>& d) {
    return Fp2R<PR>{quad_sel(role, a.c0, b.c0, c.c0, d.c0), quad_sel(role, a.c1, b.c1, c.c1, d.c1)};
}

// 2 p (dbl-2008-s-1, a = 0) in three rounds of multiplications:
//   U = 2 Y, V = U^2, W = U V, S = X V, M = 3 X^2, X3 = M^2 - 2 S, Y3 = M (S - X3) - W Y, ZZ3 = V ZZ, ZZZ3 = W ZZZ
// The curves have odd order (G2: checked at upload), so Y != 0 for every finite point.
template <class L>
__device__ __forceinline__ XyzzR<L> quad_dbl_xyzz(const XyzzR<L>& p, int role) {
    if (p.inf) return p;
    const L U = (p.Y + p.Y).norm();
    // round 1: V = U U | XX = X X | -- | --
    const L t1 = quad_sel(role, U, p.X, U, p.X) * quad_sel(role, U, p.X, U, p.X);
    const L V = quad_get<0>(t1), XX = quad_get<1>(t1);
    const L M = (XX + XX + XX).norm();
    // round 2: W = U V | S = X V | MM = M M | ZZ3 = ZZ V
    const L t2 = quad_sel(role, U, p.X, M, p.ZZ) * quad_sel(role, V, V, M, V);
    const L W = quad_get<0>(t2), S = quad_get<1>(t2), MM = quad_get<2>(t2);
    const L X3 = (MM - S - S).norm();
    // round 3: M (S - X3) | W Y | ZZZ3 = W ZZZ | --
    const L t3 = quad_sel(role, M, W, W, W) * quad_sel(role, S - X3, p.Y, p.ZZZ, p.Y);
    XyzzR<L> r;
    r.inf = false;
    r.X = X3;
    r.Y = (quad_get<0>(t3) - quad_get<1>(t3)).norm();
    r.ZZ = quad_get<3>(t2);
    r.ZZZ = quad_get<2>(t3);
    return r;
}

// p + q (add-2008-s) in four rounds:
//   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1, P = U2 - U1, R = S2 - S1, PP = P^2, PPP = P PP, Q = U1 PP,
//   X3 = R^2 - PPP - 2 Q, Y3 = R (Q - X3) - S1 PPP, ZZ3 = ZZ1 ZZ2 PP, ZZZ3 = ZZZ1 ZZZ2 PPP
template <class L>
__device__ __forceinline__ XyzzR<L> quad_add_xyzz(const XyzzR<L>& p, const XyzzR<L>& q, int role) {
    if (p.inf) return q;
    if (q.inf) return p;
    // round 1: U1 | U2 | S1 | S2
    const L t1 = quad_sel(role, p.X, q.X, p.Y, q.Y) * quad_sel(role, q.ZZ, p.ZZ, q.ZZZ, p.ZZZ);
    const L U1 = quad_get<0>(t1), S1 = quad_get<2>(t1);
    const L P = quad_get<1>(t1) - U1, R = quad_get<3>(t1) - S1;      // differences of two normal forms
    // round 2: PP = P P | RR = R R | ZZ1 ZZ2 | ZZZ1 ZZZ2
    const L t2 = quad_sel(role, P, R, p.ZZ, p.ZZZ) * quad_sel(role, P, R, q.ZZ, q.ZZZ);
    const L PP = quad_get<0>(t2), RR = quad_get<1>(t2);
    if (PP.is_zero_mod_p()) {                // same x coordinate (the same answer in all four lanes)
        if (RR.is_zero_mod_p()) return quad_dbl_xyzz(p, role);
        XyzzR<L> r = p;
        r.inf = true;
        return r;
    }
    // round 3: PPP = P PP | Q = U1 PP | ZZ3 = (ZZ1 ZZ2) PP | T = (ZZZ1 ZZZ2) PP
    const L t3 = quad_sel(role, P, U1, t2, t2) * PP;
    const L PPP = quad_get<0>(t3), Q = quad_get<1>(t3);
    const L X3 = (RR - PPP - (Q + Q)).norm();
    // round 4: R (Q - X3) | S1 PPP | -- | ZZZ3 = T P
    const L t4 = quad_sel(role, R, S1, P, t3) * quad_sel(role, Q - X3, PPP, P, P);
    XyzzR<L> r;
    r.inf = false;
    r.X = X3;
    r.Y = (quad_get<0>(t4) - quad_get<1>(t4)).norm();
    r.ZZ = quad_get<2>(t3);
    r.ZZZ = quad_get<3>(t4);
    return r;
}

// k p for a small k (the same k in the four lanes): double-and-add from the top bit
template <class L>
__device__ __forceinline__ XyzzR<L> quad_mul_small_xyzz(const XyzzR<L>& p, uint32_t k, int role) {
    XyzzR<L> acc = p;
    acc.inf = true;
    if (p.inf) return acc;
    for (int i = 31 - __builtin_clz(k | 1); i >= 0; --i) {
        acc = quad_dbl_xyzz(acc, role);
        if ((k >> i) & 1) acc = quad_add_xyzz(acc, p, role);
    }
    return acc;
}

}  // namespace zk
