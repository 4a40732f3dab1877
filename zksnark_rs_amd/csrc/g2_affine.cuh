// g2_affine.cuh -- sums of PAIRS of G2 points in affine coordinates with one shared inversion per workgroup (Montgomery's trick).
//
// Replaces, for the first rounds of a bucket's sum, the sequential G2 additions of Sum / Add for G2Local
// (/root/reference/src/groth16/fr.rs:175-223) inside exp_encrypted_g2's inner product (mod.rs:254-262): same group elements.
//
// An affine addition is lambda = (y2 - y1) / (x2 - x1), x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1: over Fq2 two products,
// one square and ONE inversion -- against 29 base-field multiplications for the XYZZ mixed addition of the accumulation loop.  The
// inversions of a whole workgroup are shared:
//   * a lane owns the pairs i = t, t + lanes, t + 2 lanes, ... (consecutive lanes read consecutive pairs: coalesced) and walks them
//     twice.  Forward: the denominator d_k, its norm n_k = d_k0^2 + d_k1^2 in Fq (an inverse in Fq2 is conj(d) / n), the running
//     product n_0 .. n_{k-1} parked in HBM (36 B per pair).
//   * the lanes' totals are multiplied up a binary tree in LDS, ONE lane inverts the root (division steps, ff.cuh inv_vartime: ~10 k instructions per
//     workgroup; the operands are differences of public CRS coordinates), and the inverses come back down the tree: two multiplications per node.
//   * backward: 1 / n_k = (inverse of n_0 .. n_k) x (parked n_0 .. n_{k-1}); the points are read again, the sum is stored.
// Exceptional pairs keep the walk uniform: a point at infinity or P + (-P) take the denominator 1, P + P takes 2 y with the
// numerator 3 x^2.  Points are canonical Montgomery values in HBM (Aff<Fq2>, 128 B; infinity = (0, 0)), so equality is exact.
#pragma once
#include "ec.cuh"
#include "lazy29.cuh"
#include "fr_tile.cuh"

namespace zk {

typedef FpR<FqParams> FqL;
typedef Fp2R<FqParams> Fq2L;

#ifndef ZK_AFF_THREADS
#define ZK_AFF_THREADS 256
#endif
#ifndef ZK_AFF_WGS_PER_CU
#define ZK_AFF_WGS_PER_CU 3
#endif
constexpr int AFF_THREADS = ZK_AFF_THREADS;   // lanes that share one inversion
constexpr uint32_t AFF_PAD = 0xffffffffu;   // entry of the sorted list that stands for the point at infinity

// pair sources: get(i, a, b) = the two points of pair i
struct PairList {      // a list of points: (2i, 2i + 1)
    const Aff<Fq2>* p;
    __device__ __forceinline__ void get(uint32_t i, Aff<Fq2>& a, Aff<Fq2>& b) const { a = p[2 * (size_t)i]; b = p[2 * (size_t)i + 1]; }
};
struct PairArrays {    // two arrays: a[i] + b[i]
    const Aff<Fq2>* pa;
    const Aff<Fq2>* pb;
    __device__ __forceinline__ void get(uint32_t i, Aff<Fq2>& a, Aff<Fq2>& b) const { a = pa[i]; b = pb[i]; }
};
struct PairTable {     // entries of an inner product's sorted list: (table index << 1 | negate), AFF_PAD = infinity
    const Aff<Fq2>* table;
    const uint32_t* sorted;
    __device__ __forceinline__ Aff<Fq2> one(uint32_t e) const {
        if (e == AFF_PAD) return Aff<Fq2>::infinity();
        Aff<Fq2> q = table[e >> 1];
        if (e & 1) q.y = -q.y;
        return q;
    }
    __device__ __forceinline__ void get(uint32_t i, Aff<Fq2>& a, Aff<Fq2>& b) const {
        const uint2 e = reinterpret_cast<const uint2*>(sorted)[i];
        a = one(e.x);
        b = one(e.y);
    }
};

// kind of a pair: 0 general, 1 doubling, 2 result = a, 3 result = b, 4 result = infinity
__device__ __forceinline__ int aff_denominator(const Aff<Fq2>& a, const Aff<Fq2>& b, Fq2L& d) {
    const FqL one = FqL::load(Fq::one());
    d = Fq2L{one, FqL::load(Fq::zero())};
    const bool ai = a.is_inf(), bi = b.is_inf();
    if (ai | bi) return ai ? (bi ? 4 : 3) : 2;
    if (a.x == b.x) {
        if (!(a.y == b.y) || a.y.is_zero()) return 4;
        const Fq2L y = Fq2L::load(a.y);
        d = (y + y).norm();   // both factors of the norm are d: limbs below 2^29
        return 1;
    }
    d = Fq2L::load(b.x) - Fq2L::load(a.x);
    return 0;
}
__device__ __forceinline__ FqL aff_norm(const Fq2L& d) { return FqL::mont_sum(d.c0, d.c0, d.c1, d.c1); }

__device__ __forceinline__ void aff_park(int32_t* __restrict__ prefix, size_t slot, size_t lanes, const FqL& v) {
#pragma unroll
    for (int l = 0; l < 9; ++l) prefix[(slot * 9 + l) * lanes] = v.v[l];
}
__device__ __forceinline__ FqL aff_unpark(const int32_t* __restrict__ prefix, size_t slot, size_t lanes) {
    FqL v;
#pragma unroll
    for (int l = 0; l < 9; ++l) v.v[l] = prefix[(slot * 9 + l) * lanes];
    return v;
}

// dst[i] = first + second point of pair i, i < pairs.  prefix: 9 x pairs (rounded up to the grid) int32 of scratch.
// total (may be null): the length of the list the pairs come from lives on the device -- pairs = min(pairs, *total >> shift).
template <class Src>
__global__ __launch_bounds__(AFF_THREADS) void k_g2_pair_sums(Src src, uint32_t pairs, const uint32_t* __restrict__ total, int shift, int32_t* __restrict__ prefix,
                                                              Aff<Fq2>* __restrict__ dst) {
    __shared__ int32_t tree[2 * AFF_THREADS][9];
    if (total) pairs = min(pairs, *total >> shift);
    const uint32_t lanes = gridDim.x * AFF_THREADS, t = blockIdx.x * AFF_THREADS + threadIdx.x;
    int32_t* const park = prefix + t;
    // ---- forward: running products of the norms ----
    FqL run = FqL::load(Fq::one());
    uint32_t steps = 0;
    for (uint32_t i = t; i < pairs; i += lanes, ++steps) {
        Aff<Fq2> a, b;
        src.get(i, a, b);
        Fq2L d;
        aff_denominator(a, b, d);
        aff_park(park, steps, lanes, run);
        run = run * aff_norm(d);
    }
    // ---- one inversion per workgroup: products up a binary tree, inverses down ----
    {
        const int me = AFF_THREADS + threadIdx.x;
#pragma unroll
        for (int l = 0; l < 9; ++l) tree[me][l] = run.v[l];
        __syncthreads();
        for (int s = AFF_THREADS / 2; s >= 1; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const int node = s + threadIdx.x;
                FqL x, y;
#pragma unroll
                for (int l = 0; l < 9; ++l) { x.v[l] = tree[2 * node][l]; y.v[l] = tree[2 * node + 1][l]; }
                const FqL z = x * y;
#pragma unroll
                for (int l = 0; l < 9; ++l) tree[node][l] = z.v[l];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            FqL r;
#pragma unroll
            for (int l = 0; l < 9; ++l) r.v[l] = tree[1][l];
            const FqL ri = FqL::load(lazy_store_exact<FqParams>(r).inv_vartime());
#pragma unroll
            for (int l = 0; l < 9; ++l) tree[1][l] = ri.v[l];
        }
        __syncthreads();
        for (int s = 1; s < AFF_THREADS; s <<= 1) {
            if ((int)threadIdx.x < s) {
                const int node = s + threadIdx.x;
                FqL inv, x, y;
#pragma unroll
                for (int l = 0; l < 9; ++l) { inv.v[l] = tree[node][l]; x.v[l] = tree[2 * node][l]; y.v[l] = tree[2 * node + 1][l]; }
                const FqL ix = inv * y, iy = inv * x;
#pragma unroll
                for (int l = 0; l < 9; ++l) { tree[2 * node][l] = ix.v[l]; tree[2 * node + 1][l] = iy.v[l]; }
            }
            __syncthreads();
        }
#pragma unroll
        for (int l = 0; l < 9; ++l) run.v[l] = tree[me][l];   // the inverse of this lane's product
    }
    // ---- backward: the sums ----
    for (uint32_t k = steps; k-- > 0;) {
        const uint32_t i = t + k * lanes;
        Aff<Fq2> a, b;
        src.get(i, a, b);
        Fq2L d;
        const int kind = aff_denominator(a, b, d);
        const FqL n = aff_norm(d);
        const FqL inv_n = run * aff_unpark(park, k, lanes);
        run = run * n;
        Aff<Fq2> out;
        if (kind >= 2) {
            out = kind == 2 ? a : (kind == 3 ? b : Aff<Fq2>::infinity());
        } else {
            const Fq2L inv_d{d.c0 * inv_n, (d.c1 * inv_n).neg()};
            const Fq2L ax = Fq2L::load(a.x), ay = Fq2L::load(a.y);
            Fq2L num;
            if (kind == 1) {
                const Fq2L xx = ax.sqr();
                num = (xx + xx + xx).norm();
            } else {
                num = Fq2L::load(b.y) - ay;
            }
            const Fq2L lam = num * inv_d;
            const Fq2L x3 = lam.sqr() - ax - Fq2L::load(b.x);
            const Fq2L y3 = lam * (ax - x3).norm() - ay;
            out.x = Fq2{lazy_store_exact<FqParams>(x3.c0), lazy_store_exact<FqParams>(x3.c1)};
            out.y = Fq2{lazy_store_exact<FqParams>(y3.c0), lazy_store_exact<FqParams>(y3.c1)};
        }
        dst[i] = out;
    }
}

// lanes for `pairs` pairs: ZK_AFF_WGS_PER_CU workgroups per compute unit, fewer when the lanes would walk less than 8 pairs each
inline unsigned aff_grid(size_t pairs, int cu_count) {
    const size_t want = (pairs + (size_t)AFF_THREADS * 8 - 1) / ((size_t)AFF_THREADS * 8);
    return (unsigned)std::max<size_t>(1, std::min<size_t>((size_t)cu_count * ZK_AFF_WGS_PER_CU, want));
}
inline size_t aff_prefix_words(size_t pairs, int cu_count) {
    const size_t lanes = (size_t)aff_grid(pairs, cu_count) * AFF_THREADS;
    return ((pairs + lanes - 1) / lanes) * lanes * 9;
}

}  // namespace zk
