// gbasis.hip -- the change of basis of an uploaded CRS in O(n log^2 n) group operations: the TRANSPOSE of interp.hip's interpolation,
// run over curve points.
//
// groth16::prove takes any (&SigmaG1, &SigmaG2) setup emitted (/root/reference/src/groth16/mod.rs:172-194, 213-217), i.e. the powers
// [x^i]; the integer-roots form (aproots.hip) multiplies with [L_k(x)] = sum_i c_{k,i} [x^i], where c_{k,.} are the coefficients of the
// Lagrange polynomial of node k.  Interpolation IS the linear map M: values -> coefficients with M[i][k] = c_{k,i}, so the wanted points
// are y = M^T G: the same straight-line program read backwards (Tellegen), with every scalar multiplication-by-a-constant turned into a
// point multiplied by that constant.  interp.hip's program is: bottom blocks (a 64 x 64 matrix per 64 nodes), then per level
//     parent = iDIT( DIF(pad(left)) . N_right + DIF(pad(right)) . N_left ),
// and its transpose, level by level from the root down:
//     u = DIF^-1(parent)            -- (F^-1 P)^T = P F^-1: decimation in frequency with the inverse twiddles, natural -> bit-reversed
//     left' = N_right . u / 2s,  right' = N_left . u / 2s          (2s multiplications of a point by a stored scalar per node)
//     left = first s of DIT(left'),  right = first s of DIT(right')   -- (P F)^T = F P, bit-reversed -> natural; pad^T = truncation
// and at the bottom y_k = sum_i q[block][k][i] g_i.  Same per-root-set tables as the prover's interpolation (InterpTree), nothing new is
// precomputed.  A butterfly is one point multiplied by a 254-bit twiddle: ~1.5 n log^2 n / 2 of them -- 3 x 10^8 at 2^20 gates, seconds
// per array, against the n^2 = 10^12 terms of basis.hip's inner products (hours).
#include <vector>
#include "pipeline.hpp"
#include "interp.hpp"
#include "lazy29.cuh"

namespace zk {

constexpr int GB = INTERP_BLOCK;

// data[i] = bases[i] for i < n, infinity beyond
template <class F>
__global__ void k_gb_load(const Aff<F>* __restrict__ bases, size_t n, size_t npad, Jac<F>* __restrict__ data) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) data[i] = i < n ? Jac<F>::from_affine(bases[i]) : Jac<F>::infinity();
}

// ---- k P with the curve's endomorphism (GLV) ------------------------------------------------------------------------------------
// phi(x, y) = (beta x, y) with beta^3 = 1 in Fq is multiplication by lambda (lambda^2 + lambda + 1 = 0 mod r) on G1, and -- with the other
// cube root as beta -- on the order-r subgroup of the twist (both checked numerically against k P: tools/glv_constants.py).  A scalar
// splits as k = k1 + k2 lambda with |k1|, |k2| < 2^128 (Babai rounding against the lattice basis (a1, b1), (a2, b2) of
// {(x, y): x + y lambda = 0 mod r}; the roundings c1 = floor(k g1 / 2^256), c2 = floor(k g2 / 2^256) with g1 = floor(2^256 b2 / r),
// g2 = floor(2^256 (-b1) / r) are off by at most one, which keeps the bound), so k P = k1 P + k2 phi(P) costs 129 doublings instead
// of 255; the two halves run as interleaved non-adjacent forms (digit i = bit i of 3k minus bit i of k): ~86 additions.
__device__ __constant__ const uint32_t GLV_G1[5] = {0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x00000002u};
__device__ __constant__ const uint32_t GLV_G2[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u};
__device__ __constant__ const uint32_t GLV_A1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
__device__ __constant__ const uint32_t GLV_NB1[2] = {0x94d213e3u, 0x89d32568u};   // -b1 = a2
__device__ __constant__ const uint32_t GLV_B2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
static const uint32_t GLV_BETA_G1[8] = {0x607cfd48u, 0xe4bd44e5u, 0xbb966e3du, 0xc28f069fu, 0xe0acccb0u, 0x5e6dd9e7u, 0xe131a029u, 0x30644e72u};
static const uint32_t GLV_BETA_G2[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0x00000000u, 0x00000000u};
template <class F> Fq glv_beta() {   // Montgomery form
    Fq b;
    for (int i = 0; i < 8; ++i) b.l[i] = sizeof(F) > sizeof(Fq) ? GLV_BETA_G2[i] : GLV_BETA_G1[i];
    return Fq::from_canonical(b);
}

// out[0 .. NO) = low NO words of a[0 .. NA) * b[0 .. NB), starting at word `skip` of the product
template <int NA, int NB, int NO>
__device__ __forceinline__ void mp_mul(const uint32_t* a, const uint32_t* b, int skip, uint32_t* out) {
    uint32_t prod[NA + NB];
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) prod[i] = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            c += (uint64_t)a[i] * b[j] + prod[i + j];
            prod[i + j] = (uint32_t)c;
            c >>= 32;
        }
        prod[i + NB] = (uint32_t)c;
    }
#pragma unroll
    for (int i = 0; i < NO; ++i) out[i] = skip + i < NA + NB ? prod[skip + i] : 0;
}
// |k1|, |k2| (5 words each, < 2^128) and their signs
__device__ __forceinline__ void glv_split(const Fr& k, uint32_t* k1, bool& neg1, uint32_t* k2, bool& neg2) {
    uint32_t kw[8], c1[5], c2[3], g1[5], g2[3], a1[4], nb1[2], b2[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) kw[i] = k.l[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) g1[i] = GLV_G1[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) g2[i] = GLV_G2[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a1[i] = GLV_A1[i]; b2[i] = GLV_B2[i]; }
    nb1[0] = GLV_NB1[0]; nb1[1] = GLV_NB1[1];
    mp_mul<8, 5, 5>(kw, g1, 8, c1);     // c1 = floor(k g1 / 2^256) < 2^128
    mp_mul<8, 3, 3>(kw, g2, 8, c2);     // c2 < 2^65
    // all in 6 words modulo 2^192 (the results fit 129 signed bits):  k1 = k - c1 a1 - c2 a2,  k2 = c1 (-b1) - c2 b2,  a2 = -b1
    uint32_t t1[6], t2[6], t3[6], t4[6];
    mp_mul<5, 4, 6>(c1, a1, 0, t1);
    mp_mul<3, 2, 6>(c2, nb1, 0, t2);
    mp_mul<5, 2, 6>(c1, nb1, 0, t3);
    mp_mul<3, 4, 6>(c2, b2, 0, t4);
    uint32_t r1[6], r2[6];
    {
        int64_t br = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int64_t v = (int64_t)kw[i] - t1[i] - t2[i] + br; r1[i] = (uint32_t)v; br = v >> 32; }
        br = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int64_t v = (int64_t)t3[i] - t4[i] + br; r2[i] = (uint32_t)v; br = v >> 32; }
    }
    neg1 = (r1[5] >> 31) != 0; neg2 = (r2[5] >> 31) != 0;
    {
        uint64_t c = neg1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { c += neg1 ? (uint32_t)~r1[i] : r1[i]; if (i < 5) k1[i] = (uint32_t)c; c >>= 32; }
        c = neg2 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { c += neg2 ? (uint32_t)~r2[i] : r2[i]; if (i < 5) k2[i] = (uint32_t)c; c >>= 32; }
    }
}
__device__ __forceinline__ uint32_t mp_bit(const uint32_t* w, int i) { return (w[i >> 5] >> (i & 31)) & 1; }

template <class F> struct GlvPhi;
template <> struct GlvPhi<Fq> {
    typedef FpR<FqParams> L;
    static __device__ __forceinline__ L mul(const L& x, const L& beta) { return x * beta; }
    static __device__ __forceinline__ L inv(const L& z) { return L::load(z.store_exact().inv_vartime()); }
};
template <> struct GlvPhi<Fq2> {
    typedef FpR<FqParams> L;
    static __device__ __forceinline__ Fp2R<FqParams> mul(const Fp2R<FqParams>& x, const L& beta) { return Fp2R<FqParams>{x.c0 * beta, x.c1 * beta}; }
    static __device__ __forceinline__ Fp2R<FqParams> inv(const Fp2R<FqParams>& z) {   // conj(z) / (z0^2 + z1^2)
        const L ni = L::load(L::mont_sum(z.c0, z.c0, z.c1, z.c1).store_exact().inv_vartime());
        return Fp2R<FqParams>{z.c0 * ni, (z.c1 * ni).neg().norm()};
    }
};

// k P for a canonical scalar k < r (P in the order-r subgroup)
template <class F>
__device__ __noinline__ JacR<F> gb_mul(const JacR<F>& p, const Fr& k, const Fq& beta) {
    JacR<F> acc = p;
    acc.inf = true;
    if (p.inf) return acc;
    uint32_t k1[6], k2[6], h1[6], h2[6];
    bool neg1, neg2;
    glv_split(k, k1, neg1, k2, neg2);
    k1[5] = k2[5] = 0;
    {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { c += 3ull * k1[i]; h1[i] = (uint32_t)c; c >>= 32; }
        c = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { c += 3ull * k2[i]; h2[i] = (uint32_t)c; c >>= 32; }
    }
    // P in affine coordinates (one variable-time inversion: the points are public), so that the ~86 additions are mixed ones
    typedef typename LazyOf<F>::type L;
    const L zi = GlvPhi<F>::inv(p.Z), zi2 = zi.sqr();
    const L x1 = p.X * zi2, x2 = GlvPhi<F>::mul(x1, FpR<FqParams>::load(beta));
    const L yp = (p.Y * zi2) * zi, yn = yp.neg().norm();
    for (int i = 130; i >= 1; --i) {
        if (!acc.inf) acc = dbl_lazy(acc);
        const uint32_t d1 = mp_bit(h1, i) - mp_bit(k1, i), d2 = mp_bit(h2, i) - mp_bit(k2, i);   // 0, 1 or 0xffffffff
        if (d1 && !madd_lazy(acc, x1, ((d1 == 1) != neg1) ? yp : yn)) acc = dbl_lazy(acc);
        if (d2 && !madd_lazy(acc, x2, ((d2 == 1) != neg2) ? yp : yn)) acc = dbl_lazy(acc);
    }
    return acc;
}
template <class F>
__device__ __forceinline__ JacR<F> gb_neg(const JacR<F>& p) {
    JacR<F> m = p;
    m.Y = p.Y.neg().norm();
    return m;
}

// One radix-2 stage over `total` points holding total >> log_size transforms of 2^log_size points each.  tw: canonical w_T^j, j < T / 2,
// for the largest transform size T = 2^log_table of the tree (forward or inverse table).
// DIF (natural -> bit-reversed), stage t = 0 .. L-1: half = N >> (t + 1); (x, y) -> (x + y, (x - y) w_N^(j 2^t))
// DIT (bit-reversed -> natural), stage t = 0 .. L-1: half = 1 << t;       (x, y) -> (x + w y, x - w y), w = w_N^(j N / (2 half))
template <class F, bool DIT>
__global__ __launch_bounds__(64) void k_gb_stage(Jac<F>* __restrict__ data, size_t total, unsigned log_size, unsigned stage, const Fr* __restrict__ tw, unsigned log_table, Fq beta) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= total / 2) return;
    const unsigned log_half = DIT ? stage : log_size - 1 - stage;
    const size_t half = (size_t)1 << log_half;
    const size_t j = b & (half - 1), i0 = ((b >> log_half) << (log_half + 1)) | j, i1 = i0 + half;
    // exponent of w_N, then of the table's w_T
    const size_t e = DIT ? j << (log_size - 1 - log_half) : j << stage;
    const size_t idx = e << (log_table - log_size);
    const JacR<F> x = jacr_load(data[i0]);
    JacR<F> y = jacr_load(data[i1]);
    if (DIT) {
        if (idx) y = gb_mul(y, tw[idx], beta);
        data[i0] = jacr_store(add_lazy(x, y));
        data[i1] = jacr_store(add_lazy(x, gb_neg(y)));
    } else {
        data[i0] = jacr_store(add_lazy(x, y));
        JacR<F> d = add_lazy(x, gb_neg(y));
        if (idx) d = gb_mul(d, tw[idx], beta);
        data[i1] = jacr_store(d);
    }
}

// children' = N_sibling . u / 2s:  h[2p][j] = (nev[2p + 1][j] scale) u[p][j],  h[2p + 1][j] = (nev[2p][j] scale) u[p][j],  j < 2s
template <class F>
__global__ __launch_bounds__(64) void k_gb_spread(const Jac<F>* __restrict__ u, const Fr* __restrict__ nev, Fr scale, size_t s2, size_t parents, Jac<F>* __restrict__ h, Fq beta) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= parents * 2 * s2) return;
    const size_t c = g / s2, j = g - c * s2, p = c >> 1;
    const Fr k = (nev[(c ^ 1) * s2 + j] * scale).to_canonical();
    h[g] = jacr_store(gb_mul(jacr_load(u[p * s2 + j]), k, beta));
}
// next[c s + j] = h[c 2s + j], j < s
template <class F>
__global__ void k_gb_truncate(const Jac<F>* __restrict__ h, size_t s, size_t total, Jac<F>* __restrict__ next) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t c = g / s, j = g - c * s;
    next[g] = h[c * 2 * s + j];
}
template <class F>
__global__ void k_gb_affine(const Jac<F>* __restrict__ in, size_t count, Aff<F>* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = jac_to_affine(in[i]);
}

// y[block GB + k] = sum_i q[block][k][i] g[block GB + i]: one wave per block, lane k; eight scalars at a time share their doublings
template <class F>
__global__ __launch_bounds__(GB) void k_gb_bottom(const Aff<F>* __restrict__ g, const Fr* __restrict__ qmat, size_t n, Jac<F>* __restrict__ y) {
    const size_t blk = blockIdx.x, base = blk * GB;
    const int k = threadIdx.x;
    if (base + k >= n) return;
    const int real = (int)(n - base < (size_t)GB ? n - base : GB);
    const Fr* q = qmat + base * GB + (size_t)k * GB;
    JacR<F> acc = jacr_load(Jac<F>::infinity());
    for (int i0 = 0; i0 < real; i0 += 8) {
        Fr sc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sc[e] = i0 + e < real ? q[i0 + e].to_canonical() : Fr::zero();
        typedef typename LazyOf<F>::type L;
        JacR<F> part = jacr_load(Jac<F>::infinity());
        for (int bit = 255; bit >= 0; --bit) {
            if (!part.inf) part = dbl_lazy(part);
            for (int e = 0; e < 8; ++e) {
                if ((sc[e].l[bit >> 5] >> (bit & 31)) & 1) {
                    const Aff<F> pt = g[base + i0 + e];
                    if (pt.is_inf()) continue;
                    if (!madd_lazy(part, L::load(pt.x), L::load(pt.y))) part = dbl_lazy(part);
                }
            }
        }
        acc = add_lazy(acc, part);
    }
    y[base + k] = jacr_store(acc);
}

// out[k] = sum_i M[i][k] bases[i], k < t.n, for the interpolation map M of the tree's node set (bases: t.n affine points, Montgomery)
template <class F>
void group_interp_transpose(zk_ctx* ctx, const InterpTree& t, const Aff<F>* d_bases, Aff<F>* d_out) {
    const size_t n = t.n, npad = (size_t)1 << t.log_npad;
    hipStream_t st = ctx->stream;
    unsigned lg = 0;
    while ((1u << lg) < (unsigned)GB) ++lg;
    const unsigned levels = t.log_npad - lg;
    DevBuf<Jac<F>> cur(npad), h(levels ? 2 * npad : 1);
    hipLaunchKernelGGL(k_gb_load<F>, dim3(ceil_div(npad, 256)), dim3(256), 0, st, d_bases, n, npad, cur.p);
    ZK_HIP(hipGetLastError());
    DevBuf<Fr> tw_f, tw_i;
    const Fq beta = glv_beta<F>();
    if (levels) {   // canonical w^j and w^-j, j < npad / 2, w of order npad (the largest transform: the root's 2s = npad points)
        tw_f.alloc(npad / 2); tw_i.alloc(npad / 2);
        const Fr w = host_root_of_unity(t.log_npad);
        fr_powers(ctx, w, Fr::one(), tw_f.p, npad / 2);
        fr_powers(ctx, w.inv(), Fr::one(), tw_i.p, npad / 2);
        fr_from_mont(ctx, tw_f.p, tw_f.p, npad / 2);
        fr_from_mont(ctx, tw_i.p, tw_i.p, npad / 2);
    }
    for (unsigned l = levels; l-- > 0;) {
        const size_t s = (size_t)GB << l, s2 = 2 * s, parents = npad / s2;
        const unsigned L = lg + l + 1;   // log2(2s)
        for (unsigned stg = 0; stg < L; ++stg)
            hipLaunchKernelGGL((k_gb_stage<F, false>), dim3(ceil_div(npad / 2, 64)), dim3(64), 0, st, cur.p, npad, L, stg, tw_i.p, t.log_npad, beta);
        const Fr scale = host_fr_pow(host_fr_from_u64(2), L).inv();
        hipLaunchKernelGGL(k_gb_spread<F>, dim3(ceil_div(2 * npad, 64)), dim3(64), 0, st, cur.p, t.nev[l].p, scale, s2, parents, h.p, beta);
        for (unsigned stg = 0; stg < L; ++stg)
            hipLaunchKernelGGL((k_gb_stage<F, true>), dim3(ceil_div(npad, 64)), dim3(64), 0, st, h.p, 2 * npad, L, stg, tw_f.p, t.log_npad, beta);
        hipLaunchKernelGGL(k_gb_truncate<F>, dim3(ceil_div(npad, 256)), dim3(256), 0, st, h.p, s, npad, cur.p);
        ZK_HIP(hipGetLastError());
    }
    DevBuf<Aff<F>> ga(npad);
    hipLaunchKernelGGL(k_gb_affine<F>, dim3(ceil_div(npad, 64)), dim3(64), 0, st, cur.p, npad, ga.p);
    DevBuf<Jac<F>> y(npad);
    hipLaunchKernelGGL(k_gb_bottom<F>, dim3(npad / GB), dim3(GB), 0, st, ga.p, t.qmat.p, n, y.p);
    hipLaunchKernelGGL(k_gb_affine<F>, dim3(ceil_div(n, 64)), dim3(64), 0, st, y.p, n, d_out);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(st));   // the temporaries go out of scope
}
template void group_interp_transpose<Fq>(zk_ctx*, const InterpTree&, const Aff<Fq>*, Aff<Fq>*);
template void group_interp_transpose<Fq2>(zk_ctx*, const InterpTree&, const Aff<Fq2>*, Aff<Fq2>*);

// nodes first .. first + count - 1 as Montgomery field elements
__global__ void k_gb_integer_nodes(uint64_t first, size_t count, Fr* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    Fr v = Fr::zero();
    v.l[0] = (uint32_t)(first + k); v.l[1] = (uint32_t)((first + k) >> 32);
    out[k] = Fr::from_canonical(v);
}

// The three Lagrange-basis arrays of an integer-roots QAP from the powers of the CRS (basis.hip's header has the definitions)
void crs_lagrange_from_powers_tree(zk_ctx* ctx, zk_crs& c, const zk_qap& q) {
    const size_t n = c.n;
    ZK_REQUIRE(q.roots && q.ap && q.n == n, ZK_ERR_ARG, "crs_lagrange_from_powers: not an integer-roots QAP of this CRS");
    ZK_REQUIRE(n <= ((size_t)1 << (NTT_MAX_LOG - 2)), ZK_ERR_UNSUPPORTED, "prove: an integer-roots QAP of more than 2^22 gates needs the CRS zk_setup made for it");
    c.lag1.alloc(n); c.lag2.alloc(n); c.lagS_t1.alloc(std::max<size_t>(n - 1, 1));
    hipStream_t st = ctx->stream;
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    {
        DevBuf<Fr> nodes(n);
        hipLaunchKernelGGL(k_gb_integer_nodes, dim3(ceil_div(n, 256)), dim3(256), 0, st, (uint64_t)1, n, nodes.p);
        ZK_HIP(hipGetLastError());
        auto tree = interp_build(ctx, nodes.p, n, flag.p);
        group_interp_transpose<Fq>(ctx, *tree, c.xi1.p, c.lag1.p);
        group_interp_transpose<Fq2>(ctx, *tree, c.xi2.p, c.lag2.p);
    }
    if (n >= 2) {
        DevBuf<Fr> nodes(n - 1);
        hipLaunchKernelGGL(k_gb_integer_nodes, dim3(ceil_div(n - 1, 256)), dim3(256), 0, st, (uint64_t)n + 1, n - 1, nodes.p);
        ZK_HIP(hipGetLastError());
        auto tree = interp_build(ctx, nodes.p, n - 1, flag.p);
        group_interp_transpose<Fq>(ctx, *tree, c.xi_t1.p, c.lagS_t1.p);
    }
    c.ap = true;
}

}  // namespace zk
