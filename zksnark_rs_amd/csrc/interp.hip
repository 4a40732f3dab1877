// interp.hip -- interpolation through ARBITRARY distinct nodes in O(n log^2 n): the coefficients of the polynomial of degree < n
// that takes given values on caller-supplied roots r_0 .. r_{n-1}.
//
// QAP::from (/root/reference/src/groth16/fr.rs:140-173) interpolates every wire polynomial through the roots of the
// RootRepresentation (circuit/mod.rs:201-214: `roots()` is caller data) with Lagrange sums (coefficient_poly.rs:159-200, O(n^2) per
// polynomial).  The prover only ever needs the two combinations U = sum a_i u_i and V (arbroots.hip: the quotient of U V alone by t is
// h), so it interpolates THOSE, per proof, from their values on the roots (the SpMV output):
//     F(x) = sum_k a_k N(x) / (x - r_k),   a_k = F_k / N'(r_k),   N(x) = prod_k (x - r_k)
// by the sub-product tree: a node that covers the leaves [lo, hi) holds N_node = prod (x - r_k) and P_node = sum a_k N_node / (x - r_k);
//     P_parent = P_left N_right + P_right N_left,   N_parent = N_left N_right.
// Layout for the GPU:
//   * the bottom of the tree is flat: blocks of 64 leaves with a per-QAP 64 x 64 matrix q[i][k] = coefficient i of N_block / (x - r_k)
//     divided by N'(r_k), so a block's P is one small matrix-vector product per proof (no tree below 64 leaves, no tiny transforms);
//   * above, one level = batched transforms (ntt.hip, one launch per pass for ALL nodes of the level and all vectors): children padded
//     to twice their size, forward DIF, P_l N_r + P_r N_l point-wise against the stored images of the children's N, inverse DIT;
//   * leaves beyond n are empty (N = 1, P = 0), so n is arbitrary; a node's N has degree = its number of real leaves.
// Per-QAP precompute, O(n log^2 n) as well: the block matrices, the images of every node's N (2 npad elements per level), t = N_root,
// and the weights 1 / N'(r_k) by the SCALED REMAINDER TREE over the same images (Bernstein): N'/N = sum_i c_i x^-i with
// c = rev(N') / rev(N) as power series (one Newton inversion at the root); a node v keeps the first d_v terms of the fractional part of
// N'/N_v as G_v = sum c_i x^(d_v - i), and a child's is a MIDDLE product of its parent's with the sibling's N -- coefficients
// [d_sibling, d_parent) of the cyclic product of twice the child's size, which the transform does not wrap; at a block,
// N' mod N_block = coefficients [d, 2 d) of G N_block, evaluated at the block's roots by Horner.  (The first version multiplied
// prod_{j != k} (r_k - r_j) out directly: 10.9 s at 2^20 gates.)
#include <vector>
#include "pipeline.hpp"
#include "interp.hpp"
#include "lazy29.cuh"
#include "fr_tile.cuh"

namespace zk {

constexpr int IB = INTERP_BLOCK;   // leaves per bottom block

// ---- per-QAP tables ---------------------------------------------------------------------------------------------------------
// number of real leaves under node `idx` of a level whose nodes cover `size` leaves each
__device__ __host__ inline size_t interp_degree(size_t idx, size_t size, size_t n) {
    const size_t lo = idx * size;
    return lo >= n ? 0 : (n - lo < size ? n - lo : size);
}

// ---- weights: 1 / N'(r_k) by the scaled remainder tree ----
// rev(N')[i] = (n - i) t_(n - i), i < n, zero padded to `size`
__global__ void k_interp_rev_derivative(const Fr* __restrict__ t, size_t n, size_t size, Fr* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    if (i >= n) { out[i] = Fr::zero(); return; }
    Fr k = Fr::zero();
    const uint64_t f = (uint64_t)(n - i);
    k.l[0] = (uint32_t)f; k.l[1] = (uint32_t)(f >> 32);
    out[i] = Fr::from_canonical(k) * t[n - i];
}
// G_root[j] = c_(n - j) = series[n - 1 - j], j < n, zero padded to npad
__global__ void k_interp_root_series(const Fr* __restrict__ series, size_t n, size_t npad, Fr* __restrict__ g) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= npad) return;
    g[j] = j < n ? series[n - 1 - j] : Fr::zero();
}
// out[c][j] = gimg[c / 2][j] nev[sibling of c][j], j < 2s: the images of the two middle products of every parent
__global__ void k_interp_down_mul(const Fr* __restrict__ gimg, const Fr* __restrict__ nev, size_t s2, size_t children, Fr* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= children * s2) return;
    const size_t c = g / s2, j = g - c * s2;
    out[g] = gimg[(c >> 1) * s2 + j] * nev[(c ^ 1) * s2 + j];
}
// G_child[j] = cyc[child][d_sibling + j], j < d_child (children of s leaves; cyc: children x 2s), zero padded to s
__global__ void k_interp_down_take(const Fr* __restrict__ cyc, size_t s, size_t children, size_t n, Fr* __restrict__ g) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= children * s) return;
    const size_t c = t / s, j = t - c * s;
    const size_t dc = interp_degree(c, s, n), ds = interp_degree(c ^ 1, s, n);
    g[t] = j < dc ? cyc[c * 2 * s + ds + j] : Fr::zero();
}
// One workgroup per block: R = N' mod N_block = coefficients [d, 2d) of G N_block, w_k = 1 / R(r_k); flag |= 16 when a root repeats.
// blk_poly: the blocks' N as full coefficient lists, 2 IB apart (what k_interp_blocks wrote)
__global__ __launch_bounds__(INTERP_BLOCK) void k_interp_weights(const Fr* __restrict__ g, const Fr* __restrict__ blk_poly, const Fr* __restrict__ r, size_t n,
                                                              Fr* __restrict__ w, int* __restrict__ flag) {
    __shared__ Fr gs[IB], ns[IB + 1], rs_[IB];
    const size_t blk = blockIdx.x, base = blk * IB;
    const int k = threadIdx.x;
    const int d = (int)(base >= n ? 0 : (n - base < (size_t)IB ? n - base : IB));
    if (d == 0) return;
    gs[k] = g[base + k];
    ns[k] = blk_poly[blk * (size_t)(2 * IB) + k];
    if (k == 0) ns[IB] = blk_poly[blk * (size_t)(2 * IB) + IB];
    __syncthreads();
    Fr acc = Fr::zero();
    if (k < d)
        for (int i = k; i < d; ++i) acc = acc + gs[i] * ns[d + k - i];     // (G N)[d + k]
    rs_[k] = acc;
    __syncthreads();
    if (k >= d) return;
    const Fr x = r[base + k];
    Fr v = Fr::zero();
    for (int i = d - 1; i >= 0; --i) v = v * x + rs_[i];
    if (v.is_zero()) { atomicOr(flag, 16); w[base + k] = Fr::zero(); return; }
    w[base + k] = v.inv();
}
// q[block][i][k] *= w_k
__global__ void k_interp_scale_q(Fr* __restrict__ qmat, const Fr* __restrict__ w, size_t n, size_t npad) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= npad * IB) return;
    const size_t blk = g / ((size_t)IB * IB), k = (g / IB) % IB, leaf = blk * IB + k;   // layout q[block][k][i]
    if (leaf < n) qmat[g] = qmat[g] * w[leaf];
}

// One workgroup (IB lanes) per block of IB leaves.  Lane k: Q_k = prod_{j in block, j != k} (x - r_j) (IB coefficients, degree = real
// leaves - 1), q[block][i][k] = Q_k[i] (k_interp_scale_q multiplies by w_k once the weights exist).  Lane 0 also writes the block's N = Q_0 (x - r_0) as a full coefficient list (degree
// d = real leaves of the block) into node[block * 2 IB ..] (zero padded to 2 IB: the layout of level 0).
__global__ __launch_bounds__(INTERP_BLOCK) void k_interp_blocks(const Fr* __restrict__ r, size_t n, Fr* __restrict__ qmat, Fr* __restrict__ node) {
    __shared__ Fr rs[IB];
    const size_t blk = blockIdx.x, base = blk * IB;
    const int k = threadIdx.x;
    const int real = (int)(base >= n ? 0 : (n - base < (size_t)IB ? n - base : IB));
    rs[k] = k < real ? r[base + k] : Fr::zero();
    __syncthreads();
    Fr c[IB + 1];
#pragma unroll 1
    for (int i = 0; i <= IB; ++i) c[i] = Fr::zero();
    c[0] = Fr::one();
    int deg = 0;
    for (int j = 0; j < real; ++j) {
        if (j == k) continue;
        const Fr rj = rs[j];
        // c <- c (x - r_j)
        for (int i = deg + 1; i >= 1; --i) c[i] = c[i - 1] - rj * c[i];
        c[0] = Fr::zero() - rj * c[0];
        ++deg;
    }
    Fr* q = qmat + blk * (size_t)IB * IB;
    // stored as q[block][k][i]: the lanes of k_interp_bottom (one per output coefficient i) read consecutive elements
    for (int i = 0; i < IB; ++i) q[(size_t)k * IB + i] = (k < real && i <= deg) ? c[i] : Fr::zero();
    if (k == 0) {
        Fr* out = node + blk * (size_t)(2 * IB);
        if (real == 0) {
            out[0] = Fr::one();
            for (int i = 1; i < 2 * IB; ++i) out[i] = Fr::zero();
        } else {
            // N_block = Q_0 (x - r_0)
            const Fr r0 = rs[0];
            Fr prev = Fr::zero();
            for (int i = 0; i <= deg + 1; ++i) {
                const Fr ci = i <= deg ? c[i] : Fr::zero();
                out[i] = prev - r0 * ci;
                prev = ci;
            }
            for (int i = deg + 2; i < 2 * IB; ++i) out[i] = Fr::zero();
        }
    }
}

// parents of a level from the cyclic products of their children (prod: parents x 2s, the product of two polynomials of degree <= s
// modulo x^(2s) - 1): the full coefficient list of degree d = d_l + d_r, zero padded to 4s -- the children layout of the next level.
// When d = 2s the leading 1 wrapped onto coefficient 0.
__global__ void k_interp_place(const Fr* __restrict__ prod, size_t parents, size_t s2, size_t n, Fr* __restrict__ next) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= parents * 2 * s2) return;
    const size_t p = g / (2 * s2), j = g - p * 2 * s2;
    const size_t d = interp_degree(p, s2, n);
    Fr v = Fr::zero();
    if (j < s2) {
        v = prod[p * s2 + j];
        if (j == 0 && d == s2) v = v - Fr::one();
    } else if (j == s2 && d == s2) {
        v = Fr::one();
    }
    next[g] = v;
}
__global__ void k_interp_root(const Fr* __restrict__ prod, size_t npad, size_t n, Fr* __restrict__ t) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n) return;
    Fr v = j < npad ? prod[j] : Fr::zero();
    if (n == npad) { if (j == 0) v = v - Fr::one(); if (j == n) v = Fr::one(); }
    t[j] = v;
}

// out[p][j] = node[2p][j] node[2p + 1][j], j < 2s
__global__ void k_interp_mul_pairs(const Fr* __restrict__ node, size_t s2, size_t parents, Fr* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= parents * s2) return;
    const size_t p = g / s2, j = g - p * s2;
    out[g] = node[(2 * p) * s2 + j] * node[(2 * p + 1) * s2 + j];
}

// ---- per proof --------------------------------------------------------------------------------------------------------------
// P of the bottom blocks: out[v][blk IB + i] = sum_k values[v][blk IB + k] q[blk][k][i]   (values beyond n are not read).
// One lane per output coefficient serves all CNT vectors of the launch, so the matrix (64 x 32 B per coefficient: 2.1 GB at 2^20
// leaves) is read once, coalesced; the sums run in the lazy radix-2^29 form of lazy29.cuh -- two products per Montgomery reduction
// (mont_sum), limb-wise accumulation, one closing reduction (fr_store_exact takes any |value| < 2^9 p: here <= 32 terms below 3p).
// (Round 3: one launch row per vector, rows of q strided by 2 KB across the lanes, 8 x 32 multiplications: 2.3 ms per 2^20-gate proof.)
template <int CNT>
__global__ __launch_bounds__(256) void k_interp_bottom(const Fr* __restrict__ values, size_t vstride, int v0, const Fr* __restrict__ qmat, size_t n, size_t npad,
                                                       Fr* __restrict__ out) {
    typedef FpR<FrParams> L;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= npad) return;
    const size_t blk = g / IB, base = blk * IB, i = g - base;
    const Fr* q = qmat + base * IB + i;
    const int real = (int)(base >= n ? 0 : (n - base < (size_t)IB ? n - base : IB));
    L acc[CNT];
#pragma unroll
    for (int v = 0; v < CNT; ++v) acc[v] = L::load(Fr::zero());
    for (int k = 0; k < real; k += 2) {
        const bool two = k + 1 < real;
        const L q0 = L::load(q[(size_t)k * IB]), q1 = two ? L::load(q[(size_t)(k + 1) * IB]) : L::load(Fr::zero());
#pragma unroll
        for (int v = 0; v < CNT; ++v) {
            const Fr* f = values + (size_t)(v0 + v) * vstride + base;
            const L f0 = L::load(f[k]), f1 = two ? L::load(f[k + 1]) : L::load(Fr::zero());
            acc[v] = (acc[v] + L::mont_sum(f0, q0, f1, q1)).norm();       // <= 32 terms of |value| < 3p: limbs stay in range
        }
    }
#pragma unroll
    for (int v = 0; v < CNT; ++v) out[(size_t)(v0 + v) * npad + g] = fr_store_exact(acc[v]);
}
// children (size s, contiguous) -> zero padded to 2s
__global__ void k_interp_pad(const Fr* __restrict__ cur, size_t s, size_t total2, Fr* __restrict__ tmp) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total2) return;
    const size_t c = g / (2 * s), j = g - c * 2 * s;
    tmp[g] = j < s ? cur[c * s + j] : Fr::zero();
}
// out[v][p][j] = P_l[j] N_r[j] + P_r[j] N_l[j], j < 2s; nev: the images of the level's children N (same order), shared by the vectors
// keep: the result is also left in the left child's slot of tmp -- the first half of the parent's image at twice the size, i.e. what
// k_interp_half(.., 0, ..) would copy there (only this lane reads or writes those two slots, so the update in place is safe)
__global__ void k_interp_combine(Fr* tmp, const Fr* __restrict__ nev, size_t s2, size_t parents, Fr* __restrict__ out, bool keep) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= parents * s2) return;
    const size_t p = g / s2, j = g - p * s2;
    Fr* t = tmp + (size_t)blockIdx.y * parents * 2 * s2;
    const size_t l = (2 * p) * s2 + j, r = (2 * p + 1) * s2 + j;
    const Fr v = fr_store_exact(FrL::mont_sum(FrL::load(t[l]), FrL::load(nev[r]), FrL::load(t[r]), FrL::load(nev[l])));   // one reduction for both products
    out[(size_t)blockIdx.y * parents * s2 + g] = v;
    if (keep) t[l] = v;
}

// The image (DIF order, 4s points) of a parent's 2s coefficients zero padded to 4s is [its 2s-point image | the 2s-point image of the
// coefficients times w_4s^j]: the first stage of the larger transform only copies and twists.  The left half is what the combination just
// produced; so a level costs one inverse and one forward transform of 2s points per node instead of one inverse of 2s and one forward of 4s.
// next[v][p][half 2s + j] = src[v][p][j]
__global__ void k_interp_half(const Fr* __restrict__ src, size_t s2, size_t total, int half, Fr* __restrict__ next) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t p = g / s2, j = g - p * s2;
    next[p * 2 * s2 + (size_t)half * s2 + j] = src[g];
}
// coefficients times tw[(j mod 2s) step]  (interp_run: the level's table w_4s^j / 2s, step 1)
__global__ void k_interp_twist(Fr* __restrict__ c, const Fr* __restrict__ tw, size_t s2, size_t step, size_t total) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    c[g] = c[g] * tw[(g & (s2 - 1)) * step];
}

std::shared_ptr<InterpTree> interp_build(zk_ctx* ctx, const Fr* d_roots_mont, size_t n, int* d_flag) {
    ZK_REQUIRE(n >= 1 && n <= ((size_t)1 << (NTT_MAX_LOG - 1)), ZK_ERR_SIZE, "interpolation: n must be in [1, 2^23]");
    auto t = std::make_shared<InterpTree>();
    hipStream_t st = ctx->stream;
    t->n = n;
    unsigned L = 0;
    while (((size_t)1 << L) < std::max<size_t>(n, IB)) ++L;
    t->log_npad = L;
    const size_t npad = (size_t)1 << L;
    t->roots.alloc(n);
    ZK_HIP(hipMemcpyAsync(t->roots.p, d_roots_mont, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    t->w.alloc(n);
    const size_t blocks = npad / IB;
    t->qmat.alloc(npad * IB);
    DevBuf<Fr> node(2 * npad), prod(npad), blk_poly(2 * npad);
    hipLaunchKernelGGL(k_interp_blocks, dim3(blocks), dim3(IB), 0, st, t->roots.p, n, t->qmat.p, node.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(blk_poly.p, node.p, 2 * npad * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    // levels: children of s = IB << l leaves, images at size 2s
    unsigned lg = 0;
    while ((1u << lg) < (unsigned)IB) ++lg;
    for (unsigned l = 0; lg + l < L; ++l) {
        const size_t s = (size_t)IB << l, s2 = 2 * s, children = npad / s, parents = children / 2;
        if (lg + l + 1 < L) {   // interp_run: the twist w_4s^j of a parent's coefficients, with the 1 / 2s of the unscaled inverse transform before it
            t->tws.emplace_back();
            t->tws.back().alloc(s2);
            fr_powers(ctx, host_root_of_unity(lg + l + 2), host_fr_pow(host_fr_from_u64(2), lg + l + 1).inv(), t->tws.back().p, s2);
        }
        ntt_dif(ctx, node.p, lg + l + 1, false, false, children);
        t->nev.emplace_back();
        t->nev.back().alloc(2 * npad);
        ZK_HIP(hipMemcpyAsync(t->nev.back().p, node.p, 2 * npad * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        // N_parent = N_l N_r (cyclic, size 2s)
        hipLaunchKernelGGL(k_interp_mul_pairs, dim3(ceil_div(parents * s2, 256)), dim3(256), 0, st, node.p, s2, parents, prod.p);
        ntt_dit(ctx, prod.p, lg + l + 1, true, true, nullptr, parents);
        if (parents > 1) hipLaunchKernelGGL(k_interp_place, dim3(ceil_div(parents * 2 * s2, 256)), dim3(256), 0, st, prod.p, parents, s2, n, node.p);
        ZK_HIP(hipGetLastError());
    }
    t->t.alloc(n + 1);
    if (L == lg) {   // a single block: its N sits in node[0 .. n]
        ZK_HIP(hipMemcpyAsync(t->t.p, node.p, (n + 1) * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL(k_interp_root, dim3(ceil_div(n + 1, 256)), dim3(256), 0, st, prod.p, npad, n, t->t.p);
        ZK_HIP(hipGetLastError());
    }
    // ---- the weights 1 / N'(r_k): scaled remainder tree, top down over the images stored above ----
    {
        unsigned lc = 1;
        while (((size_t)1 << lc) < 2 * n) ++lc;
        const size_t nc = (size_t)1 << lc;
        DevBuf<Fr> one(1), ginv, series(nc);
        const Fr o = Fr::one();
        ZK_HIP(hipMemcpyAsync(one.p, &o, sizeof(Fr), hipMemcpyHostToDevice, st));
        poly_rev_inverse_ntt(ctx, t->t.p, n, one.p, n, lc, ginv);                     // 1 / rev(N) mod x^n
        hipLaunchKernelGGL(k_interp_rev_derivative, dim3(ceil_div(nc, 256)), dim3(256), 0, st, t->t.p, n, nc, series.p);
        ntt_dif(ctx, series.p, lc, false, false);
        fr_pointwise_mul(ctx, series.p, ginv.p, series.p, nc);
        ntt_dit(ctx, series.p, lc, true, true, nullptr);                              // c_1 .. c_n = its first n coefficients
        Fr* g = prod.p;                                                               // npad: G of the current level's nodes
        hipLaunchKernelGGL(k_interp_root_series, dim3(ceil_div(npad, 256)), dim3(256), 0, st, series.p, n, npad, g);
        for (unsigned l = L - lg; l-- > 0;) {
            const size_t s = (size_t)IB << l, s2 = 2 * s, children = npad / s, parents = children / 2;
            ntt_dif(ctx, g, lg + l + 1, false, false, parents);                        // images of the parents' G (size 2s each)
            hipLaunchKernelGGL(k_interp_down_mul, dim3(ceil_div(children * s2, 256)), dim3(256), 0, st, g, t->nev[l].p, s2, children, node.p);
            ntt_dit(ctx, node.p, lg + l + 1, true, true, nullptr, children);
            hipLaunchKernelGGL(k_interp_down_take, dim3(ceil_div(children * s, 256)), dim3(256), 0, st, node.p, s, children, n, g);
            ZK_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(k_interp_weights, dim3(blocks), dim3(IB), 0, st, g, blk_poly.p, t->roots.p, n, t->w.p, d_flag);
        hipLaunchKernelGGL(k_interp_scale_q, dim3(ceil_div(npad * IB, 256)), dim3(256), 0, st, t->qmat.p, t->w.p, n, npad);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(st));   // the temporaries go out of scope
    }
    return t;
}

void interp_run(zk_ctx* ctx, const InterpTree& t, const Fr* d_values, size_t vstride, size_t count, Fr* d_work, Fr* d_out) {
    const size_t n = t.n, npad = (size_t)1 << t.log_npad;
    hipStream_t st = ctx->stream;
    Fr* cur = d_out;                      // count x npad
    Fr* tmp = d_work;                     // count x 2 npad
    Fr* alt = d_work + 2 * count * npad;  // count x npad
    for (size_t v = 0; v < count; v += 2) {
        if (count - v >= 2) hipLaunchKernelGGL(k_interp_bottom<2>, dim3(ceil_div(npad, 256)), dim3(256), 0, st, d_values, vstride, (int)v, t.qmat.p, n, npad, cur);
        else hipLaunchKernelGGL(k_interp_bottom<1>, dim3(ceil_div(npad, 256)), dim3(256), 0, st, d_values, vstride, (int)v, t.qmat.p, n, npad, cur);
    }
    unsigned lg = 0;
    while ((1u << lg) < (unsigned)IB) ++lg;
    const unsigned levels = t.log_npad - lg;
    if (npad * count < ((size_t)1 << std::min<long>(std::max<long>(ctx->opt_interp_large_log, 0), 40))) {
        // small trees: three more launches per level cost more than a third of its transforms saves -- zero pad and transform at 4s
        for (unsigned l = 0; l < levels; ++l) {
            const size_t s = (size_t)IB << l, s2 = 2 * s, children = npad / s, parents = children / 2;
            hipLaunchKernelGGL(k_interp_pad, dim3(ceil_div(2 * npad * count, 256)), dim3(256), 0, st, cur, s, 2 * npad * count, tmp);
            ntt_dif(ctx, tmp, lg + l + 1, false, false, children * count);
            Fr* nxt = cur == d_out ? alt : d_out;
            hipLaunchKernelGGL(k_interp_combine, dim3(ceil_div(parents * s2, 256), count), dim3(256), 0, st, tmp, t.nev[l].p, s2, parents, nxt, false);
            ntt_dit(ctx, nxt, lg + l + 1, true, true, nullptr, parents * count);
            cur = nxt;
        }
        if (cur != d_out) ZK_HIP(hipMemcpyAsync(d_out, cur, count * npad * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipGetLastError());
        return;
    }
    if (levels) {   // images of the bottom blocks at twice their size
        hipLaunchKernelGGL(k_interp_pad, dim3(ceil_div(2 * npad * count, 256)), dim3(256), 0, st, cur, (size_t)IB, 2 * npad * count, tmp);
        ntt_dif(ctx, tmp, lg + 1, false, false, (npad / IB) * count);
    }
    for (unsigned l = 0; l < levels; ++l) {
        const size_t s2 = (size_t)IB << (l + 1), parents = npad / s2, total = npad * count;
        hipLaunchKernelGGL(k_interp_combine, dim3(ceil_div(parents * s2, 256), count), dim3(256), 0, st, tmp, t.nev[l].p, s2, parents, alt, l + 1 < levels);   // parents' images, 2s points (+ the first half of their 4s-point images, in place)
        ntt_dit(ctx, alt, lg + l + 1, true, l + 1 == levels, nullptr, parents * count);   // their coefficients -- times 2s below the root: the twist table divides
        if (l + 1 == levels) break;
        if (lg + l + 1 <= 22) {
            ntt_dif_pre(ctx, alt, lg + l + 1, t.tws[l].p, 1, parents * count);   // the twist w_4s^j / 2s rides on the transform's first load
        } else {
            hipLaunchKernelGGL(k_interp_twist, dim3(ceil_div(total, 256)), dim3(256), 0, st, alt, t.tws[l].p, s2, (size_t)1, total);
            ntt_dif(ctx, alt, lg + l + 1, false, false, parents * count);
        }
        hipLaunchKernelGGL(k_interp_half, dim3(ceil_div(total, 256)), dim3(256), 0, st, alt, s2, total, 1, tmp);
    }
    if (levels) ZK_HIP(hipMemcpyAsync(d_out, alt, count * npad * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ZK_HIP(hipGetLastError());
}

// zk_interpolate_fr: host arrays in, n coefficients out (diagnostic entry point of the parity tests)
void interp_host(zk_ctx* ctx, const uint64_t* roots, const uint64_t* values, size_t n, uint64_t* coeffs) {
    ZK_REQUIRE(roots && values && coeffs && n >= 1, ZK_ERR_ARG, "zk_interpolate_fr: null pointer or n = 0");
    hipStream_t st = ctx->stream;
    DevBuf<Fr> r(n), v(n);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    ZK_HIP(hipMemcpyAsync(r.p, roots, n * sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(v.p, values, n * sizeof(Fr), hipMemcpyHostToDevice, st));
    fr_to_mont(ctx, r.p, r.p, n, flag.p);
    fr_to_mont(ctx, v.p, v.p, n, flag.p);
    auto t = interp_build(ctx, r.p, n, flag.p);
    const size_t npad = (size_t)1 << t->log_npad;
    DevBuf<Fr> work(3 * npad), out(npad);
    interp_run(ctx, *t, v.p, n, 1, work.p, out.p);
    fr_from_mont(ctx, out.p, out.p, n);
    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(coeffs, out.p, n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_REQUIRE(!(hflag & 3), ZK_ERR_RANGE, "zk_interpolate_fr: element >= modulus");
    ZK_REQUIRE(!(hflag & 16), ZK_ERR_ARG, "zk_interpolate_fr: the roots are not distinct");
}

}  // namespace zk
