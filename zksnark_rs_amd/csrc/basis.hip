// basis.hip -- the Lagrange-basis points of an integer-roots QAP from a CRS that carries only the reference's powers.
//
// groth16::prove takes any (&SigmaG1, &SigmaG2) that setup emitted (/root/reference/src/groth16/mod.rs:172-194, 213-217): the arrays
// [x^i]_1, [x^i]_2, [x^i t(x)/delta]_1.  The integer-roots form (aproots.hip) proves in the evaluation basis and multiplies with
//     lag1[k-1]  = [L_k(x)]_1,  lag2[k-1] = [L_k(x)]_2          k in R = {1..n}        (L_k: Lagrange basis of R)
//     lagS_t1[i] = [L^S_s(x) t(x)/delta]_1                      s = n + 1 + i in S = {n+1 .. 2n-1}
// which zk_setup writes directly because it knows x.  For a CRS made elsewhere (zk_crs_upload, a ZKCRSv1 file) they are public
// linear combinations of the powers:  L_k(x) = w_k N(x) / (x - k) = sum_i c_{k,i} x^i,  so  [L_k(x)] = sum_i c_{k,i} [x^i] -- one inner
// product of n terms per node, n of them per array: O(n^2) scalar-point products, done here ONCE per CRS (and kept by zk_crs_save
// in the ZKCRSv2 container) for n <= 2^16 + 2^10 gates; beyond that the CRS must come from zk_setup (ZK_ERR_UNSUPPORTED).
//   1. N(x) = prod_{j in R} (x - j) in coefficient form: its values on 2^k-th roots of unity (n multiplications per point, all
//      points in parallel), then one inverse NTT;  likewise N_S.
//   2. rows of scalars, 64 nodes at a time: c_{k,.} = w_k . (N / (x - k)) by synthetic division, a workgroup per node (the
//      recurrence e_p = N_{D-p} + k e_{p-1} as a chunked scan);
//   3. the 64 inner products of a batch as ONE grouped MSM over the same bases (msm_impl.hpp: group j = node j with its own buckets).
#include "pipeline.hpp"
#include "qap_kernels.hpp"

namespace zk {

constexpr size_t BASIS_MAX_N = ((size_t)1 << 16) + 1024;
size_t basis_max_n() { return BASIS_MAX_N; }
constexpr int BASIS_GROUPS = 64;          // nodes per grouped MSM (the level-1 counters of 64 groups x 2^8 bins fill 64 KiB of LDS)

__device__ __forceinline__ Fr fr_from_u64_dev(uint64_t v) {   // Montgomery form of a small integer
    Fr k = Fr::zero();
    k.l[0] = (uint32_t)v;
    k.l[1] = (uint32_t)(v >> 32);
    return Fr::from_canonical(k);
}

// out[i] = prod_{j < count} (w^i - (first + j)),  i < 2^log_p
__global__ void k_basis_eval_product(Fr w, uint64_t first, size_t count, unsigned log_p, Fr* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_p) return;
    Fr x = Fr::one(), b = w;   // x = w^i
    for (size_t e = i; e; e >>= 1) { if (e & 1) x = x * b; b = b.sqr(); }
    Fr node = fr_from_u64_dev(first);
    const Fr one = Fr::one();
    Fr acc = one;
    for (size_t j = 0; j < count; ++j) { acc = acc * (x - node); node = node + one; }
    out[i] = acc;
}

// coefficients (Montgomery) of prod_{j < count} (x - (first + j)): count + 1 of them, the last one is 1
static void poly_from_integer_roots(zk_ctx* ctx, uint64_t first, size_t count, DevBuf<Fr>& coef) {
    unsigned lg = 1;
    while (((size_t)1 << lg) < count + 1) ++lg;
    const size_t P = (size_t)1 << lg;
    coef.alloc(P);
    hipLaunchKernelGGL(k_basis_eval_product, dim3(ceil_div(P, 64)), dim3(64), 0, ctx->stream, host_root_of_unity(lg), first, count, lg, coef.p);
    ZK_HIP(hipGetLastError());
    DevBuf<Fr> tmp(P);
    bitrev_permute(ctx, coef.p, tmp.p, lg);                       // natural -> bit-reversed, the order ntt_dit takes
    ntt_dit(ctx, tmp.p, lg, true, true, nullptr);                 // inverse transform: values on <w> -> coefficients, natural order
    ZK_HIP(hipMemcpyAsync(coef.p, tmp.p, P * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

// Rows of scalars.  Node g of the batch (value node0 + g, barycentric weight wt[g]) gets row[g][i] = canonical(wt . m_i), i < D, where
// N(x) / (x - node) = sum_i m_i x^i for the monic N of degree D given by `coef` (D + 1 coefficients).  With e_p = m_{D-1-p}:
// e_0 = 1, e_p = N_{D-p} + node e_{p-1}.  A workgroup per node: every lane runs the recurrence over its own chunk of p from zero,
// the lanes' carries are combined (lane 0, 256 steps), a second sweep adds carry x node^(p - chunk start + 1).
__global__ __launch_bounds__(256) void k_basis_rows(const Fr* __restrict__ coef, size_t D, uint64_t node0, const Fr* __restrict__ wt, size_t row_stride,
                                                    Fr* __restrict__ rows) {
    __shared__ Fr carry[256];
    __shared__ Fr step_pow;   // node^L
    const int t = threadIdx.x;
    const uint64_t nodev = node0 + blockIdx.x;
    const Fr node = fr_from_u64_dev(nodev);
    const Fr w = wt[blockIdx.x];
    Fr* row = rows + (size_t)blockIdx.x * row_stride;
    const size_t L = (D + 255) / 256, p0 = (size_t)t * L, p1 = min(p0 + L, D);
    Fr run = Fr::zero();
    for (size_t p = p0; p < p1; ++p) {
        run = run * node + coef[D - p];
        row[D - 1 - p] = run;                // e'_p, carry-in taken as zero
    }
    carry[t] = run;
    if (t == 0) {
        Fr pw = Fr::one();
        for (size_t i = 0; i < L; ++i) pw = pw * node;
        step_pow = pw;
    }
    __syncthreads();
    if (t == 0) {   // carry[t] <- the value e_{p0 - 1} that enters lane t's chunk
        Fr in = Fr::zero();
        const Fr pw = step_pow;
        for (int u = 0; u < 256; ++u) {
            const Fr out = carry[u] + in * pw;       // e at the end of chunk u (a short last chunk has fewer steps, but nothing follows it)
            carry[u] = in;
            in = out;
        }
    }
    __syncthreads();
    const Fr cin = carry[t];
    Fr pw = node;
    for (size_t p = p0; p < p1; ++p) {
        row[D - 1 - p] = ((row[D - 1 - p] + cin * pw) * w).to_canonical();
        pw = pw * node;
    }
}

template <class F>
__global__ void k_basis_store(const Jac<F>* __restrict__ in, size_t count, Aff<F>* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = jac_to_affine(in[i]);
}

// out[g] = sum_i row_g[i] bases[i] for the `nodes` nodes node0 .. : the rows in batches, one grouped MSM per batch
template <class F>
static void basis_products(zk_ctx* ctx, const Aff<F>* d_bases, size_t D, const DevBuf<Fr>& coef, uint64_t node0, size_t nodes, const Fr* d_weights, Aff<F>* d_out) {
    if (!nodes || !D) return;
    MsmTable<F> tab;
    msm_build_table<F>(ctx, d_bases, D, msm_auto_window(D), tab);
    MsmWorkspace ws;
    DevBuf<Fr> rows((size_t)BASIS_GROUPS * D);
    DevBuf<Jac<F>> sums(BASIS_GROUPS);
    hipStream_t st = ctx->stream;
    for (size_t at = 0; at < nodes; at += BASIS_GROUPS) {
        const int g = (int)std::min<size_t>(BASIS_GROUPS, nodes - at);
        hipLaunchKernelGGL(k_basis_rows, dim3(g), dim3(256), 0, st, coef.p, D, node0 + at, d_weights + at, D, rows.p);
        ZK_HIP(hipGetLastError());
        MsmGroups grp;
        grp.groups = g; grp.glen = D; grp.valid = D; grp.out_stride = sizeof(Jac<F>);
        if (g == 1) msm_run<F>(ctx, ws, st, tab, rows.p, D, 0, 1, sums.p);
        else msm_run<F>(ctx, ws, st, tab, rows.p, 0, 0, 1, sums.p, nullptr, nullptr, 0, grp);
        hipLaunchKernelGGL(k_basis_store<F>, dim3(1), dim3(64), 0, st, sums.p, (size_t)g, d_out + at);
        ZK_HIP(hipGetLastError());
    }
    ZK_HIP(hipStreamSynchronize(st));   // the table, the workspace and the rows go out of scope
}

void crs_lagrange_from_powers(zk_ctx* ctx, zk_crs& c, const zk_qap& q) {
    const size_t n = c.n;
    ZK_REQUIRE(q.roots && q.ap && q.n == n, ZK_ERR_ARG, "crs_lagrange_from_powers: not an integer-roots QAP of this CRS");
    if (ctx->opt_basis_tree_min >= 0 && n >= (size_t)ctx->opt_basis_tree_min) { crs_lagrange_from_powers_tree(ctx, c, q); return; }
    ZK_REQUIRE(n <= BASIS_MAX_N, ZK_ERR_UNSUPPORTED,
               "prove: an integer-roots QAP of more than 2^16 + 2^10 gates needs the CRS zk_setup made for it (the change of basis of an uploaded CRS is O(n^2))");
    c.lag1.alloc(n); c.lag2.alloc(n); c.lagS_t1.alloc(std::max<size_t>(n - 1, 1));
    DevBuf<Fr> coef;
    poly_from_integer_roots(ctx, 1, n, coef);                    // N = t, degree n
    basis_products<Fq>(ctx, c.xi1.p, n, coef, 1, n, q.ap->w.p, c.lag1.p);
    basis_products<Fq2>(ctx, c.xi2.p, n, coef, 1, n, q.ap->w.p, c.lag2.p);
    if (n >= 2) {
        poly_from_integer_roots(ctx, n + 1, n - 1, coef);        // N_S, degree n - 1: L^S_s = wS_s N_S / (x - s) has n - 1 coefficients
        basis_products<Fq>(ctx, c.xi_t1.p, n - 1, coef, n + 1, n - 1, q.ap->ws.p, c.lagS_t1.p);
    }
    c.ap = true;
}

}  // namespace zk
