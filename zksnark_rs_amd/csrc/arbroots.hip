// arbroots.hip -- the sparse QAP form for ARBITRARY distinct roots (RootRepresentation::roots() is caller data,
// /root/reference/src/groth16/circuit/mod.rs:201-214, dummy_rep.rs:47), at any size.
//
// The reference interpolates all 3 m wire polynomials through the roots (QAP::from, fr.rs:140-173; Lagrange sums
// coefficient_poly.rs:159-200: O(nnz n^2)) and then proves in coefficient form.  The dense device form does the same and stops at
// 16384 gates (3 m n field elements).  Here the rows stay as they are and the PROVER interpolates, per proof, the three polynomials
// it needs -- U = sum a_i u_i, V, and E = the interpolant of the products U_k V_k -- from their values on the roots (the SpMV output)
// by the sub-product tree of interp.hip, O(n log^2 n):
//   * A, B, r B1 + s A:  inner products of the coefficients of U, V with the reference's own [x^i]_1, [x^i]_2 -- ANY CRS serves (zk_setup,
//     zk_crs_upload, a file), nothing is held in a Lagrange basis;
//   * h = (U V - W) div t:  U V - W = h t + rem with rem = E - W (both W and E have degree < n and E interpolates U_k V_k = (U V)(r_k)),
//     so h = (U V - E) / t exactly, for EVERY witness (aproots.hip has the same argument).  U, V, E are evaluated on a coset g <w> of
//     2^k >= n points that misses every root, h's values are (U V - E) / t there, one inverse transform gives its n - 1 coefficients.
//     W is never evaluated.
// Same group elements as the reference's proof, hence the same 259 bytes.  Per root set, once: the tree (interp.hip, O(n^2) multiplications
// for N'(r_k)), t = prod (x - r_k), 1 / t on the coset.
#include <algorithm>
#include <vector>
#include "pipeline.hpp"
#include "qap_kernels.hpp"

namespace zk {

// ---- per-QAP tables -------------------------------------------------------------------------------------------------------
// buf[j] = t_j g^j (j < M; t_j = 0 beyond the degree; the leading 1 of a degree-M t is added after the transform)
__global__ void k_arb_t_scaled(const Fr* __restrict__ t, size_t n, size_t M, const Fr* __restrict__ gpow, Fr* __restrict__ buf) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    buf[j] = j <= n ? t[j] * gpow[j] : Fr::zero();
}
// buf[j] <- 1 / (buf[j] + top); flag |= 32 when t vanishes on the coset
__global__ void k_arb_t_invert(Fr* __restrict__ buf, size_t M, Fr top, int* __restrict__ flag) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const Fr v = buf[j] + top;
    if (v.is_zero()) { atomicOr(flag, 32); return; }
    buf[j] = v.inv();
}

static void arb_build_tables(zk_ctx* ctx, zk_qap& q, const Fr* d_roots_mont, int* d_flag) {
    const size_t n = q.n;
    hipStream_t st = ctx->stream;
    auto a = std::make_shared<ArbTables>();
    a->tree = interp_build(ctx, d_roots_mont, n, d_flag);
    {
        int h = 0;
        ZK_HIP(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        ZK_REQUIRE(!(h & 16), ZK_ERR_ARG, "arbitrary-roots QAP: the roots are not distinct");
    }
    a->log_m = a->tree->log_npad;
    const size_t M = (size_t)1 << a->log_m;
    a->host_roots.resize(n);
    ZK_HIP(hipMemcpyAsync(a->host_roots.data(), a->tree->roots.p, n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    a->gpow.alloc(M); a->ginv_pow.alloc(M); a->tinv.alloc(M);
    // the coset g <w_M>, g = 5^e for the first odd e whose coset misses every root (5 generates Fr*: 5^e with e odd is outside <w_M>)
    const Fr five = host_fr_from_u64(5);
    bool found = false;
    for (uint64_t e = 1; e <= 15 && !found; e += 2) {
        const Fr g = host_fr_pow(five, e);
        fr_powers(ctx, g, Fr::one(), a->gpow.p, M);
        hipLaunchKernelGGL(k_arb_t_scaled, dim3(ceil_div(M, 256)), dim3(256), 0, st, a->tree->t.p, n, M, a->gpow.p, a->tinv.p);
        ntt_dif(ctx, a->tinv.p, a->log_m, false, false);
        // n == M: t has M + 1 coefficients; x^M is the constant g^M on the coset
        const Fr top = n == M ? host_fr_pow(g, M) : Fr::zero();
        ZK_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_arb_t_invert, dim3(ceil_div(M, 256)), dim3(256), 0, st, a->tinv.p, M, top, d_flag);
        ZK_HIP(hipGetLastError());
        int h = 0;
        ZK_HIP(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        if (!(h & 32)) {
            found = true;
            fr_powers(ctx, g.inv(), Fr::one(), a->ginv_pow.p, M);
        }
    }
    ZK_REQUIRE(found, ZK_ERR_UNSUPPORTED, "arbitrary-roots QAP: every candidate evaluation coset contains a root");
    ZK_HIP(hipStreamSynchronize(st));
    q.arb = a;
}

zk_qap* qap_upload_sparse_roots(zk_ctx* ctx, const zk_qap_sparse_desc& desc, const uint64_t* roots, size_t n) {
    ZK_REQUIRE(roots, ZK_ERR_ARG, "arbitrary-roots QAP: null roots");
    ZK_REQUIRE(n >= 1 && n <= ((size_t)1 << (NTT_MAX_LOG - 2)), ZK_ERR_SIZE, "arbitrary-roots QAP: n must be in [1, 2^22]");
    zk_qap* q = qap_upload_rows(ctx, desc, n);
    std::unique_ptr<zk_qap> guard(q);
    q->roots = 2;
    q->log_n = 0;
    DevBuf<Fr> r(n);
    DevBuf<int> flag(1);
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    ZK_HIP(hipMemcpyAsync(r.p, roots, n * sizeof(Fr), hipMemcpyHostToDevice, st));
    fr_to_mont(ctx, r.p, r.p, n, flag.p);
    {
        int h = 0;
        ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        ZK_REQUIRE(!h, ZK_ERR_RANGE, "arbitrary-roots QAP: root >= r");
    }
    arb_build_tables(ctx, *q, r.p, flag.p);
    return guard.release();
}

void arb_download_roots(zk_ctx* ctx, const zk_qap& q, uint64_t* out) {
    DevBuf<Fr> tmp(q.n);
    fr_from_mont(ctx, q.arb->tree->roots.p, tmp.p, q.n);
    ZK_HIP(hipMemcpyAsync(out, tmp.p, q.n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
}

// ---- setup: Lagrange-basis values at x -------------------------------------------------------------------------------------
// L[k] = t(x) w_k / (x - r_k)
__global__ void k_arb_lagrange(Fr x, Fr tx, const Fr* __restrict__ r, const Fr* __restrict__ w, size_t n, Fr* __restrict__ L, int* __restrict__ flag) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Fr den = x - r[k];
    if (den.is_zero()) { atomicOr(flag, 8); return; }
    L[k] = tx * w[k] * den.inv();
}
static Fr arb_x(const uint64_t trapdoor[20]) {
    Fr xc;
    for (int i = 0; i < 4; ++i) { xc.l[2 * i] = (uint32_t)trapdoor[16 + i]; xc.l[2 * i + 1] = (uint32_t)(trapdoor[16 + i] >> 32); }
    return Fr::from_canonical(xc);
}
Fr arb_t_at_x(const zk_qap& q, const uint64_t trapdoor[20]) {
    const Fr x = arb_x(trapdoor);
    Fr tx = Fr::one();
    for (const Fr& r : q.arb->host_roots) tx = tx * (x - r);
    return tx;
}
void arb_setup_lagrange(zk_ctx* ctx, const zk_qap& q, const uint64_t trapdoor[20], Fr* d_L, int* d_flag) {
    const InterpTree& t = *q.arb->tree;
    hipLaunchKernelGGL(k_arb_lagrange, dim3(ceil_div(q.n, 256)), dim3(256), 0, ctx->stream, arb_x(trapdoor), arb_t_at_x(q, trapdoor), t.roots.p, t.w.p, q.n, d_L, d_flag);
    ZK_HIP(hipGetLastError());
}

// ---- prove: the scalars of A, B, H + r B1 + s A ------------------------------------------------------------------------------
__global__ void k_arb_products(const Fr* __restrict__ ue, const Fr* __restrict__ ve, size_t n, Fr* __restrict__ ee) {
    ZK_LATENCY_KERNEL();
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) ee[j] = ue[j] * ve[j];
}
// three coefficient vectors (M apart) -> scaled by g^j for the coset transform
__global__ void k_arb_scale3(Fr* __restrict__ c, const Fr* __restrict__ gpow, size_t M) {
    ZK_LATENCY_KERNEL();
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 3 * M) return;
    c[j] = c[j] * gpow[j & (M - 1)];
}
// h on the coset: (U V - E) / t, into the first vector
__global__ void k_arb_h(Fr* __restrict__ c, const Fr* __restrict__ tinv, size_t M) {
    ZK_LATENCY_KERNEL();
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    c[j] = (c[j] * c[M + j] - c[2 * M + j]) * tinv[j];
}
// coefficients of h: the inverse transform's output times g^-j -> canonical
__global__ void k_arb_h_out(const Fr* __restrict__ c, const Fr* __restrict__ ginv_pow, size_t count, Fr* __restrict__ out) {
    ZK_LATENCY_KERNEL();
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) out[j] = (c[j] * ginv_pow[j]).to_canonical();
}

size_t arb_work_elems(const zk_qap& q) { return (size_t)12 << q.arb->log_m; }

// ue, ve: the SpMV outputs (values of U, V on the roots, Montgomery; 3 n elements available at ue: ue | ve | products).  Writes the
// canonical coefficients of V to vc_can (n), of U to uc_can (n), r V + s U to hb_can + (n - 1) (n) and h to hb_can (n - 1).
// work: arb_work_elems(q) elements.
void arb_scalars(zk_ctx* ctx, const zk_qap& q, Fr* vals, Fr* work, const Fr& r_mont, const Fr& s_mont, Fr* vc_can, Fr* uc_can, Fr* hb_can) {
    const ArbTables& a = *q.arb;
    const size_t n = q.n, M = (size_t)1 << a.log_m;
    hipStream_t st = ctx->stream;
    Fr *ue = vals, *ve = vals + n, *ee = vals + 2 * n;
    hipLaunchKernelGGL(k_arb_products, dim3(ceil_div(n, 256)), dim3(256), 0, st, ue, ve, n, ee);
    Fr* coef = work;                   // 3 M: U | V | E
    interp_run(ctx, *a.tree, vals, n, 3, work + 3 * M, coef);
    fr_from_mont(ctx, coef, uc_can, n);
    fr_from_mont(ctx, coef + M, vc_can, n);
    fr_lincomb_to_canonical(ctx, coef + M, r_mont, coef, s_mont, hb_can + (n - 1), n);
    if (n >= 2) {
        hipLaunchKernelGGL(k_arb_scale3, dim3(ceil_div(3 * M, 256)), dim3(256), 0, st, coef, a.gpow.p, M);
        ntt_dif(ctx, coef, a.log_m, false, false, 3);
        hipLaunchKernelGGL(k_arb_h, dim3(ceil_div(M, 256)), dim3(256), 0, st, coef, a.tinv.p, M);
        ntt_dit(ctx, coef, a.log_m, true, true, nullptr, 1);
        hipLaunchKernelGGL(k_arb_h_out, dim3(ceil_div(n - 1, 256)), dim3(256), 0, st, coef, a.ginv_pow.p, n - 1, hb_can);
    }
    ZK_HIP(hipGetLastError());
}

}  // namespace zk
