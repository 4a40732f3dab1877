// arbroots.hip -- the sparse QAP form for ARBITRARY distinct roots (RootRepresentation::roots() is caller data,
// /root/reference/src/groth16/circuit/mod.rs:201-214, dummy_rep.rs:47), at any size.
//
// The reference interpolates all 3 m wire polynomials through the roots (QAP::from, fr.rs:140-173; Lagrange sums
// coefficient_poly.rs:159-200: O(nnz n^2)) and then proves in coefficient form.  The dense device form does the same and stops at
// 16384 gates (3 m n field elements).  Here the rows stay as they are and the PROVER interpolates, per proof, the two polynomials
// it needs -- U = sum a_i u_i and V -- from their values on the roots (the SpMV output) by the sub-product tree of interp.hip,
// O(n log^2 n):
//   * A, B, r B1 + s A:  inner products of the coefficients of U, V with the reference's own [x^i]_1, [x^i]_2 -- ANY CRS serves (zk_setup,
//     zk_crs_upload, a file), nothing is held in a Lagrange basis;
//   * h = (U V - W) div t:  U V - W = h t + rem with rem = E - W, where E is the interpolant of the products U_k V_k = (U V)(r_k): both
//     W and E have degree < n, so U V = h t + E and the quotient of U V ALONE by t is h, for EVERY witness (aproots.hip has the same
//     argument).  W is never evaluated; the product and the division by t (monic, degree n; Newton's iteration on rev(t) once per QAP,
//     long division below 512 quotient coefficients) are the coefficient-form tail the dense form already has (prove.hip, qap.hip).
// Same group elements as the reference's proof, hence the same 259 bytes.  Per root set, once, O(n log^2 n) too: the tree with its
// weights 1 / N'(r_k) (interp.hip), t = prod (x - r_k) and the power-series inverse of rev(t).
#include <algorithm>
#include <vector>
#include "pipeline.hpp"
#include "qap_kernels.hpp"

namespace zk {

// ---- per-QAP tables -------------------------------------------------------------------------------------------------------
static void arb_build_tables(zk_ctx* ctx, zk_qap& q, const Fr* d_roots_mont, int* d_flag) {
    const size_t n = q.n;
    hipStream_t st = ctx->stream;
    auto a = std::make_shared<ArbTables>();
    a->tree = interp_build(ctx, d_roots_mont, n, d_flag);
    int h = 0;
    ZK_HIP(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    a->host_roots.resize(n);
    ZK_HIP(hipMemcpyAsync(a->host_roots.data(), a->tree->roots.p, n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    // t = prod (x - r_k): monic of degree n -- the divisor of the coefficient-form tail the dense form shares (prove.hip)
    q.dt.alloc(n + 1);
    ZK_HIP(hipMemcpyAsync(q.dt.p, a->tree->t.p, (n + 1) * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    q.t_degree = n;
    q.t_is_zero = false;
    q.t_cinv.alloc(1);
    const Fr one = Fr::one();
    ZK_HIP(hipMemcpyAsync(q.t_cinv.p, &one, sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_REQUIRE(!(h & 16), ZK_ERR_ARG, "arbitrary-roots QAP: the roots are not distinct");
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

// An integer-roots QAP (aproots.hip) whose CRS carries only the reference's powers and is too large for the change of basis
// (basis.hip: O(n^2) group operations) proves in THIS form instead: the roots 1..n as caller data.  Built on first use.
__global__ void k_arb_integers(size_t n, Fr* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fr v = Fr::zero();
    v.l[0] = (uint32_t)(k + 1); v.l[1] = (uint32_t)((uint64_t)(k + 1) >> 32);
    out[k] = Fr::from_canonical(v);
}
void arb_attach_integer_roots(zk_ctx* ctx, zk_qap& q) {
    if (q.arb) return;
    ZK_REQUIRE(q.n <= ((size_t)1 << (NTT_MAX_LOG - 2)), ZK_ERR_UNSUPPORTED,
               "prove: an integer-roots QAP of more than 2^22 gates needs the CRS zk_setup made for it");
    DevBuf<Fr> r(q.n);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_arb_integers, dim3(ceil_div(q.n, 256)), dim3(256), 0, ctx->stream, q.n, r.p);
    ZK_HIP(hipGetLastError());
    arb_build_tables(ctx, q, r.p, flag.p);
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

// ---- prove: the coefficients of U and V ----------------------------------------------------------------------------------------
size_t arb_work_elems(const zk_qap& q) { return (size_t)8 << q.arb->tree->log_npad; }

// vals: the SpMV outputs U_k | V_k (values on the roots, Montgomery, n each) -> uc, vc: the n coefficients of the interpolants
// (Montgomery).  work: arb_work_elems(q) elements.
void arb_coefficients(zk_ctx* ctx, const zk_qap& q, const Fr* vals, Fr* work, Fr* uc, Fr* vc) {
    const InterpTree& t = *q.arb->tree;
    const size_t n = q.n, npad = (size_t)1 << t.log_npad;
    hipStream_t st = ctx->stream;
    Fr* coef = work;                   // 2 npad: U | V
    interp_run(ctx, t, vals, n, 2, work + 2 * npad, coef);
    ZK_HIP(hipMemcpyAsync(uc, coef, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ZK_HIP(hipMemcpyAsync(vc, coef + npad, n * sizeof(Fr), hipMemcpyDeviceToDevice, st));
}

}  // namespace zk
