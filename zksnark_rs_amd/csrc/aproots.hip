// aproots.hip -- the sparse QAP form for the roots ASTParser actually emits: the integers 1..n
// (/root/reference/src/groth16/circuit/mod.rs:517), at any size.
//
// The reference interpolates every wire polynomial through those roots (QAP::from, fr.rs:140-173; Lagrange sums
// coefficient_poly.rs:159-200: O(nnz n^2)), multiplies and divides in coefficient form (mod.rs:277) and takes inner products
// with the CRS's powers [x^i] (mod.rs:255-290).  The dense device form does the same and stops at 16384 gates (3 m n field
// elements).  This form never leaves the EVALUATION basis:
//   * U = sum a_i u_i and V are kept as their values on R = {1..n} (the SpMV output).  A = U(x) G is the inner product of those
//     values with the Lagrange-basis points [L_k(x)]_1 -- public linear combinations of the reference's [x^i]_1 (the same CRS
//     in another basis), which zk_setup emits directly since it knows x; likewise B with [L_k(x)]_2.
//   * h = (U V - W) div t has degree <= n - 2 and is kept as its values on S = {n+1 .. 2n-1}:
//         U V - W = h t + rem,  rem = (U V - W) mod t = the interpolant of U_k V_k - W_k on R
//     so on S:  h(s) = (U(s) V(s) - E(s)) / t(s)  with E = the interpolant of the products U_k V_k on R  (W + rem = E: both
//     are interpolants on R).  W is never evaluated, and h is the reference's quotient for EVERY witness (an unsatisfying one has
//     rem != 0, which E carries).  The inner product sum_j h_j [x^j t(x)/delta]_1 becomes sum_s h(s) [L^S_s(x) t(x)/delta]_1.
//   * Values at S from values at R (an arithmetic-progression shift): with the barycentric weights of consecutive integers,
//         F(s) = N(s) sum_k w_k F_k / (s - k),   w_k = (-1)^(n-k) / ((k-1)! (n-k)!),   N(s) = prod_j (s - j) = (s-1)! / (s-n-1)!
//     the sum is ONE cyclic convolution with the fixed sequence 1/d (NTT size >= 2n - 2): three forward and three inverse
//     transforms per proof for U, V, E.  t(s) = N(s), so h(s) = N(s) c_U(s) c_V(s) - c_E(s) with c_F the convolution outputs.
// Same group elements as the reference's coefficient-form proof, hence the same 259 bytes; O(n log n) per proof.
#include <algorithm>
#include <vector>
#include "pipeline.hpp"
#include "qap_kernels.hpp"

namespace zk {

// ---- per-size tables -------------------------------------------------------------------------------------------
static void host_factorials(size_t count, std::vector<Fr>& fact, std::vector<Fr>& ifact) {
    fact.resize(count); ifact.resize(count);
    fact[0] = Fr::one();
    for (size_t j = 1; j < count; ++j) fact[j] = fact[j - 1] * host_fr_from_u64(j);
    ifact[count - 1] = fact[count - 1].inv();
    for (size_t j = count - 1; j >= 1; --j) ifact[j - 1] = ifact[j] * host_fr_from_u64(j);
}

__global__ void k_ap_seq_inverse(const Fr* __restrict__ fact, const Fr* __restrict__ ifact, size_t count, size_t m, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    out[i] = i < count ? fact[i] * ifact[i + 1] : Fr::zero();   // 1 / (i + 1) = i! / (i + 1)!
}

void ap_build_tables(zk_ctx* ctx, zk_qap& q) {
    const size_t n = q.n;
    auto t = std::make_shared<ApTables>();
    t->n = n;
    std::vector<Fr> fact, ifact;
    host_factorials(2 * n + 1, fact, ifact);
    std::vector<Fr> w(n), ntab(std::max<size_t>(n - 1, 1)), ws(std::max<size_t>(n - 1, 1));
    for (size_t k = 1; k <= n; ++k) {
        Fr v = ifact[k - 1] * ifact[n - k];
        w[k - 1] = ((n - k) & 1) ? -v : v;
    }
    for (size_t i = 0; i + 1 < n; ++i) {
        const size_t s = n + 1 + i;
        ntab[i] = fact[s - 1] * ifact[s - n - 1];                   // N(s) = t(s)
        Fr v = ifact[i] * ifact[n - 2 - i];                         // weights of S itself (n - 1 consecutive integers)
        ws[i] = ((n - 2 - i) & 1) ? -v : v;
    }
    t->fact.alloc(2 * n + 1); t->ifact.alloc(2 * n + 1); t->w.alloc(n); t->ntab.alloc(ntab.size()); t->ws.alloc(ws.size());
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemcpyAsync(t->fact.p, fact.data(), fact.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(t->ifact.p, ifact.data(), ifact.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(t->w.p, w.data(), n * sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(t->ntab.p, ntab.data(), ntab.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
    ZK_HIP(hipMemcpyAsync(t->ws.p, ws.data(), ws.size() * sizeof(Fr), hipMemcpyHostToDevice, st));
    // cyclic convolution size: the wanted outputs t = s - 2 in [n-1, 2n-3] must not alias, M >= 2n - 2
    unsigned lg = 1;
    while (((size_t)1 << lg) < std::max<size_t>(2 * n - 2, 2)) ++lg;
    ZK_REQUIRE(lg <= NTT_MAX_LOG, ZK_ERR_SIZE, "integer-roots QAP: too many gates for the NTT (2n - 2 <= 2^24)");
    t->log_m = lg;
    const size_t M = (size_t)1 << lg;
    t->bhat.alloc(M);
    if (n >= 2) {
        hipLaunchKernelGGL(k_ap_seq_inverse, dim3(ceil_div(M, 256)), dim3(256), 0, st, t->fact.p, t->ifact.p, 2 * n - 2, M, t->bhat.p);
        ZK_HIP(hipGetLastError());
        ntt_dif(ctx, t->bhat.p, lg, false, false);
    }
    ZK_HIP(hipStreamSynchronize(st));
    t->fact.release(); t->ifact.release();   // only the construction above reads them (64 B per gate)
    q.ap = t;
}

// ---- setup: Lagrange-basis values at x -----------------------------------------------------------------------------
// L[k-1] = t(x) w_k / (x - k), k in R;   LS[i] = N_S(x) wS_i / (x - s_i) * t(x) / delta, s_i = n + 1 + i
__global__ void k_ap_lagrange(Fr x, Fr tx, Fr nsx_tx_dinv, const Fr* __restrict__ w, const Fr* __restrict__ ws, size_t n,
                              Fr* __restrict__ L, Fr* __restrict__ LS, int* __restrict__ flag) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * n - 1) return;
    Fr kk = Fr::zero();
    kk.l[0] = (uint32_t)(j + 1);
    kk.l[1] = (uint32_t)((uint64_t)(j + 1) >> 32);
    Fr den = x - Fr::from_canonical(kk);
    if (den.is_zero()) { atomicOr(flag, 8); return; }
    Fr inv = den.inv();
    if (j < n) L[j] = tx * w[j] * inv;
    else LS[j - n] = nsx_tx_dinv * ws[j - n] * inv;
}

void ap_setup_lagrange(zk_ctx* ctx, const zk_qap& q, const uint64_t trapdoor[20], Fr* d_L, Fr* d_LS, int* d_flag) {
    const size_t n = q.n;
    Fr xc, dc;
    for (int i = 0; i < 4; ++i) {
        xc.l[2 * i] = (uint32_t)trapdoor[16 + i]; xc.l[2 * i + 1] = (uint32_t)(trapdoor[16 + i] >> 32);
        dc.l[2 * i] = (uint32_t)trapdoor[12 + i]; dc.l[2 * i + 1] = (uint32_t)(trapdoor[12 + i] >> 32);
    }
    const Fr x = Fr::from_canonical(xc), dinv = Fr::from_canonical(dc).inv();
    Fr tx = Fr::one(), nsx = Fr::one();
    for (size_t k = 1; k <= n; ++k) tx = tx * (x - host_fr_from_u64(k));
    for (size_t s = n + 1; s <= 2 * n - 1; ++s) nsx = nsx * (x - host_fr_from_u64(s));
    hipLaunchKernelGGL(k_ap_lagrange, dim3(ceil_div(2 * n - 1, 256)), dim3(256), 0, ctx->stream, x, tx, nsx * tx * dinv, q.ap->w.p, q.ap->ws.p, n, d_L, d_LS, d_flag);
    ZK_HIP(hipGetLastError());
}
Fr ap_t_at_x(const zk_qap& q, const uint64_t trapdoor[20]) {
    Fr xc;
    for (int i = 0; i < 4; ++i) { xc.l[2 * i] = (uint32_t)trapdoor[16 + i]; xc.l[2 * i + 1] = (uint32_t)(trapdoor[16 + i] >> 32); }
    const Fr x = Fr::from_canonical(xc);
    Fr tx = Fr::one();
    for (size_t k = 1; k <= q.n; ++k) tx = tx * (x - host_fr_from_u64(k));
    return tx;
}

// ---- prove: the scalars of the four inner products ----------------------------------------------------------------------
// Batch form: blockIdx.y = proof j; its SpMV outputs sit at ue + j n / ve + j n, its three transforms at work + 3 j M.
// buf[0] = w . Ue, buf[1] = w . Ve, buf[2] = w . (Ue . Ve), zero padded to M
__global__ void k_ap_prep(const Fr* __restrict__ ue, const Fr* __restrict__ ve, const Fr* __restrict__ w, size_t n, size_t M, Fr* __restrict__ buf) {
    ZK_LATENCY_KERNEL();
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    ue += (size_t)blockIdx.y * n; ve += (size_t)blockIdx.y * n; buf += (size_t)blockIdx.y * 3 * M;
    if (j < n) {
        const Fr u = ue[j], v = ve[j], wj = w[j];
        buf[j] = wj * u;
        buf[M + j] = wj * v;
        buf[2 * M + j] = wj * (u * v);
    } else {
        buf[j] = buf[M + j] = buf[2 * M + j] = Fr::zero();
    }
}
__global__ void k_ap_mul_bhat(Fr* __restrict__ buf, const Fr* __restrict__ bhat, size_t M, size_t total) {
    ZK_LATENCY_KERNEL();
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    buf[j] = buf[j] * bhat[j & (M - 1)];
}
// h(s_i) = N(s_i) c_U c_V - c_E with c_F = buf_F[n - 1 + i]  ->  canonical, proof j at out + j out_stride
__global__ void k_ap_h(const Fr* __restrict__ buf, const Fr* __restrict__ ntab, size_t n, size_t M, Fr* __restrict__ out, size_t out_stride) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n) return;
    buf += (size_t)blockIdx.y * 3 * M; out += (size_t)blockIdx.y * out_stride;
    const size_t t = n - 1 + i;
    out[i] = (ntab[i] * buf[t] * buf[M + t] - buf[2 * M + t]).to_canonical();
}

// ue / ve: the SpMV outputs (`count` x n each); writes the n - 1 values of h of proof j to hb_can + j hb_stride (canonical);
// `work` holds 3 M count elements
void ap_quotient_values(zk_ctx* ctx, const zk_qap& q, const Fr* ue, const Fr* ve, Fr* work, Fr* hb_can, size_t count, size_t hb_stride) {
    const size_t n = q.n;
    if (n < 2) return;
    const ApTables& t = *q.ap;
    const size_t M = (size_t)1 << t.log_m;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_ap_prep, dim3(ceil_div(M, 256), count), dim3(256), 0, st, ue, ve, t.w.p, n, M, work);
    ntt_dif(ctx, work, t.log_m, false, false, 3 * count);
    hipLaunchKernelGGL(k_ap_mul_bhat, dim3(ceil_div(3 * M * count, 256)), dim3(256), 0, st, work, t.bhat.p, M, 3 * M * count);
    ntt_dit(ctx, work, t.log_m, true, true, nullptr, 3 * count);
    hipLaunchKernelGGL(k_ap_h, dim3(ceil_div(n - 1, 256), count), dim3(256), 0, st, work, t.ntab.p, n, M, hb_can, hb_stride);
    ZK_HIP(hipGetLastError());
}

// ---- upload -------------------------------------------------------------------------------------------------------
zk_qap* qap_upload_sparse_integers(zk_ctx* ctx, const zk_qap_sparse_desc& desc, size_t n) {
    ZK_REQUIRE(n >= 1 && n <= ((size_t)1 << (NTT_MAX_LOG - 1)), ZK_ERR_SIZE, "integer-roots QAP: n must be in [1, 2^23]");
    zk_qap* q = qap_upload_rows(ctx, desc, n);
    std::unique_ptr<zk_qap> guard(q);
    q->roots = 1;
    q->log_n = 0;
    ap_build_tables(ctx, *q);
    return guard.release();
}

}  // namespace zk
