// qap.hip -- device-resident QAP handles and the kernels that evaluate the QAP against a witness.
//
// Replaces, on groth16::prove's path (/root/reference/src/groth16/mod.rs:233-253,277):
//   u_sum = sum_i qap.u[i] * weights[i]  (CoefficientPoly Mul<T> + Sum, coefficient_poly.rs:75-91,132-146)
//   h = (u_sum * v_sum - w_sum) / qap.t  (coefficient_poly.rs:93-157, field/mod.rs:428-469)
// Two device forms:
//   sparse : rows of (gate, value) over the domain w^j; the weighted sum is one CSR SpMV per
//            matrix producing EVALUATIONS, interpolation is an inverse NTT.  Dense per-wire
//            polynomials are never built.
//   dense  : the literal fields of QAP<CoefficientPoly<FrLocal>> for small circuits with
//            arbitrary roots (e.g. ASTParser's 1..n); weighted sum is a dense mat-vec, the product
//            an NTT convolution, the division by t a long division in one workgroup.
#include <algorithm>
#include <numeric>
#include "pipeline.hpp"
#include "qap_kernels.hpp"
#include "fr_tile.cuh"

namespace zk {

// ---- sparse upload -------------------------------------------------------------------------
// `twice`: the values are stored as val R (the Montgomery form of the Montgomery form): k_spmv multiplies them with the CANONICAL
// witness and gets Montgomery-form sums, so that a proof never converts its witness (one multiplication per wire and a pass over it)
static void upload_csr(zk_ctx* ctx, DevCsr& d, const std::vector<uint32_t>& ptr, const std::vector<uint32_t>& idx,
                       const std::vector<uint64_t>& val_words, int* d_flag, bool twice = false) {
    d.rows = ptr.size() - 1;
    d.nnz = idx.size();
    d.ptr.alloc(ptr.size());
    d.idx.alloc(std::max<size_t>(idx.size(), 1));
    d.val.alloc(std::max<size_t>(idx.size(), 1));
    ZK_HIP(hipMemcpyAsync(d.ptr.p, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (d.nnz) {
        ZK_HIP(hipMemcpyAsync(d.idx.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(hipMemcpyAsync(d.val.p, val_words.data(), d.nnz * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        fr_to_mont(ctx, d.val.p, d.val.p, d.nnz, d_flag);
        if (twice) fr_to_mont(ctx, d.val.p, d.val.p, d.nnz, d_flag);
    }
    ZK_HIP(hipStreamSynchronize(ctx->stream));  // host vectors may go out of scope
}

// by-wire rows (as given) -> device; optionally also the by-gate transpose
static void upload_rows(zk_ctx* ctx, const zk_sparse_rows& rows, size_t m, size_t n, DevCsr* by_wire, DevCsr* by_gate, int* d_flag) {
    ZK_REQUIRE(rows.ptr, ZK_ERR_ARG, "sparse rows: null ptr array");
    size_t nnz = rows.ptr[m];
    ZK_REQUIRE(nnz < ((size_t)1 << 32), ZK_ERR_SIZE, "sparse rows: too many entries");
    ZK_REQUIRE(nnz == 0 || (rows.gate && rows.val), ZK_ERR_ARG, "sparse rows: null gate/val array");
    std::vector<uint32_t> ptr(m + 1), idx(rows.gate, rows.gate + nnz);
    for (size_t i = 0; i <= m; ++i) {
        ZK_REQUIRE(rows.ptr[i] <= nnz && (i == 0 || rows.ptr[i] >= rows.ptr[i - 1]), ZK_ERR_ARG, "sparse rows: ptr not monotone");
        ptr[i] = (uint32_t)rows.ptr[i];
    }
    for (size_t k = 0; k < nnz; ++k) ZK_REQUIRE(idx[k] < n, ZK_ERR_ARG, "sparse rows: gate index out of range");
    std::vector<uint64_t> val(rows.val, rows.val + nnz * 4);
    if (by_wire) upload_csr(ctx, *by_wire, ptr, idx, val, d_flag);
    if (by_gate) {
        // counting-sort transpose: rows = gates, columns = wires
        std::vector<uint32_t> gptr(n + 1, 0), gidx(nnz);
        std::vector<uint64_t> gval(nnz * 4);
        for (size_t k = 0; k < nnz; ++k) ++gptr[idx[k] + 1];
        for (size_t j = 0; j < n; ++j) gptr[j + 1] += gptr[j];
        std::vector<uint32_t> cur(gptr.begin(), gptr.end() - 1);
        for (size_t i = 0; i < m; ++i)
            for (size_t k = ptr[i]; k < ptr[i + 1]; ++k) {
                uint32_t pos = cur[idx[k]]++;
                gidx[pos] = (uint32_t)i;
                std::copy(val.begin() + 4 * k, val.begin() + 4 * k + 4, gval.begin() + 4 * (size_t)pos);
            }
        upload_csr(ctx, *by_gate, gptr, gidx, gval, d_flag, true);
    }
}

static void check_flag(zk_ctx* ctx, int* d_flag, const char* what) {
    int h = 0;
    ZK_HIP(hipMemcpy(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost));
    ZK_REQUIRE(!(h & 2), ZK_ERR_RANGE, std::string(what) + ": field element >= modulus");
    (void)ctx;
}

zk_qap* qap_upload_rows(zk_ctx* ctx, const zk_qap_sparse_desc& desc, size_t n) {
    ZK_REQUIRE(desc.m >= 1 && desc.input < desc.m && desc.m < ((size_t)1 << 31), ZK_ERR_ARG, "sparse QAP: need input < m");
    std::unique_ptr<zk_qap> q(new zk_qap());
    q->ctx = ctx;
    q->dense = false;
    q->n = n;
    q->m = desc.m;
    q->input = desc.input;
    DevBuf<int> flag(1);
    ZK_HIP(hipMemset(flag.p, 0, sizeof(int)));
    upload_rows(ctx, desc.u, q->m, q->n, &q->u_wire, &q->u_gate, flag.p);
    upload_rows(ctx, desc.v, q->m, q->n, &q->v_wire, &q->v_gate, flag.p);
    upload_rows(ctx, desc.w, q->m, q->n, &q->w_wire, nullptr, flag.p);
    check_flag(ctx, flag.p, "zk_qap_upload_sparse");
    return q.release();
}
zk_qap* qap_upload_sparse(zk_ctx* ctx, const zk_qap_sparse_desc& desc) {
    ZK_REQUIRE(desc.log_n <= NTT_MAX_LOG - 1, ZK_ERR_SIZE, "sparse QAP: log_n too large");
    zk_qap* q = qap_upload_rows(ctx, desc, (size_t)1 << desc.log_n);
    q->log_n = desc.log_n;
    return q;
}

__global__ void k_inv_single(const Fr* in, Fr* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = in->inv();
}

zk_qap* qap_upload_dense(zk_ctx* ctx, const uint64_t* u, const uint64_t* v, const uint64_t* w, const uint64_t* t, size_t m, size_t n, size_t input) {
    ZK_REQUIRE(u && v && w && t, ZK_ERR_ARG, "dense QAP: null pointer");
    ZK_REQUIRE(m >= 1 && n >= 1 && input < m, ZK_ERR_ARG, "dense QAP: need n >= 1 and input < m");
    unsigned log_conv = 1;
    while (((size_t)1 << log_conv) < 2 * n) ++log_conv;
    ZK_REQUIRE(log_conv <= NTT_MAX_LOG, ZK_ERR_SIZE, "dense QAP: n too large");
    std::unique_ptr<zk_qap> q(new zk_qap());
    q->ctx = ctx;
    q->dense = true;
    q->n = n;
    q->m = m;
    q->input = input;
    DevBuf<int> flag(1);
    ZK_HIP(hipMemset(flag.p, 0, sizeof(int)));
    auto up = [&](DevBuf<Fr>& d, const uint64_t* src, size_t count) {
        d.alloc(count);
        ZK_HIP(hipMemcpyAsync(d.p, src, count * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        fr_to_mont(ctx, d.p, d.p, count, flag.p);
    };
    up(q->du, u, m * n);
    up(q->dv, v, m * n);
    up(q->dw, w, m * n);
    up(q->dt, t, n + 1);
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    check_flag(ctx, flag.p, "zk_qap_upload_dense");
    // degree of t as Polynomial::degree (field/mod.rs:291-297)
    size_t d = n + 1;
    while (d > 0) {
        const uint64_t* c = t + 4 * (d - 1);
        if (c[0] | c[1] | c[2] | c[3]) break;
        --d;
    }
    q->t_is_zero = d == 0;
    q->t_degree = d == 0 ? 0 : d - 1;
    q->t_cinv.alloc(1);
    hipLaunchKernelGGL(k_inv_single, dim3(1), dim3(64), 0, ctx->stream, q->dt.p + q->t_degree, q->t_cinv.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    return q.release();
}

void qap_free(zk_qap* q) {
    if (!q) return;
    (void)hipSetDevice(q->ctx->device);
    delete q;
}

// ---- kernels -------------------------------------------------------------------------------
// out[row] = sum_k a[idx[k]] * val[k]   (one lane per row; rows of the chain circuit have 1-2 entries).  `a` = the witness as the caller
// gave it (CANONICAL integers), val = the coefficients times R (upload_csr, twice): the products are the Montgomery forms, so the proof
// never converts its witness.  Sums stay in the multiplier's lazy radix (9 limb-wise adds + a carry pass instead of a carry chain with a
// conditional correction) and leave through one exact reduction.
__global__ void k_spmv(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ idx, const Fr* __restrict__ val,
                       const Fr* __restrict__ a, size_t a_len, Fr* __restrict__ out, size_t rows) {
    ZK_LATENCY_KERNEL();
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= rows) return;
    FrL acc = FrL::load(Fr::zero());
    uint32_t cnt = 0;
    for (uint32_t k = ptr[j]; k < ptr[j + 1]; ++k) {
        uint32_t i = idx[k];
        if (i < a_len) {   // zip(weights) truncates (mod.rs:233-253)
            acc = (acc + FrL::load(a[i]) * FrL::load(val[k])).norm();
            if ((++cnt & 63u) == 0) acc = fr_reduce(acc);   // |value| stays below 2^7 p whatever the row length
        }
    }
    out[j] = fr_store_exact(acc);
}
void spmv(zk_ctx* ctx, const DevCsr& m, const Fr* a, size_t a_len, Fr* out) {
    if (!m.rows) return;
    ProfScope ps(ctx, "qap_spmv", 36.0 * m.nnz + 4.0 * (m.rows + 1) + 32.0 * (m.nnz + m.rows));
    hipLaunchKernelGGL(k_spmv, dim3(ceil_div(m.rows, 256)), dim3(256), 0, ctx->stream, m.ptr.p, m.idx.p, m.val.p, a, a_len, out, m.rows);
    ZK_HIP(hipGetLastError());
}

// dense: out[k] = sum_{i < rows} a[i] * M[i*n + k]
__global__ void k_dense_matvec(const Fr* __restrict__ M, const Fr* __restrict__ a, size_t rows, size_t n, Fr* __restrict__ out) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fr acc = Fr::zero();
    for (size_t i = 0; i < rows; ++i) acc = acc + a[i] * M[i * n + k];
    out[k] = acc;
}
void dense_matvec(zk_ctx* ctx, const Fr* M, const Fr* a, size_t rows, size_t n, Fr* out) {
    ProfScope ps(ctx, "qap_dense_matvec", 32.0 * (rows * n + rows + n));
    hipLaunchKernelGGL(k_dense_matvec, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, M, a, rows, n, out);
    ZK_HIP(hipGetLastError());
}

// The element-wise kernels that END in canonical integers (the scalars of the inner products) take their constant factors as PLAIN
// integers: mont(x R, k) = x k is already the canonical value, so "multiply, then leave the Montgomery form" is one multiplication
// instead of two, and a r + b s is one reduction (FpR::mont_sum) instead of three.  The launchers keep their Montgomery-form arguments
// and convert on the host.  All in the multiplier's own radix (lazy29.cuh); the exact reduction closes them.
// h[pos] = canonical( x[pos] * 1/2  -  tab[pos] * y[pos] ),  tab = g^-brev(pos) / 2 as plain integers (ntt_ensure_coset_tables)
__global__ void k_h_combine(const Fr* __restrict__ x, const Fr* __restrict__ y, const Fr* __restrict__ tab, Fr half_c, Fr* __restrict__ out, size_t n) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_store_exact(FrL::mont_diff(FrL::load(x[i]), FrL::load(half_c), FrL::load(tab[i]), FrL::load(y[i])));
}
void h_combine(zk_ctx* ctx, const Fr* x, const Fr* y, const Fr* tab, Fr half, Fr* out, size_t n) {
    ProfScope ps(ctx, "qap_h_combine", 128.0 * n);
    hipLaunchKernelGGL(k_h_combine, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, x, y, tab, half.to_canonical(), out, n);
    ZK_HIP(hipGetLastError());
}

// out[i] = canonical(in[i] * k)   (in Montgomery form, kc = k as a plain integer)
__global__ void k_scale_to_canonical(const Fr* __restrict__ in, Fr kc, Fr* __restrict__ out, size_t n) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_store_exact(FrL::load(in[i]) * FrL::load(kc));
}
void fr_scale_to_canonical(zk_ctx* ctx, const Fr* in, Fr k, Fr* out, size_t n) {
    if (!n) return;
    ProfScope ps(ctx, "fr_scale_to_canonical", 64.0 * n);
    hipLaunchKernelGGL(k_scale_to_canonical, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, k.to_canonical(), out, n);
    ZK_HIP(hipGetLastError());
}

// out[i] = canonical(a[i] * ka + b[i] * kb), ka / kb as plain integers
__global__ void k_lincomb_to_canonical(const Fr* __restrict__ a, Fr kac, const Fr* __restrict__ b, Fr kbc, Fr* __restrict__ out, size_t n) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fr_store_exact(FrL::mont_sum(FrL::load(a[i]), FrL::load(kac), FrL::load(b[i]), FrL::load(kbc)));
}
void fr_lincomb_to_canonical(zk_ctx* ctx, const Fr* a, Fr ka, const Fr* b, Fr kb, Fr* out, size_t n) {
    if (!n) return;
    ProfScope ps(ctx, "fr_lincomb_to_canonical", 96.0 * n);
    hipLaunchKernelGGL(k_lincomb_to_canonical, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, a, ka.to_canonical(), b, kb.to_canonical(), out, n);
    ZK_HIP(hipGetLastError());
}

__global__ void k_sub_inplace(Fr* __restrict__ a, const Fr* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] - b[i];
}
void fr_sub_inplace(zk_ctx* ctx, Fr* a, const Fr* b, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_sub_inplace, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, a, b, n);
    ZK_HIP(hipGetLastError());
}

// Long division of r (len_r coefficients, destroyed) by t of degree d with leading coefficient
// inverse cinv: q[k-d] = r[k]*cinv, r[k-d+j] -= q*t[j]  for k = len_r-1 .. d.  One workgroup;
// mirrors field::polynomial_division (field/mod.rs:428-469) for a monic-or-not divisor.
__global__ __launch_bounds__(1024) void k_poly_divide(Fr* __restrict__ r, size_t len_r, const Fr* __restrict__ t, size_t d, const Fr* __restrict__ cinv_p, Fr* __restrict__ q) {
    __shared__ Fr s_sh;
    const Fr cinv = *cinv_p;
    for (size_t k = len_r; k-- > d;) {
        if (threadIdx.x == 0) {
            Fr s = r[k] * cinv;
            q[k - d] = s;
            s_sh = s;
        }
        __syncthreads();
        Fr s = s_sh;
        for (size_t j = threadIdx.x; j <= d; j += blockDim.x) r[k - d + j] = r[k - d + j] - s * t[j];
        __syncthreads();
    }
}
void poly_divide(zk_ctx* ctx, Fr* r, size_t len_r, const Fr* t, size_t d, const Fr* cinv, Fr* q) {
    ProfScope ps(ctx, "qap_poly_divide", 32.0 * (len_r + d));
    hipLaunchKernelGGL(k_poly_divide, dim3(1), dim3(1024), 0, ctx->stream, r, len_r, t, d, cinv, q);
    ZK_HIP(hipGetLastError());
}


// ---- quotient by t through the power-series inverse of rev(t) (dense form at large n) -----------
// polynomial_division (field/mod.rs:428-469) is O(n^2) and strictly sequential: n dependent steps, one workgroup.
// The quotient of P (len_r coefficients) by t (degree d) is also
//     rev_K(q) = rev_K(P) * rev(t)^-1  mod x^K,   K = len_r - d,
// the same unique polynomial, so the same field elements.  rev(t)^-1 mod x^K comes from Newton's iteration
// g <- 2g - rev(t) g^2 (doubling precision, three NTTs per step) once per QAP; a proof then costs two NTTs.
__global__ void k_reverse_prefix(const Fr* __restrict__ src, size_t src_len, size_t count, Fr* __restrict__ dst, size_t dst_len) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dst_len) return;
    dst[i] = i < count ? src[src_len - 1 - i] : Fr::zero();   // dst[i] = coefficient src_len-1-i, zero padded
}
__global__ void k_newton_combine(const Fr* __restrict__ g, size_t have, const Fr* __restrict__ e, size_t want, Fr* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= want) return;
    Fr gi = i < have ? g[i] : Fr::zero();
    out[i] = gi + gi - e[i];
}
static void reverse_prefix(zk_ctx* ctx, const Fr* src, size_t src_len, size_t count, Fr* dst, size_t dst_len) {
    hipLaunchKernelGGL(k_reverse_prefix, dim3(ceil_div(dst_len, 256)), dim3(256), 0, ctx->stream, src, src_len, count, dst, dst_len);
    ZK_HIP(hipGetLastError());
}

// out = the NTT image (DIF order, size 2^log_size) of 1 / rev(t) mod x^K for a polynomial t of degree d (d + 1 coefficients, Montgomery;
// d_cinv: 1 / its leading coefficient)
void poly_rev_inverse_ntt(zk_ctx* ctx, const Fr* t, size_t d, const Fr* d_cinv, size_t K, unsigned log_size, DevBuf<Fr>& out) {
    const size_t size = (size_t)1 << log_size;
    hipStream_t st = ctx->stream;
    // rt = rev(t): rt[i] = t[d - i], i <= d
    size_t bufsz = size;
    while (bufsz < 2 * K) bufsz <<= 1;
    DevBuf<Fr> rt(d + 1), g(bufsz), f(bufsz), e(bufsz), g2(bufsz);
    reverse_prefix(ctx, t, d + 1, d + 1, rt.p, d + 1);
    // g0 = 1 / rt[0] = 1 / leading coefficient of t
    ZK_HIP(hipMemsetAsync(g.p, 0, bufsz * sizeof(Fr), st));
    ZK_HIP(hipMemcpyAsync(g.p, d_cinv, sizeof(Fr), hipMemcpyDeviceToDevice, st));
    for (size_t have = 1; have < K;) {
        const size_t want = std::min(2 * have, K);
        unsigned lg = 1;
        while (((size_t)1 << lg) < 2 * want) ++lg;   // rt_trunc g^2 has want + 2 have - 2 <= 2 want coefficients: no wrap-around
        const size_t sz = (size_t)1 << lg;
        // f = rt mod x^want, g2 = g mod x^have, both zero padded to sz
        ZK_HIP(hipMemsetAsync(f.p, 0, sz * sizeof(Fr), st));
        ZK_HIP(hipMemcpyAsync(f.p, rt.p, std::min(want, d + 1) * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ZK_HIP(hipMemsetAsync(g2.p, 0, sz * sizeof(Fr), st));
        ZK_HIP(hipMemcpyAsync(g2.p, g.p, have * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        ntt_dif(ctx, f.p, lg, false, false);
        ntt_dif(ctx, g2.p, lg, false, false);
        fr_pointwise_mul(ctx, f.p, g2.p, e.p, sz);
        fr_pointwise_mul(ctx, e.p, g2.p, e.p, sz);
        ntt_dit(ctx, e.p, lg, true, true, nullptr);                 // e = rt_trunc * g^2 (natural order)
        hipLaunchKernelGGL(k_newton_combine, dim3(ceil_div(want, 256)), dim3(256), 0, st, g.p, have, e.p, want, f.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipMemcpyAsync(g.p, f.p, want * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        have = want;
    }
    // NTT image of g mod x^K at the size the product uses
    out.alloc(size);
    ZK_HIP(hipMemsetAsync(out.p, 0, size * sizeof(Fr), st));
    ZK_HIP(hipMemcpyAsync(out.p, g.p, K * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    ntt_dif(ctx, out.p, log_size, false, false);
    ZK_HIP(hipStreamSynchronize(st));
}

void qap_ensure_tinv(zk_ctx* ctx, zk_qap& q, size_t K, unsigned log_size) {
    if (q.t_rinv_ntt.p && q.tinv_log == log_size) return;
    poly_rev_inverse_ntt(ctx, q.dt.p, q.t_degree, q.t_cinv.p, K, log_size, q.t_rinv_ntt);
    q.tinv_log = log_size;
}

// quotient (K = len_r - d coefficients, natural order) of P = r[0 .. len_r) by t into out[0 .. K); `work` has 2^log_size elements
void poly_divide_newton(zk_ctx* ctx, const zk_qap& q, const Fr* r, size_t len_r, unsigned log_size, Fr* work, Fr* out) {
    const size_t K = len_r - q.t_degree, size = (size_t)1 << log_size;
    ProfScope ps(ctx, "qap_poly_divide", 32.0 * (len_r + 3 * size));
    reverse_prefix(ctx, r, len_r, K, work, size);                    // rev_K(P), zero padded
    ntt_dif(ctx, work, log_size, false, false);
    fr_pointwise_mul(ctx, work, q.t_rinv_ntt.p, work, size);
    ntt_dit(ctx, work, log_size, true, true, nullptr);               // (rev_K(P) * rev(t)^-1), first K coefficients are rev_K(q)
    reverse_prefix(ctx, work, K, K, out, K);
}

// sum_i a_i P_i for P = u / v / w (groth16/mod.rs:233-253: zip with the weights, Mul<T>, Sum): the first stage of prove on its own.
// Dense form: the n coefficients (k_dense_matvec).  Sparse forms: the n values on the QAP's domain (k_spmv over the rows by gate;
// u and v only -- W is never evaluated by the prover, see prove.hip).  Canonical output on the host.
void qap_weighted_sum(zk_ctx* ctx, const zk_qap& q, const uint64_t* weights, size_t m_in, int which, uint64_t* out) {
    ZK_REQUIRE(weights && out && which >= 0 && which <= 2, ZK_ERR_ARG, "zk_qap_weighted_sum: bad argument");
    ZK_REQUIRE(q.dense || which < 2, ZK_ERR_UNSUPPORTED, "zk_qap_weighted_sum: a sparse QAP holds only u and v by gate");
    const size_t a_len = std::min(m_in, q.m);   // zip truncates
    DevBuf<Fr> a(std::max<size_t>(a_len, 1)), res(q.n);
    DevBuf<int> flag(1);
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    if (a_len) ZK_HIP(hipMemcpyAsync(a.p, weights, a_len * sizeof(Fr), hipMemcpyHostToDevice, st));
    if (q.dense) {
        fr_to_mont(ctx, a.p, a.p, a_len, flag.p);
        dense_matvec(ctx, which == 0 ? q.du.p : which == 1 ? q.dv.p : q.dw.p, a.p, a_len, q.n, res.p);
    } else {
        fr_check_range(ctx, a.p, a_len, flag.p);
        spmv(ctx, which == 0 ? q.u_gate : q.v_gate, a.p, a_len, res.p);   // the witness as given (k_spmv)
    }
    fr_from_mont(ctx, res.p, res.p, q.n);
    int h = 0;
    ZK_HIP(hipMemcpyAsync(&h, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(out, res.p, q.n * sizeof(Fr), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_REQUIRE(!h, ZK_ERR_RANGE, "zk_qap_weighted_sum: weight >= r");
}

}  // namespace zk
