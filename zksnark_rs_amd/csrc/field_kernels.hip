// field_kernels.hip -- element-wise field / group kernels behind the diagnostic C-ABI entry
// points (zk_fr_batch, zk_fq_batch, zk_g1_mul_batch, ...) and the canonical<->Montgomery
// conversions used at the ABI boundary.  One element per lane; 32 B per Fr/Fq so each lane
// issues two global_load_dwordx4 per operand and consecutive lanes touch consecutive memory.
#include "common.hpp"
#include "kernels.hpp"
#include "fr_tile.cuh"

namespace zk {

template <class F>
__global__ void k_field_batch(int op, const F* __restrict__ a, const F* __restrict__ b, F* __restrict__ out, size_t n, int* __restrict__ flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = a[i];
    if (!x.raw_in_range()) { atomicOr(flag, 2); return; }
    x = F::from_canonical(x);
    F r;
    if (op >= 3) {
        if (x.is_zero()) { atomicOr(flag, 1); return; }
        r = op == 3 ? x.inv() : op == 4 ? x.inv_euclid() : x.inv_divsteps();
    } else {
        F y = b[i];
        if (!y.raw_in_range()) { atomicOr(flag, 2); return; }
        y = F::from_canonical(y);
        r = op == 0 ? x + y : (op == 1 ? x - y : x * y);
    }
    out[i] = r.to_canonical();
}

template <class F>
void field_batch(zk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    ZK_REQUIRE(a && out && (op >= 3 || b) && op >= 0 && op <= 5, ZK_ERR_ARG, "zk_f*_batch: bad argument");
    if (n == 0) return;
    DevBuf<F> da(n), db(op >= 3 ? 0 : n), dout(n);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemcpyAsync(da.p, a, n * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    if (op < 3) ZK_HIP(hipMemcpyAsync(db.p, b, n * sizeof(F), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_field_batch<F>, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, op, da.p, db.p, dout.p, n, flag.p);
    ZK_HIP(hipGetLastError());
    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipMemcpyAsync(out, dout.p, n * sizeof(F), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!(hflag & 2), ZK_ERR_RANGE, "field element >= modulus");
    ZK_REQUIRE(!(hflag & 1), ZK_ERR_DIV_BY_ZERO, "Tried to divide by zero");
}
template void field_batch<Fr>(zk_ctx*, int, const uint64_t*, const uint64_t*, uint64_t*, size_t);
template void field_batch<Fq>(zk_ctx*, int, const uint64_t*, const uint64_t*, uint64_t*, size_t);

// ---- canonical <-> Montgomery over arrays ------------------------------------------------
__global__ void k_fr_to_mont(const Fr* __restrict__ in, Fr* __restrict__ out, size_t n, int* flag) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x = in[i];
    if (!x.raw_in_range()) { atomicOr(flag, 2); return; }
    out[i] = Fr::from_canonical(x);
}
// *flag |= 2 when an element is not below r (the range check of k_fr_to_mont without the conversion: the sparse prover multiplies its
// witness in canonical form, qap.hip k_spmv)
__global__ void k_fr_check_range(const Fr* __restrict__ in, size_t n, int* flag) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !in[i].raw_in_range()) atomicOr(flag, 2);
}
void fr_check_range(zk_ctx* ctx, const Fr* in, size_t n, int* d_flag) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_check_range, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, n, d_flag);
    ZK_HIP(hipGetLastError());
}
__global__ void k_fr_from_mont(const Fr* __restrict__ in, Fr* __restrict__ out, size_t n) {
    ZK_LATENCY_KERNEL();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {   // one multiplication by the plain integer 1 in the lazy radix (the asm multiplier), closed by the exact reduction
        Fr one = Fr::zero();
        one.l[0] = 1;
        out[i] = fr_store_exact(FrL::load(in[i]) * FrL::load(one));
    }
}
template <class A>
__global__ void k_pts_to_mont(const A* __restrict__ in, A* __restrict__ out, size_t n, int* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    A p = in[i];
    if (!p.x.raw_in_range() || !p.y.raw_in_range()) { atomicOr(flag, 2); return; }
    out[i] = pt_from_canonical(p);
}
template <class A>
__global__ void k_pts_from_mont(const A* __restrict__ in, A* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pt_to_canonical(in[i]);
}

void fr_to_mont(zk_ctx* ctx, const Fr* in, Fr* out, size_t n, int* d_flag) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_to_mont, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, out, n, d_flag);
    ZK_HIP(hipGetLastError());
}
void fr_from_mont(zk_ctx* ctx, const Fr* in, Fr* out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_fr_from_mont, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, out, n);
    ZK_HIP(hipGetLastError());
}
template <class A>
void pts_to_mont(zk_ctx* ctx, const A* in, A* out, size_t n, int* d_flag) {
    if (!n) return;
    hipLaunchKernelGGL(k_pts_to_mont<A>, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, out, n, d_flag);
    ZK_HIP(hipGetLastError());
}
template <class A>
void pts_from_mont(zk_ctx* ctx, const A* in, A* out, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_pts_from_mont<A>, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, in, out, n);
    ZK_HIP(hipGetLastError());
}
// flag |= 4 when a finite point (Montgomery form) does not satisfy y^2 = x^3 + b.  b is passed in Montgomery form:
// 3 for G1, 3 / (9 + i) for the twist (zk_crs_upload / zk_crs_load: a CRS is not trusted to be on the curve --
// an off-curve base would leak witness scalars through the inner products, invalid-curve style).  G1 has prime order; for the G2
// arrays being on the twist is not enough (composite cofactor): crs.hip g2_subgroup_check multiplies them by r as well.
template <class A, class F>
__global__ void k_pts_on_curve(const A* __restrict__ pts, size_t n, F b, int* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    A p = pts[i];
    if (p.is_inf()) return;
    if (!(p.y.sqr() == p.x.sqr() * p.x + b)) atomicOr(flag, 4);
}
static Fq2 twist_b() { return Fq2{Fq::from_u32(3), Fq::zero()} * Fq2{Fq::from_u32(9), Fq::from_u32(1)}.inv(); }
template <>
void pts_check_on_curve<G1A>(zk_ctx* ctx, const G1A* d_pts, size_t n, int* d_flag) {
    if (!n) return;
    hipLaunchKernelGGL((k_pts_on_curve<G1A, Fq>), dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_pts, n, Fq::from_u32(3), d_flag);
    ZK_HIP(hipGetLastError());
}
template <>
void pts_check_on_curve<G2A>(zk_ctx* ctx, const G2A* d_pts, size_t n, int* d_flag) {
    if (!n) return;
    hipLaunchKernelGGL((k_pts_on_curve<G2A, Fq2>), dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_pts, n, twist_b(), d_flag);
    ZK_HIP(hipGetLastError());
}
template void pts_to_mont<G1A>(zk_ctx*, const G1A*, G1A*, size_t, int*);
template void pts_to_mont<G2A>(zk_ctx*, const G2A*, G2A*, size_t, int*);
template void pts_from_mont<G1A>(zk_ctx*, const G1A*, G1A*, size_t);
template void pts_from_mont<G2A>(zk_ctx*, const G2A*, G2A*, size_t);

// ---- out[i] = k[i] * P[i] and out[i] = a[i] + b[i]  (diagnostics; one point per lane) -----
template <class F>
__global__ void k_point_mul(const Aff<F>* __restrict__ pts, const Fr* __restrict__ sc, Aff<F>* __restrict__ out, size_t n, int* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<F> p = pts[i];
    Fr k = sc[i];
    if (!p.x.raw_in_range() || !p.y.raw_in_range() || !k.raw_in_range()) { atomicOr(flag, 2); return; }
    Jac<F> r = jac_mul_words(Jac<F>::from_affine(pt_from_canonical(p)), k.l);
    out[i] = pt_to_canonical(jac_to_affine(r));
}
template <class F>
__global__ void k_point_add(const Aff<F>* __restrict__ a, const Aff<F>* __restrict__ b, Aff<F>* __restrict__ out, size_t n, int* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Aff<F> p = a[i], q = b[i];
    if (!p.x.raw_in_range() || !p.y.raw_in_range() || !q.x.raw_in_range() || !q.y.raw_in_range()) { atomicOr(flag, 2); return; }
    // exercise both the general and the mixed formula; they must agree as group elements
    Jac<F> r = jac_madd_ni(Jac<F>::from_affine(pt_from_canonical(p)), pt_from_canonical(q));
    out[i] = pt_to_canonical(jac_to_affine(r));
}

template <class F>
void point_mul_batch(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t* out, size_t n) {
    ZK_REQUIRE(points && scalars && out, ZK_ERR_ARG, "zk_g*_mul_batch: null pointer");
    if (!n) return;
    DevBuf<Aff<F>> dp(n), dout(n);
    DevBuf<Fr> ds(n);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemcpyAsync(dp.p, points, n * sizeof(Aff<F>), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(ds.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_point_mul<F>, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, dp.p, ds.p, dout.p, n, flag.p);
    ZK_HIP(hipGetLastError());
    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipMemcpyAsync(out, dout.p, n * sizeof(Aff<F>), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "coordinate or scalar >= modulus");
}
// canonical <-> Montgomery coordinates of whole arrays (the affine pair sums work on device-form points)
template <class F>
__global__ void k_points_from_canonical(Aff<F>* __restrict__ p, size_t n, int* flag) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Aff<F> q = p[i];
    if (!q.x.raw_in_range() || !q.y.raw_in_range()) { atomicOr(flag, 2); return; }
    p[i] = pt_from_canonical(q);
}
template <class F>
__global__ void k_points_to_canonical(Aff<F>* __restrict__ p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = pt_to_canonical(p[i]);
}
template <class F>
void point_add_batch(zk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    ZK_REQUIRE(a && b && out, ZK_ERR_ARG, "zk_g*_add_batch: null pointer");
    if (!n) return;
    DevBuf<Aff<F>> da(n), db(n), dout(n);
    DevBuf<int> flag(1);
    ZK_HIP(hipMemcpyAsync(da.p, a, n * sizeof(Aff<F>), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemcpyAsync(db.p, b, n * sizeof(Aff<F>), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_point_add<F>, dim3(ceil_div(n, 64)), dim3(64), 0, ctx->stream, da.p, db.p, dout.p, n, flag.p);
    ZK_HIP(hipGetLastError());
    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipMemcpyAsync(out, dout.p, n * sizeof(Aff<F>), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "coordinate >= modulus");
}
template void point_mul_batch<Fq>(zk_ctx*, const uint64_t*, const uint64_t*, uint64_t*, size_t);
template void point_mul_batch<Fq2>(zk_ctx*, const uint64_t*, const uint64_t*, uint64_t*, size_t);
template void point_add_batch<Fq>(zk_ctx*, const uint64_t*, const uint64_t*, uint64_t*, size_t);
template void point_add_batch<Fq2>(zk_ctx*, const uint64_t*, const uint64_t*, uint64_t*, size_t);

}  // namespace zk
