// quad_check.hip -- quad29.cuh against lazy29.cuh on the GPU: the four-lane addition / doubling / small multiple of random points
// must equal the one-lane forms (compared as Jacobian images through xyzz_store, which is canonical per representation, so both are
// brought to affine x Z-independent form: X/ZZ, Y/ZZZ cross-multiplied).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I zksnark_rs_amd/csrc tools/quad_check.hip -o /tmp/quad_check && /tmp/quad_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "quad29.cuh"
using namespace zk;

template <class L>
__device__ bool same_point(const XyzzR<L>& a, const XyzzR<L>& b) {
    if (a.inf || b.inf) return a.inf == b.inf;
    // x_a = X_a / ZZ_a: X_a ZZ_b == X_b ZZ_a, Y_a ZZZ_b == Y_b ZZZ_a
    L l = a.X * b.ZZ - b.X * a.ZZ, r = a.Y * b.ZZZ - b.Y * a.ZZZ;
    return l.norm().sqr().is_zero_mod_p() && r.norm().sqr().is_zero_mod_p();
}

template <class L> __device__ bool same_limbs(const L& a, const L& b);
template <class PR> __device__ bool same_limbs(const FpR<PR>& a, const FpR<PR>& b) { int d = 0; for (int i = 0; i < 9; ++i) d |= a.v[i] ^ b.v[i]; return d == 0; }
template <class PR> __device__ bool same_limbs(const Fp2R<PR>& a, const Fp2R<PR>& b) { return same_limbs(a.c0, b.c0) && same_limbs(a.c1, b.c1); }

template <class F>
__global__ void k_check(const Aff<F>* pts, int n, int* bad, int* dpp, int* per_lane) {
    typedef typename LazyOf<F>::type L;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, job = t >> 2, role = t & 3;
    if (blockIdx.x == 0) {
        dpp[threadIdx.x * 4 + 0] = quad_get<0>((int32_t)threadIdx.x); dpp[threadIdx.x * 4 + 1] = quad_get<1>((int32_t)threadIdx.x);
        dpp[threadIdx.x * 4 + 2] = quad_get<2>((int32_t)threadIdx.x); dpp[threadIdx.x * 4 + 3] = quad_get<3>((int32_t)threadIdx.x);
    }
    if (job + 3 >= n) return;
    XyzzR<L> a, b;
    acc_clear(a); acc_clear(b);
    madd_xyzz(a, L::load(pts[job].x), L::load(pts[job].y));
    madd_xyzz(a, L::load(pts[job + 1].x), L::load(pts[job + 1].y));     // a = P0 + P1 (non-trivial ZZ)
    madd_xyzz(b, L::load(pts[job + 2].x), L::load(pts[job + 2].y));
    madd_xyzz(b, L::load(pts[job + 3].x), L::load(pts[job + 3].y));
    int fail = 0;
    {   // addition
        XyzzR<L> want = add_xyzz(a, b), got = quad_add_xyzz(a, b, role);
        if (!same_point(want, got)) fail |= 1;
        // which coordinate differs from the (correct) one lane 1 holds
        if (!same_limbs(got.X, quad_get<1>(got.X))) fail |= 256;
        if (!same_limbs(got.Y, quad_get<1>(got.Y))) fail |= 512;
        if (!same_limbs(got.ZZ, quad_get<1>(got.ZZ))) fail |= 1024;
        if (!same_limbs(got.ZZZ, quad_get<1>(got.ZZZ))) fail |= 2048;
    }
    {   // doubling: quad_dbl against add of the same point through the Jacobian path of add_xyzz
        XyzzR<L> want = add_xyzz(a, a), got = quad_dbl_xyzz(a, role), got2 = quad_add_xyzz(a, a, role);
        if (!same_point(want, got)) fail |= 2;
        if (!same_point(want, got2)) fail |= 4;
    }
    {   // P + (-P), infinity operands
        XyzzR<L> na = a; na.Y = a.Y.neg().norm();
        if (!quad_add_xyzz(a, na, role).inf) fail |= 8;
        XyzzR<L> z; acc_clear(z);
        if (!same_point(quad_add_xyzz(z, a, role), a) || !same_point(quad_add_xyzz(a, z, role), a)) fail |= 16;
    }
    {   // small multiples
        const uint32_t k = (uint32_t)(job * 2654435761u) & 0x7fff;
        XyzzR<L> want; acc_clear(want);
        XyzzR<L> run = a;
        for (uint32_t e = k; e; e >>= 1) { if (e & 1) want = add_xyzz(want, run); run = add_xyzz(run, run); }
        if (!same_point(want, quad_mul_small_xyzz(a, k, role))) fail |= 32;
    }
    per_lane[t] = fail;
    if (fail) atomicOr(bad, fail);
}

// round by round: what the lanes exchange against the same products computed by every lane for itself
template <class F>
__global__ void k_diag(const Aff<F>* pts, int* out) {
    typedef typename LazyOf<F>::type L;
    const int role = threadIdx.x & 3, job = threadIdx.x >> 2;
    XyzzR<L> p, q;
    acc_clear(p); acc_clear(q);
    madd_xyzz(p, L::load(pts[job].x), L::load(pts[job].y)); madd_xyzz(p, L::load(pts[job + 1].x), L::load(pts[job + 1].y));
    madd_xyzz(q, L::load(pts[job + 2].x), L::load(pts[job + 2].y)); madd_xyzz(q, L::load(pts[job + 3].x), L::load(pts[job + 3].y));
    int bad = 0;
    const L sa = quad_sel(role, p.X, q.X, p.Y, q.Y), sb = quad_sel(role, q.ZZ, p.ZZ, q.ZZZ, p.ZZZ);
    const L wa = role == 0 ? p.X : role == 1 ? q.X : role == 2 ? p.Y : q.Y;
    if (!same_limbs(sa, wa)) bad |= 1;                               // the selection
    const L t1 = sa * sb;
    const L E0 = p.X * q.ZZ, E1 = q.X * p.ZZ, E2 = p.Y * q.ZZZ, E3 = q.Y * p.ZZZ;
    if (!same_limbs(t1, role == 0 ? E0 : role == 1 ? E1 : role == 2 ? E2 : E3)) bad |= 2;   // this lane's product
    if (!same_limbs(quad_get<0>(t1), E0)) bad |= 4;
    if (!same_limbs(quad_get<1>(t1), E1)) bad |= 8;
    if (!same_limbs(quad_get<2>(t1), E2)) bad |= 16;
    if (!same_limbs(quad_get<3>(t1), E3)) bad |= 32;
    const L P = quad_get<1>(t1) - quad_get<0>(t1);
    if (!same_limbs(P, E1 - E0)) bad |= 64;                          // the subtraction the compiler folds into a DPP instruction
    out[threadIdx.x] = bad;
}

// a few hundred points k G by repeated addition (host side: affine through the library's own conversion on the device)
template <class F>
__global__ void k_points(Aff<F> g, int n, Aff<F>* out) {
    Jac<F> acc = Jac<F>::from_affine(g);
    for (int i = 0; i < n; ++i) { out[i] = jac_to_affine(acc); acc = jac_add(acc, Jac<F>::from_affine(g)); acc = jac_dbl(acc); }
}

template <class F>
int run(const char* name, Aff<F> g) {
    const int n = 1024;
    Aff<F>* pts; int *bad, *dpp;
    hipMalloc(&pts, n * sizeof(Aff<F>)); hipMalloc(&bad, 4); hipMalloc(&dpp, 64 * 4 * 4);
    int* per_lane; hipMalloc(&per_lane, n * 4 * 4); hipMemset(per_lane, 0, n * 4 * 4);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k_points<F>, dim3(1), dim3(1), 0, 0, g, n, pts);
    hipLaunchKernelGGL(k_check<F>, dim3(n * 4 / 64), dim3(64), 0, 0, pts, n, bad, dpp, per_lane);
    int* diag; hipMalloc(&diag, 64 * 4);
    hipLaunchKernelGGL(k_diag<F>, dim3(1), dim3(64), 0, 0, pts, diag);
    int hdiag[64];
    hipMemcpy(hdiag, diag, sizeof hdiag, hipMemcpyDeviceToHost);
    printf("%s diag per lane:", name);
    for (int l = 0; l < 16; ++l) printf(" %d", hdiag[l]);
    printf("\n");
    int hbad = -1, hdpp[256];
    hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hdpp, dpp, sizeof hdpp, hipMemcpyDeviceToHost);
    hipError_t e = hipDeviceSynchronize();
    static int hl[4096];
    hipMemcpy(hl, per_lane, n * 4 * 4, hipMemcpyDeviceToHost);
    int nz = 0;
    for (int i = 0; i < n * 4; ++i) nz += hl[i] != 0;
    printf("%s lanes failing: %d of %d; first 32:", name, nz, n * 4);
    for (int i = 0; i < 32; ++i) printf(" %d", hl[i]);
    printf("\n");
    int dpp_bad = 0;
    for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) dpp_bad |= hdpp[l * 4 + k] != (l & ~3) + k;
    printf("%s: failure mask %d (0 = all equal), quad_get %s, hip %s\n", name, hbad, dpp_bad ? "WRONG" : "ok", hipGetErrorString(e));
    return hbad | dpp_bad;
}

int main() {
    Aff<Fq> g1;
    Fq one = Fq::zero(), two = Fq::zero();
    one.l[0] = 1; two.l[0] = 2;
    g1.x = Fq::from_canonical(one); g1.y = Fq::from_canonical(two);
    int rc = run<Fq>("G1", g1);
    Aff<Fq2> g2;   // the generator of G2 (EIP-197), canonical limbs little-endian
    const uint32_t x0[8] = {0xd992f6ed, 0x46debd5c, 0xf75edadd, 0x674322d4, 0x5e5c4479, 0x426a0066, 0x121f1e76, 0x1800deef};
    const uint32_t x1[8] = {0xaef312c2, 0x97e485b7, 0x35a9e712, 0xf1aa4933, 0x31fb5d25, 0x7260bfb7, 0x920d483a, 0x198e9393};
    const uint32_t y0[8] = {0x66fa7daa, 0x4ce6cc01, 0x0c43d37b, 0xe3d1e769, 0x8dcb408f, 0x4aab7180, 0xdb8c6deb, 0x12c85ea5};
    const uint32_t y1[8] = {0xd122975b, 0x55acdadc, 0x70b38ef3, 0xbc4b3133, 0x690c3395, 0xec9e99ad, 0x585ff075, 0x090689d0};
    Fq a, b, c, d;
    for (int i = 0; i < 8; ++i) { a.l[i] = x0[i]; b.l[i] = x1[i]; c.l[i] = y0[i]; d.l[i] = y1[i]; }
    g2.x = Fq2{Fq::from_canonical(a), Fq::from_canonical(b)}; g2.y = Fq2{Fq::from_canonical(c), Fq::from_canonical(d)};
    rc |= run<Fq2>("G2", g2);
    return rc != 0;
}
