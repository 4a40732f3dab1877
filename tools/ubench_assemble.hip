// ubench_assemble.hip -- what the closing kernel of a proof (k_assemble) spends its time on, one lane per wave as there:
// a Jacobian addition, the blinded variable-time inversion, the affine conversion, the byte encoding.  Times from s_memrealtime
// (100 MHz) around each part, medians over repeats.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I zksnark_rs_amd/csrc tools/ubench_assemble.hip -o tools/_bin/ubench_assemble
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include "lazy29.cuh"
using namespace zk;

__device__ Fq inv_second(const Fq& x) { return x.inv_euclid(); }
__device__ Fq2 inv_second(const Fq2& x) { Fq d = (x.c0.sqr() + x.c1.sqr()).inv_euclid(); return Fq2{x.c0 * d, -(x.c1 * d)}; }

template <class F>
__global__ void k_parts(Aff<F> g, F lambda, unsigned long long* t, Aff<F>* out) {
    if (threadIdx.x & 63) return;
    Jac<F> p = Jac<F>::from_affine(g);
    for (int i = 0; i < 5; ++i) p = jac_dbl_ni(jac_add_ni(p, Jac<F>::from_affine(g)));
    Jac<F> q = jac_dbl_ni(p);
    unsigned long long t0 = wall_clock64();
    Jac<F> s = jac_add_ni(p, q);
    unsigned long long t1 = wall_clock64();
    F zl = s.Z * lambda;
    unsigned long long t2 = wall_clock64();
    F zi = zl.inv_vartime();
    unsigned long long t3 = wall_clock64();
    F zi_e = inv_second(zl);
    unsigned long long t3b = wall_clock64();
    zi = zi * lambda;
    F zi2 = zi.sqr();
    Aff<F> a{s.X * zi2, s.Y * zi2 * zi};
    unsigned long long t4 = wall_clock64() - (t3b - t3);
    // lazy-form addition of the same points, for comparison
    JacR<F> lp = jacr_load(p), lq = jacr_load(q);
    unsigned long long t5 = wall_clock64();
    Jac<F> s2 = jacr_store(add_lazy(lp, lq));
    unsigned long long t6 = wall_clock64();
    t[0] = t1 - t0; t[1] = t3 - t2; t[2] = (t2 - t1) + (t4 - t3); t[3] = t6 - t5; t[4] = t3b - t3;
    if (!(zi_e == zl.inv_vartime())) t[4] = 0;
    out[0] = a;
    out[1] = Aff<F>{s2.X, s2.Y};
}

template <class F>
void run(const char* name, Aff<F> g, F lambda) {
    unsigned long long* t; Aff<F>* out;
    hipMalloc(&t, 64); hipMalloc(&out, 2 * sizeof(Aff<F>));
    unsigned long long h[5], acc[5][9];
    for (int r = 0; r < 9; ++r) {
        hipLaunchKernelGGL(k_parts<F>, dim3(1), dim3(64), 0, 0, g, lambda, t, out);
        hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
        for (int k = 0; k < 5; ++k) acc[k][r] = h[k];
    }
    for (int k = 0; k < 5; ++k) std::sort(acc[k], acc[k] + 9);
    printf("%s (us): jac_add (8x32 form) %.1f | inversion of Z lambda %.1f (binary Euclid: %.1f) | rest of the affine conversion %.1f | the same addition in the lazy form incl. store %.1f\n", name,
           acc[0][4] / 100.0, acc[1][4] / 100.0, acc[4][4] / 100.0, acc[2][4] / 100.0, acc[3][4] / 100.0);
}

int main() {
    Fq one = Fq::zero(), two = Fq::zero(), lam = Fq::zero();
    one.l[0] = 1; two.l[0] = 2; lam.l[0] = 0x9e3779b9; lam.l[3] = 0x7f4a7c15; lam.l[6] = 12345;
    Aff<Fq> g1{Fq::from_canonical(one), Fq::from_canonical(two)};
    run<Fq>("G1", g1, Fq::from_canonical(lam));
    const uint32_t x0[8] = {0xd992f6ed, 0x46debd5c, 0xf75edadd, 0x674322d4, 0x5e5c4479, 0x426a0066, 0x121f1e76, 0x1800deef};
    const uint32_t x1[8] = {0xaef312c2, 0x97e485b7, 0x35a9e712, 0xf1aa4933, 0x31fb5d25, 0x7260bfb7, 0x920d483a, 0x198e9393};
    const uint32_t y0[8] = {0x66fa7daa, 0x4ce6cc01, 0x0c43d37b, 0xe3d1e769, 0x8dcb408f, 0x4aab7180, 0xdb8c6deb, 0x12c85ea5};
    const uint32_t y1[8] = {0xd122975b, 0x55acdadc, 0x70b38ef3, 0xbc4b3133, 0x690c3395, 0xec9e99ad, 0x585ff075, 0x090689d0};
    Fq a, b, c, d;
    for (int i = 0; i < 8; ++i) { a.l[i] = x0[i]; b.l[i] = x1[i]; c.l[i] = y0[i]; d.l[i] = y1[i]; }
    Aff<Fq2> g2{Fq2{Fq::from_canonical(a), Fq::from_canonical(b)}, Fq2{Fq::from_canonical(c), Fq::from_canonical(d)}};
    run<Fq2>("G2", g2, Fq2{Fq::from_canonical(lam), Fq::from_canonical(two)});
    return 0;
}
