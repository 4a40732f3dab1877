// tools/ubench_field.hip -- throughput of 254-bit Montgomery multiplication variants on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zksnark_rs_amd/csrc tools/ubench_field.hip -o tools/ubench_field
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "ff.cuh"
#include "ec.cuh"
using namespace zk;

// ---- variant B: radix-2^29 product-scanning (Comba) Montgomery, R = 2^261 --------------------
struct Fq29 {
    static constexpr uint32_t P[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV = 0x04866389u;
};
#define M29 0x1fffffffu
__device__ __forceinline__ void to29(const uint32_t* x, uint32_t* o) {
    // o[k] = bits [29k, 29k+29) of the 256-bit x
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int bit = 29 * k, w = bit >> 5, s = bit & 31;
        uint32_t lo = x[w] >> s;
        if (s > 3 && w + 1 < 8) lo |= x[w + 1] << (32 - s);
        o[k] = lo & M29;
    }
}
__device__ __forceinline__ void from29(const uint32_t* a, uint32_t* x) {
    // a[k] < 2^29 (a[8] small); x = sum a[k] 2^(29k) as 8x32
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        int bit = 32 * w, k = bit / 29, s = bit - 29 * k;   // word w starts inside limb k at offset s
        uint32_t v = a[k] >> s;
        int have = 29 - s;
        if (have < 32 && k + 1 < 9) v |= a[k + 1] << have;
        if (have + 29 < 32 && k + 2 < 9) v |= a[k + 2] << (have + 29);
        x[w] = v;
    }
}
template <class PR>
__device__ __forceinline__ void mul29(const uint32_t* a, const uint32_t* b, uint32_t* r) {
    uint32_t m[9];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * PR::P[k - i];
        m[k] = ((uint32_t)acc * PR::INV) & M29;
        acc += (uint64_t)m[k] * PR::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - 8; i <= 8; ++i) acc += (uint64_t)m[i] * PR::P[k - i];
        r[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
}
// full multiply on 8x32 operands via radix-29 (result fully reduced)
__device__ __forceinline__ Fq mulB(const Fq& x, const Fq& y) {
    uint32_t a[9], b[9], r[9];
    to29(x.l, a); to29(y.l, b);
    mul29<Fq29>(a, b, r);
    Fq o; from29(r, o.l);
    return Fq::reduce_once(o, 0);
}

template <int V>
__global__ void k_mul_chain(const Fq* __restrict__ a, const Fq* __restrict__ b, Fq* __restrict__ out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fq x = a[i], y = b[i];
    for (int it = 0; it < iters; ++it) {
        if (V == 0) x = x * y; else x = mulB(x, y);
    }
    out[i] = x;
}

template <class F, int WPS>
__global__ __launch_bounds__(64, WPS) void k_madd_chain(const Aff<F>* __restrict__ pts, Jac<F>* __restrict__ out, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Jac<F> acc = Jac<F>::infinity();
    for (int it = 0; it < iters; ++it) acc = jac_madd(acc, pts[(i * 7 + it) & 1023]);
    out[i] = acc;
}

template <class K, class... A>
static double timeit(K kern, dim3 g, dim3 b, A... args) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, g, b, 0, 0, args...);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int CUS = 256;
    const size_t maxthreads = (size_t)CUS * 4 * 8 * 64;
    std::vector<Fq> ha(maxthreads), hb(maxthreads);
    uint64_t s = 88172645463325252ULL;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (size_t i = 0; i < maxthreads; ++i) for (int k = 0; k < 8; ++k) { ha[i].l[k] = rnd(); hb[i].l[k] = rnd(); if (k == 7) { ha[i].l[k] &= 0x0fffffff; hb[i].l[k] &= 0x0fffffff; } }
    Fq *da, *db, *dout;
    (void)hipMalloc(&da, maxthreads * sizeof(Fq)); (void)hipMalloc(&db, maxthreads * sizeof(Fq)); (void)hipMalloc(&dout, maxthreads * sizeof(Fq));
    (void)hipMemcpy(da, ha.data(), maxthreads * sizeof(Fq), hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), maxthreads * sizeof(Fq), hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        dim3 g(CUS * wps), b(256);
        double t0 = timeit(k_mul_chain<0>, g, b, da, db, dout, iters);
        double t1 = timeit(k_mul_chain<1>, g, b, da, db, dout, iters);
        double n = (double)CUS * wps * 256 * iters;
        printf("waves/SIMD=%d  cios32: %.3f ms %.1f Gmul/s   comba29: %.3f ms %.1f Gmul/s\n", wps, t0, n / t0 * 1e-6, t1, n / t1 * 1e-6);
    }
    // correctness of comba29 vs cios32: a*b*2^-261 * 2^5 == a*b*2^-256  (mod q)
    {
        std::vector<Fq> o0(1024), o1(1024);
        hipLaunchKernelGGL(k_mul_chain<0>, dim3(4), dim3(256), 0, 0, da, db, dout, 1);
        (void)hipMemcpy(o0.data(), dout, 1024 * sizeof(Fq), hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_mul_chain<1>, dim3(4), dim3(256), 0, 0, da, db, dout, 1);
        (void)hipMemcpy(o1.data(), dout, 1024 * sizeof(Fq), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) {
            Fq x = o1[i];
            for (int k = 0; k < 5; ++k) x = x + x;     // * 2^5
            Fq y = o0[i];
            if (!(Fq::reduce_once(y, 0) == x)) ++bad;  // inputs were arbitrary < 2^252 (maybe >= q): compare reduced
        }
        printf("comba29 vs cios32 mismatches: %d / 1024\n", bad);
    }
    // madd chains
    std::vector<Aff<Fq>> hp(1024);
    for (auto& p : hp) for (int k = 0; k < 8; ++k) { p.x.l[k] = rnd() >> (k == 7 ? 4 : 0); p.y.l[k] = rnd() >> (k == 7 ? 4 : 0); }
    Aff<Fq>* dp; Jac<Fq>* dj;
    (void)hipMalloc(&dp, 1024 * sizeof(Aff<Fq2>)); (void)hipMalloc(&dj, maxthreads * sizeof(Jac<Fq2>));
    (void)hipMemcpy(dp, hp.data(), 1024 * sizeof(Aff<Fq>), hipMemcpyHostToDevice);
    const int mit = 200;
    for (int wps : {1, 2, 3, 4}) {
        size_t threads = (size_t)CUS * 4 * wps * 64;
        double n = (double)threads * mit;
        double tg1 = wps == 1 ? timeit(k_madd_chain<Fq, 1>, dim3(threads / 64), dim3(64), dp, dj, mit)
                   : wps == 2 ? timeit(k_madd_chain<Fq, 2>, dim3(threads / 64), dim3(64), dp, dj, mit)
                   : wps == 3 ? timeit(k_madd_chain<Fq, 3>, dim3(threads / 64), dim3(64), dp, dj, mit)
                              : timeit(k_madd_chain<Fq, 4>, dim3(threads / 64), dim3(64), dp, dj, mit);
        double tg2 = wps == 1 ? timeit(k_madd_chain<Fq2, 1>, dim3(threads / 64), dim3(64), (Aff<Fq2>*)dp, (Jac<Fq2>*)dj, mit)
                   : wps == 2 ? timeit(k_madd_chain<Fq2, 2>, dim3(threads / 64), dim3(64), (Aff<Fq2>*)dp, (Jac<Fq2>*)dj, mit)
                   : wps == 3 ? timeit(k_madd_chain<Fq2, 3>, dim3(threads / 64), dim3(64), (Aff<Fq2>*)dp, (Jac<Fq2>*)dj, mit)
                              : timeit(k_madd_chain<Fq2, 4>, dim3(threads / 64), dim3(64), (Aff<Fq2>*)dp, (Jac<Fq2>*)dj, mit);
        printf("madd launch_bounds waves/SIMD=%d: G1 %.3f ms %.2f Gadd/s   G2 %.3f ms %.2f Gadd/s\n", wps, tg1, n / tg1 * 1e-6, tg2, n / tg2 * 1e-6);
    }
    return 0;
}
