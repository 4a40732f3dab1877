// msm_quad.hpp -- the reduction tail of a SMALL MSM with four lanes per point addition (quad29.cuh).  Included by msm_impl.hpp.
//
// Same stages and the same data as k_msm_merge / k_msm_fold / k_msm_weigh / k_msm_sum_points; job j of a kernel is executed by lanes
// 4j .. 4j+3, every lane of the quad follows the job's control flow and holds the complete result, lane 0 of the quad stores.  The
// terms of the weighted sum stay accumulator images (XYZZ) up to the last addition.
// (inside namespace zk; quad29.cuh is included at the top of msm_impl.hpp)
constexpr int QUAD_THREADS = 64;                  // one wave = 16 jobs: fits wherever an accumulation wave retires (see TAIL_THREADS)
constexpr int QUAD_JOBS = QUAD_THREADS / 4;
// Registers: the Fq2 kernels are left all 256 + AGPRs (one wave per SIMD).  They serve products of few buckets, i.e. small circuits, where no
// chip-filling accumulation competes for the register files (that argument, TailWaves, is for 2^20 gates: there the G2 product has
// 2^19 buckets and takes the one-lane tail); capped at 168 they carried ~450 B of spills through every addition.
#ifndef ZK_QUAD_G2_WAVES
#define ZK_QUAD_G2_WAVES 1
#endif
template <class F> struct QuadWaves { static constexpr int value = TailWaves<F>::value; };
template <> struct QuadWaves<Fq2> { static constexpr int value = ZK_QUAD_G2_WAVES; };
// BIG: the same kernels at the end of the tail of a LARGE product (msm_run: quad_end), where accumulations of other products do fill
// the chip: the register budget of the one-lane tail kernels (TailWaves), so that a wave fits wherever one accumulation wave retires.
template <class F, bool BIG> struct QuadWavesOf { static constexpr int value = BIG ? TailWaves<F>::value : QuadWaves<F>::value; };

template <class F>
__global__ __launch_bounds__(QUAD_THREADS, QuadWaves<F>::value) void k_msm_merge_q(const uint32_t* __restrict__ start, int buckets, uint32_t T, const uint32_t* __restrict__ xbase,
                                                     AccSlot<F>* __restrict__ img, uint32_t* __restrict__ heavy, uint32_t heavy_cap) {
    ZK_LATENCY_KERNEL();
    const uint32_t b = blockIdx.x * QUAD_JOBS + (threadIdx.x >> 2);
    const int role = threadIdx.x & 3;
    if (b >= (uint32_t)buckets) return;
    const uint32_t z = start[b + 1] - start[b];
    typename AccOf<F>::type acc;
    if (!z) { acc_clear(acc); if (role == 0) img[b].a = acc; return; }
    const uint32_t r = (z + T - 1) / T;
    if (r == 1) return;
    if (r - 1 > MSM_HEAVY) {   // as k_msm_merge
        if (role == 0) {
            const uint32_t chunks = (r + MSM_HEAVY_CHUNK - 1) / MSM_HEAVY_CHUNK, at = atomicAdd(&heavy[0], chunks);
            for (uint32_t w = 0; w < chunks; ++w) { heavy[2 + 2 * (at + w)] = b; heavy[3 + 2 * (at + w)] = w; }
            if (chunks > 1) heavy[heavy_cap - 1 - atomicAdd(&heavy[1], 1u)] = b;
        }
        return;
    }
    const AccSlot<F>* more = img + (size_t)buckets + xbase[b];
    acc = img[b].a;
    for (uint32_t j = 0; j + 1 < r; ++j) acc = quad_add_xyzz(acc, more[j].a, role);
    if (role == 0) img[b].a = acc;
}

template <class F, bool BIG>
__global__ __launch_bounds__(QUAD_THREADS, (QuadWavesOf<F, BIG>::value)) void k_msm_fold_q(FoldJob j0, FoldJob j1) {
    ZK_LATENCY_KERNEL();
    const bool second = blockIdx.x >= j0.blocks;
    const AccSlot<F>* in = reinterpret_cast<const AccSlot<F>*>(second ? j1.in : j0.in);
    AccSlot<F>* out = reinterpret_cast<AccSlot<F>*>(second ? j1.out : j0.out);
    const uint32_t A = second ? j1.A : j0.A, f = second ? j1.f : j0.f, B = second ? j1.B : j0.B;
    const uint32_t j = (second ? blockIdx.x - j0.blocks : blockIdx.x) * QUAD_JOBS + (threadIdx.x >> 2);
    const int role = threadIdx.x & 3;
    if (j >= A * B) return;
    const uint32_t a = j / B, b = j - a * B;
    const AccSlot<F>* src = in + (size_t)a * f * B + b;
    typename AccOf<F>::type acc = src[0].a;
    for (uint32_t i = 1; i < f; ++i) acc = quad_add_xyzz(acc, src[(size_t)i * B].a, role);
    if (role == 0) out[j].a = acc;
}

// term[group][j] as in k_msm_weigh, kept as accumulator images
template <class F, bool BIG>
__global__ __launch_bounds__(QUAD_THREADS, (QuadWavesOf<F, BIG>::value)) void k_msm_weigh_q(const AccSlot<F>* __restrict__ C, const AccSlot<F>* __restrict__ R, int kbits, int rows,
                                                     uint32_t wbase, AccSlot<F>* __restrict__ term) {
    ZK_LATENCY_KERNEL();
    const int K = 1 << kbits, g = blockIdx.y;
    const int j = blockIdx.x * QUAD_JOBS + (threadIdx.x >> 2), role = threadIdx.x & 3;
    if (j >= K + rows) return;
    const AccSlot<F>* src = j < K ? C + (size_t)g * K + j : R + (size_t)g * rows + (j - K);
    const uint32_t w = j < K ? (uint32_t)j : ((uint32_t)(j - K) << kbits) + 1u + wbase;   // wbase: first bucket of a bucket-range shard
    const typename AccOf<F>::type t = quad_mul_small_xyzz(src->a, w, role);
    if (role == 0) term[(size_t)g * (K + rows) + j].a = t;
}

// Sums `count` accumulator images per group (blockIdx.y): workgroup x takes the images [256 x, 256 x + 256) -- each of its 64 quads its
// share, then a tree over LDS.  FINAL: one workgroup per group, the sum goes out as a Jacobian point at (bytes) out + y out_stride;
// otherwise the partial sums go to part[y gridDim.x + x] (a second launch with FINAL adds them).
constexpr int QUAD_SUM_THREADS = 256;
template <class F, bool FINAL, bool BIG>
__global__ __launch_bounds__(QUAD_SUM_THREADS, (QuadWavesOf<F, BIG>::value)) void k_msm_sum_q(const AccSlot<F>* __restrict__ in, int count, AccSlot<F>* __restrict__ part,
                                                            Jac<F>* __restrict__ out, size_t out_stride) {
    ZK_LATENCY_KERNEL();
    __shared__ AccSlot<F> sh[QUAD_SUM_THREADS / 4];
    constexpr int Q = QUAD_SUM_THREADS / 4;
    in += (size_t)blockIdx.y * count;
    const int q = threadIdx.x >> 2, role = threadIdx.x & 3;
    const int lo = FINAL ? 0 : (int)blockIdx.x * QUAD_SUM_THREADS, hi = FINAL ? count : min(lo + QUAD_SUM_THREADS, count);
    typename AccOf<F>::type acc;
    acc_clear(acc);
    for (int k = lo + q; k < hi; k += Q) acc = quad_add_xyzz(acc, in[k].a, role);
    if (role == 0) sh[q].a = acc;
    __syncthreads();
    int d0 = Q / 2;
    while (d0 >= 2 && d0 >= hi - lo) d0 >>= 1;   // quads at and behind the number of images hold infinity
    for (int d = d0; d >= 1; d >>= 1) {
        if (q < d) {
            acc = quad_add_xyzz(sh[q].a, sh[q + d].a, role);
            if (role == 0) sh[q].a = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (FINAL) *reinterpret_cast<Jac<F>*>(reinterpret_cast<uint8_t*>(out) + (size_t)blockIdx.y * out_stride) = acc_store(sh[0].a);
        else part[(size_t)blockIdx.y * gridDim.x + blockIdx.x].a = sh[0].a;
    }
}
