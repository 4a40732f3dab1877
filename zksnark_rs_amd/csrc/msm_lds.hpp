// msm_lds.hpp -- the Pippenger form BASELINE.json's north_star words: "one wavefront per window, LDS bucket accumulation and a
// warp-reduce over partial sums", kept as a MEASURED comparator of the shipped form (msm_impl.hpp), never on the prove path.
//
// Replaces the same inner products (/root/reference/src/groth16/mod.rs:255-272, 279-290; fr.rs:114-119).  One wave owns one
// (window, scalar chunk): its 2^(c-1) buckets are XYZZ accumulators in LDS (c <= 9: 256 x 148 B = 37 KB, four waves per compute
// unit), lane l takes scalar l of every 64, forms the window's signed digit and adds T[w][i] into the digit's bucket under a
// per-bucket LDS lock (lanes of a wave that hit the same bucket take turns).  At the end the wave folds sum_b b S_b: every lane runs
// the running-sum over its own 2^(c-1) / 64 buckets, the 64 partial results are summed by a tree over LDS.  The window tables
// T[w][i] = 2^(c w) P_i are the shipped ones, so the (window, chunk) results simply add up (no Horner tail).
// Why it loses (profiles/r3_window_sweep_2p20.jsonl): c <= 9 means >= 29 windows where the shipped form has 13 -- 2.2 x
// the additions --, a wave per SIMD instead of three, and 64 lanes over 256 buckets collide (the slowest bucket of a round of 64
// additions takes ~3 turns).
// (Included by msm_impl.hpp inside namespace zk, behind k_msm_sum_points, in the translation unit that holds the shared kernels.)
#pragma once

constexpr int LDS_MSM_MAX_C = 10;   // 2^9 XYZZ buckets + locks = 76 KB of the 160 KB per compute unit: two waves per unit (c = 9: four)

__global__ __launch_bounds__(64) void k_msm_lds_g1(const Aff<Fq>* __restrict__ table, size_t n, const Fr* __restrict__ scalars, size_t n_used, int c, int windows,
                                                   size_t chunk_len, Jac<Fq>* __restrict__ out) {
    typedef FpR<FqParams> L;
    typedef XyzzR<L> Acc;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int buckets = 1 << (c - 1);
    Acc* bucket = reinterpret_cast<Acc*>(smem);
    uint32_t* lock = reinterpret_cast<uint32_t*>(smem + (size_t)buckets * sizeof(Acc));
    const int lane = threadIdx.x, w = blockIdx.y;
    for (int b = lane; b < buckets; b += 64) { acc_clear(bucket[b]); lock[b] = 0; }
    __syncthreads();
    const size_t lo = (size_t)blockIdx.x * chunk_len, hi = min(lo + chunk_len, n_used);
    const Aff<Fq>* tw = table + (size_t)w * n;
    for (size_t base = lo; base < hi; base += 64) {
        const size_t i = base + lane;
        uint32_t mag = 0, neg = 0;
        if (i < hi) for_each_digit(scalars[i], c, windows, w, windows, [&](int, uint32_t m_, uint32_t n_) { mag = m_; neg = n_; });
        Aff<Fq> p = mag ? tw[i] : Aff<Fq>::infinity();
        bool pending = mag != 0 && !p.is_inf();
        L qx = L::load(p.x), qy = L::load(p.y);
        if (neg) qy = qy.neg();
        // LDS bucket accumulation: the lane that gets the bucket's lock adds, the others of the same bucket take the next turn
        while (__any(pending)) {
            if (pending && atomicCAS(&lock[mag - 1], 0u, 1u) == 0u) {
                Acc a = bucket[mag - 1];
                if (!acc_madd(a, qx, qy)) acc_load(a, jac_dbl(acc_store(a)));
                bucket[mag - 1] = a;
                __threadfence_block();
                atomicExch(&lock[mag - 1], 0u);
                pending = false;
            }
        }
    }
    __syncthreads();
    // sum_b b S_b: lane l owns buckets [l per, (l + 1) per) (ids b - 1); running sum from its top bucket down gives
    // sum (b - l per) S_b and T = sum S_b; the offset l per T is a small multiple
    const int per = (buckets + 63) / 64;
    JacR<Fq> run = jacr_load(Jac<Fq>::infinity()), acc = run;
    for (int k = per - 1; k >= 0; --k) {
        const int b = lane * per + k;
        if (b < buckets) run = add_lazy(run, jacr_load(acc_store(bucket[b])));
        acc = add_lazy(acc, run);
    }
    acc = add_lazy(acc, mul_small_lazy(run, (uint32_t)(lane * per)));
    __syncthreads();
    Jac<Fq>* sh = reinterpret_cast<Jac<Fq>*>(smem);     // the buckets are dead: reuse their space for the tree
    sh[lane] = jacr_store(acc);
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) {
        if (lane < d) sh[lane] = jacr_store(add_lazy(jacr_load(sh[lane]), jacr_load(sh[lane + d])));
        __syncthreads();
    }
    if (lane == 0) out[(size_t)w * gridDim.x + blockIdx.x] = sh[0];
}

// sum_i scalars[i] P_i by the kernel above; table = T[w][i] with c <= 9 bits (msm_build_table); result in d_out (Jacobian)
inline void msm_lds_run_g1(zk_ctx* ctx, hipStream_t st, const MsmTable<Fq>& tab, const Fr* d_scalars, size_t n_used, DevBuf<Jac<Fq>>& parts, Jac<Fq>* d_out) {
    ZK_REQUIRE(tab.c >= 2 && tab.c <= LDS_MSM_MAX_C, ZK_ERR_ARG, "msm (LDS buckets): window_bits must be in [2, 10]");
    const int buckets = 1 << (tab.c - 1);
    // one wave per (window, chunk): ~4 waves per compute unit fit (LDS), one round of them fills the chip
    int chunks = std::max(1, std::min<int>((int)((n_used + 63) / 64), std::max(1, 4 * ctx->cu_count / tab.windows)));
    const size_t chunk_len = ((n_used + chunks - 1) / chunks + 63) / 64 * 64;
    chunks = (int)((n_used + chunk_len - 1) / chunk_len);
    parts.ensure((size_t)chunks * tab.windows);
    const size_t lds = (size_t)buckets * (sizeof(XyzzR<FpR<FqParams>>) + 4) + 64 * sizeof(Jac<Fq>);
    if (lds > 65536) {   // c = 10: beyond the default dynamic LDS limit
        static PerDeviceOnce once;   // per device, not per process
        once.run(ctx->device, [] { ZK_HIP(hipFuncSetAttribute((const void*)k_msm_lds_g1, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); });
    }
    {
        ProfScope ps(ctx, "msm_lds_buckets_g1", 96.0 * n_used, st);
        hipLaunchKernelGGL(k_msm_lds_g1, dim3(chunks, tab.windows), dim3(64), lds, st, tab.table.p, tab.n, d_scalars, n_used, tab.c, tab.windows, chunk_len, parts.p);
    }
    hipLaunchKernelGGL(k_msm_sum_points<Fq>, dim3(1, 1), dim3(256), 256 * sizeof(Jac<Fq>), st, parts.p, chunks * tab.windows, d_out, (size_t)0);
    ZK_HIP(hipGetLastError());
}
