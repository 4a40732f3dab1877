// msm_sort.hpp -- the field-independent kernels of the MSM (included by msm_impl.hpp inside namespace zk): the two-level counting
// sort of the (window, point) digits by bucket and the cut of the sorted list into the runs of the accumulation.  Defined in the
// translation unit that sets ZK_MSM_COMMON (msm_g1.hip), declared in the other.
// Replaces nothing of the reference one to one: its inner products (/root/reference/src/groth16/mod.rs:255-272, 279-290) are n
// independent scalar multiplications; the sort is what turns them into bucket sums.
#pragma once

// Where scalar i of a (possibly grouped, possibly two-array) product lives, and whether it is one: group grp = i / glen, position il in
// the group; il < split: scalars[grp stride1 + il] (valid below valid1), otherwise scalars2[grp stride2 + il - split] (valid below gvalid
// counted from the group's start).  Without a second array split is the largest size_t and stride1 = glen: scalars[i].
struct ScalarSrc {
    const Fr* scalars; const Fr* scalars2;
    size_t split, stride1, stride2;
    uint32_t glen, gvalid, valid1;
    __device__ __forceinline__ bool get(size_t i, uint32_t& grp, uint32_t& il, Fr& k) const {
        grp = (uint32_t)i / glen; il = (uint32_t)i - grp * glen;
        if (il >= gvalid) return false;
        if (il < split) {
            if (il >= valid1) return false;
            k = scalars[(size_t)grp * stride1 + il];
        } else {
            k = scalars2[(size_t)grp * stride2 + (il - split)];
        }
        return true;
    }
};
#ifdef ZK_MSM_COMMON
// ---- two-level counting sort of the digits by bucket -------------------------------------------
// A bucket id (|digit| - 1, c - 1 bits) splits into a bin (high bits, at most 2^10 bins) and a sub-bucket.
// Level 1 (one workgroup per scalar chunk) groups the digits by bin: per (chunk, bin) the 8-byte records
// (sub-bucket, (w*n + i) << 1 | sign) form runs of hundreds of bytes, so the stores fill whole lines --
// a direct counting sort over 2^15 buckets emits 16-byte runs and was bound by partial-line write
// bursts (1.35 ms of a 2^20 proof).  Level 2 (one workgroup per bin) finishes the sort inside a bin
// whose records and 4-byte output both sit in L2, and emits the bucket offsets.  LDS holds only
// 2^10 + 2^(c-11) counters, so the window size is no longer tied to the LDS capacity.

// atomicAdd(&ctr[idx], 1) for all active lanes, with lanes of a wave that hit the SAME counter served by one
// atomic (skewed digit distributions put most lanes of a wave on one or two counters, and same-address LDS
// atomics serialise).  Groups are peeled while they are large; evenly spread indices fall through to the
// plain atomic after one round.
__device__ __forceinline__ uint32_t lds_inc(uint32_t* ctr, uint32_t idx) {
    // Evenly spread indices (every product of random scalars) take the plain atomic after ONE wave-uniform test -- how many active
    // lanes share the first active lane's counter? -- instead of a round of the peeling below: the peeling's bookkeeping was ~10 of the
    // ~35 VALU instructions a digit costs in each of the four kernels that count (round 5: profiles/r5_experiments.txt).
    {
        const uint32_t v0 = __builtin_amdgcn_readfirstlane(idx);
        if (__popcll(__ballot(idx == v0)) < 8) return atomicAdd(&ctr[idx], 1u);
    }
    uint32_t res = 0;
    bool pending = true;
    for (int round = 0; round < 6; ++round) {
        bool big = false;
        if (pending) {
            const uint32_t v = __builtin_amdgcn_readfirstlane(idx);
            if (idx == v) {
                const uint64_t m = __ballot(1);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t cnt = (uint32_t)__popcll(m);
                uint32_t base = 0;
                if (rank == 0) base = atomicAdd(&ctr[v], cnt);
                res = __builtin_amdgcn_readfirstlane(base) + rank;
                pending = false;
                big = cnt >= 8;
            }
        }
        if (!__any(big)) break;
    }
    if (pending) res = atomicAdd(&ctr[idx], 1u);
    return res;
}

// Block-wide exclusive scan: out[i] = sum_{j < i} in[j] for i < count (in and out may alias), using THREADS lanes
// and a THREADS-entry scratch array; returns the total.  Every lane of the block must call it.
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(const uint32_t* in, uint32_t* out, int count, uint32_t* scratch) {
    const int per = (count + THREADS - 1) / THREADS;
    const int lo = min((int)threadIdx.x * per, count), hi = min(lo + per, count);
    uint32_t sum = 0;
    for (int b = lo; b < hi; ++b) sum += in[b];
    scratch[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < THREADS; d <<= 1) {
        uint32_t v = (int)threadIdx.x >= d ? scratch[threadIdx.x - d] : 0;
        __syncthreads();
        scratch[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? scratch[threadIdx.x - 1] : 0;
    const uint32_t total = scratch[THREADS - 1];
    for (int b = lo; b < hi; ++b) {
        uint32_t v = in[b];
        out[b] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// hist[chunk][bin] = number of digits of the chunk whose bucket falls into the bin
// Grouped form (several independent products over the same bases in one pass): scalar i belongs to group i / glen and
// multiplies point i % glen (scalars at or behind gvalid are ignored); group g owns the bins [g bins_pg, (g+1) bins_pg).
// Split form (MsmSplit): scalar i comes from scalars2[i - split] for i >= split.
// Bucket range (partial sums by bucket range, MsmGroups::bucket_shard): only digits whose bucket lies in [blo, blo + 2^bucket_bits) are kept,
// re-indexed from blo; a whole product has blo = 0 and bucket_bits = c - 1.
__global__ __launch_bounds__(SORT_THREADS) void k_msm_hist(ScalarSrc src, size_t n, size_t chunk_len, int c, int windows,
                                                           int first, int step, int sub_bits, int groups, uint32_t blo, int bucket_bits, uint32_t* __restrict__ hist) {
    ZK_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int bins_pg = 1 << (bucket_bits - sub_bits), bins = bins_pg * groups;
#pragma unroll 1
    for (int b = threadIdx.x; b < bins; b += SORT_THREADS) lds[b] = 0;
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * chunk_len, hi = min(lo + chunk_len, n);
    // one scalar at a time (the compiler otherwise keeps two in flight: 50 registers, and four such waves per SIMD -- a 1024-lane
    // workgroup -- do not fit into the 208 registers that one retired accumulation wave leaves)
#pragma unroll 1
    for (size_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
        uint32_t grp, il;
        Fr k;
        if (!src.get(i, grp, il, k)) continue;
        const uint32_t bin0 = grp * (uint32_t)bins_pg;
        for_each_digit_auto(k, c, windows, first, step, [&](int, uint32_t mag, uint32_t) {
            const uint32_t b = mag - 1 - blo;
            if ((b >> bucket_bits) == 0) lds_inc(lds, bin0 + (b >> sub_bits));
        });
    }
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * bins;
#pragma unroll 1
    for (int b = threadIdx.x; b < bins; b += SORT_THREADS) row[b] = lds[b];
}

// total[b] = sum over chunks of hist[chunk][b]
__global__ void k_msm_bin_totals(const uint32_t* __restrict__ hist, int chunks, int bins, uint32_t* __restrict__ total) {
    ZK_LATENCY_KERNEL();
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bins) return;
    uint32_t run = 0;
#pragma unroll 8
    for (int ch = 0; ch < chunks; ++ch) run += hist[(size_t)ch * bins + b];
    total[b] = run;
}

// hist[chunk][b] -> position of the chunk's first record of bin b
__global__ void k_msm_chunk_prefix(uint32_t* __restrict__ hist, int chunks, int bins, const uint32_t* __restrict__ start) {
    ZK_LATENCY_KERNEL();
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bins) return;
    uint32_t run = start[b];
#pragma unroll 8
    for (int ch = 0; ch < chunks; ++ch) {
        uint32_t v = hist[(size_t)ch * bins + b];
        hist[(size_t)ch * bins + b] = run;
        run += v;
    }
}

// exclusive scan of total[0..count) -> start[0..count]; one workgroup
__global__ __launch_bounds__(1024) void k_msm_scan(const uint32_t* __restrict__ total, uint32_t* __restrict__ start, int count) {
    ZK_LATENCY_KERNEL();
    __shared__ uint32_t part[1024];
    int per = (count + 1023) / 1024;
    int lo = min((int)threadIdx.x * per, count), hi = min(lo + per, count);
    uint32_t s = 0;
    for (int b = lo; b < hi; ++b) s += total[b];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        uint32_t v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int b = lo; b < hi; ++b) {
        start[b] = run;
        run += total[b];
    }
    if (threadIdx.x == 1023) start[count] = part[1023];
}

// level 1: records[pos] = (sub-bucket << 32) | ((w*n + i) << 1 | neg), grouped by bin: one 8-byte store per digit at the
// position taken from the bin's LDS counter.  (An LDS-staged form that writes the records in runs, like level 2 below,
// was slower here: 0.85 vs 0.50 ms per proof; with only 2^8 bins the hot lines of a chunk stay in L2.)
__global__ __launch_bounds__(SORT_THREADS) void k_msm_scatter(ScalarSrc src, size_t n, size_t stride, size_t chunk_len, int c, int windows,
                                                                     int first, int step, int sub_bits, int groups, uint32_t blo, int bucket_bits,
                                                                     const uint32_t* __restrict__ prefix, uint64_t* __restrict__ records) {
    ZK_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int bins_pg = 1 << (bucket_bits - sub_bits), bins = bins_pg * groups;
    const uint32_t sub_mask = (1u << sub_bits) - 1;
    const uint32_t* row = prefix + (size_t)blockIdx.x * bins;
    for (int b = threadIdx.x; b < bins; b += SORT_THREADS) lds[b] = row[b];
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * chunk_len, hi = min(lo + chunk_len, n);
    for (size_t i = lo + threadIdx.x; i < hi; i += SORT_THREADS) {
        uint32_t grp, il;
        Fr k;
        if (!src.get(i, grp, il, k)) continue;
        const uint32_t bin0 = grp * (uint32_t)bins_pg;
        for_each_digit_auto(k, c, windows, first, step, [&](int w, uint32_t mag, uint32_t neg) {
            const uint32_t b = mag - 1 - blo;
            if (b >> bucket_bits) return;
            uint32_t pos = lds_inc(lds, bin0 + (b >> sub_bits));
            records[pos] = ((uint64_t)(b & sub_mask) << 32) | (((uint32_t)((size_t)w * stride + il) << 1) | neg);
        });
    }
}

// level 2: counting sort by sub-bucket inside every bin, `parts` workgroups per bin (a bin that holds a
// heavy bucket can be a large share of all records), positions by a per-bin prefix over (sub-bucket, part)
// Level-2 workgroups are dealt to the bins in proportion to their record counts (part_start[b] = first workgroup of bin b): a
// top window of few bits puts ALL its digits into bin 0, a boolean-heavy witness puts them into one bucket; with the same number of
// workgroups for every bin such a bin serialised level 2.  Returns false for surplus workgroups of the (upper-bound) grid.
__device__ __forceinline__ bool bin_slice(const uint32_t* __restrict__ bin_start, const uint32_t* __restrict__ part_start, int bins, uint32_t& lo, uint32_t& hi) {
    const uint32_t g = blockIdx.x;
    if (g >= part_start[bins]) return false;
    int l = 0, h = bins;   // part_start[l] <= g < part_start[h]
    while (h - l > 1) {
        const int mid = (l + h) >> 1;
        if (part_start[mid] <= g) l = mid; else h = mid;
    }
    const uint32_t parts = part_start[l + 1] - part_start[l], part = g - part_start[l];
    const uint32_t s = bin_start[l], len = bin_start[l + 1] - s;
    lo = s + (uint32_t)((uint64_t)len * part / parts);
    hi = s + (uint32_t)((uint64_t)len * (part + 1) / parts);
    return true;
}

// part_start[b] = exclusive scan of ceil(len_b / target) (at least one workgroup per bin); one workgroup
__global__ __launch_bounds__(1024) void k_msm_bin_parts(const uint32_t* __restrict__ bin_start, int bins, uint32_t target, uint32_t* __restrict__ part_start) {
    ZK_LATENCY_KERNEL();
    __shared__ uint32_t scratch[1024];
    for (int b = threadIdx.x; b < bins; b += 1024) {
        const uint32_t len = bin_start[b + 1] - bin_start[b];
        part_start[b] = max(1u, (len + target - 1) / target);
    }
    __syncthreads();
    const uint32_t total = block_exclusive_scan<1024>(part_start, part_start, bins, scratch);
    if (threadIdx.x == 0) part_start[bins] = total;
}

__global__ __launch_bounds__(SORT2_THREADS) void k_msm_bin_hist(const uint64_t* __restrict__ records, const uint32_t* __restrict__ bin_start,
                                                                const uint32_t* __restrict__ part_start, int bins, int sub_bits, uint32_t* __restrict__ cnt) {
    ZK_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int subs = 1 << sub_bits;
    uint32_t lo, hi;
    if (!bin_slice(bin_start, part_start, bins, lo, hi)) return;
    for (int b = threadIdx.x; b < subs; b += SORT2_THREADS) lds[b] = 0;
    __syncthreads();
    // four records in flight per lane: a bin holding a heavy bucket makes this loop long and latency-bound
    for (uint32_t k = lo + threadIdx.x; k < hi; k += 4 * SORT2_THREADS) {
        uint64_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = k + j * SORT2_THREADS < hi ? records[k + j * SORT2_THREADS] : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j * SORT2_THREADS < hi) lds_inc(lds, (uint32_t)(r[j] >> 32) & (uint32_t)(subs - 1));
    }
    __syncthreads();
    uint32_t* row = cnt + (size_t)blockIdx.x * subs;
    for (int b = threadIdx.x; b < subs; b += SORT2_THREADS) row[b] = lds[b];
}

// one workgroup per bin: cnt[bin][part][sub] -> first position of that (part, sub); start[bucket]
__global__ __launch_bounds__(SORT2_THREADS) void k_msm_bin_offsets(uint32_t* __restrict__ cnt, const uint32_t* __restrict__ bin_start, const uint32_t* __restrict__ part_start,
                                                                   int bins, int sub_bits, uint32_t* __restrict__ start) {
    ZK_LATENCY_KERNEL();
    __shared__ uint32_t part_sum[SORT2_THREADS];
    const int subs = 1 << sub_bits;
    const int bin = blockIdx.x;
    const int parts = (int)(part_start[bin + 1] - part_start[bin]);
    uint32_t* rows = cnt + (size_t)part_start[bin] * subs;
    const int per = (subs + SORT2_THREADS - 1) / SORT2_THREADS;
    const int lo = min((int)threadIdx.x * per, subs), hi = min(lo + per, subs);
    uint32_t sum = 0;
    // 2^11 sub-buckets: a lane owns 8 consecutive counters of every part -- two 16-byte accesses, coalesced over the workgroup (read
    // counter by counter the same loop was 72 dependent 4-byte loads at a 32-byte stride per lane and pass: 80 us per launch)
    const bool vec = per == 8 && hi - lo == 8;
    if (vec) {
        uint4 s0 = make_uint4(0, 0, 0, 0), s1 = s0;
        for (int p = 0; p < parts; ++p) {
            const uint4* r4 = reinterpret_cast<const uint4*>(rows + (size_t)p * subs + lo);
            const uint4 a = r4[0], b = r4[1];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        sum = s0.x + s0.y + s0.z + s0.w + s1.x + s1.y + s1.z + s1.w;
    } else {
        for (int b = lo; b < hi; ++b) {
            uint32_t col = 0;
            for (int p = 0; p < parts; ++p) col += rows[(size_t)p * subs + b];
            sum += col;
        }
    }
    part_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < SORT2_THREADS; d <<= 1) {
        uint32_t v = (int)threadIdx.x >= d ? part_sum[threadIdx.x - d] : 0;
        __syncthreads();
        part_sum[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = bin_start[bin] + (threadIdx.x ? part_sum[threadIdx.x - 1] : 0);
    if (vec) {
        // per sub-bucket b the positions run over the parts: first pass the column totals (in registers), then the offsets
        uint32_t col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < parts; ++p) {
            const uint4* r4 = reinterpret_cast<const uint4*>(rows + (size_t)p * subs + lo);
            const uint4 a = r4[0], b = r4[1];
            col[0] += a.x; col[1] += a.y; col[2] += a.z; col[3] += a.w; col[4] += b.x; col[5] += b.y; col[6] += b.z; col[7] += b.w;
        }
        uint32_t at[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { at[i] = run; run += col[i]; }
        uint4* st4 = reinterpret_cast<uint4*>(start + (size_t)bin * subs + lo);
        st4[0] = make_uint4(at[0], at[1], at[2], at[3]);
        st4[1] = make_uint4(at[4], at[5], at[6], at[7]);
        for (int p = 0; p < parts; ++p) {
            uint4* r4 = reinterpret_cast<uint4*>(rows + (size_t)p * subs + lo);
            const uint4 a = r4[0], b = r4[1];
            r4[0] = make_uint4(at[0], at[1], at[2], at[3]);
            r4[1] = make_uint4(at[4], at[5], at[6], at[7]);
            at[0] += a.x; at[1] += a.y; at[2] += a.z; at[3] += a.w; at[4] += b.x; at[5] += b.y; at[6] += b.z; at[7] += b.w;
        }
        if (bin == bins - 1 && threadIdx.x == 0) start[(size_t)bins * subs] = bin_start[bins];
        return;
    }
    for (int b = lo; b < hi; ++b) {
        start[(size_t)bin * subs + b] = run;
        for (int p = 0; p < parts; ++p) {
            uint32_t v = rows[(size_t)p * subs + b];
            rows[(size_t)p * subs + b] = run;
            run += v;
        }
    }
    if (bin == bins - 1 && threadIdx.x == 0) start[(size_t)bins * subs] = bin_start[bins];
}

// level-2 scatter, staged like level 1: a workgroup takes BIN_STAGE records of its slice at a time, counting-sorts
// them by sub-bucket inside LDS and stores runs of consecutive 4-byte entries
__global__ __launch_bounds__(BINS_THREADS) void k_msm_bin_scatter(const uint64_t* __restrict__ records, const uint32_t* __restrict__ bin_start,
                                                                  const uint32_t* __restrict__ part_start, int bins, int sub_bits,
                                                                  const uint32_t* __restrict__ pos_in, uint32_t* __restrict__ sorted) {
    ZK_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t scratch[BINS_THREADS];
    const int subs = 1 << sub_bits;
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem);
    uint16_t* ssub = reinterpret_cast<uint16_t*>(smem + (size_t)BIN_STAGE * 4);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + (size_t)BIN_STAGE * 6);
    uint32_t* lstart = cnt + subs;
    uint32_t* pos = lstart + subs;     // running write position per sub-bucket for this (bin, part)
    uint32_t lo, hi;
    if (!bin_slice(bin_start, part_start, bins, lo, hi)) return;
    const uint32_t* row = pos_in + (size_t)blockIdx.x * subs;
    for (int b = threadIdx.x; b < subs; b += BINS_THREADS) pos[b] = row[b];
    for (uint32_t base = lo; base < hi; base += BIN_STAGE) {
        for (int b = threadIdx.x; b < subs; b += BINS_THREADS) cnt[b] = 0;
        __syncthreads();
        uint32_t ent[BIN_PER_LANE], sub[BIN_PER_LANE], rank[BIN_PER_LANE];
#pragma unroll
        for (int j = 0; j < BIN_PER_LANE; ++j) {
            const uint32_t k = base + j * BINS_THREADS + threadIdx.x;
            const uint64_t r = k < hi ? records[k] : 0;
            ent[j] = (uint32_t)r;
            sub[j] = (uint32_t)(r >> 32) & (uint32_t)(subs - 1);
        }
#pragma unroll
        for (int j = 0; j < BIN_PER_LANE; ++j)
            if (base + j * BINS_THREADS + threadIdx.x < hi) rank[j] = lds_inc(cnt, sub[j]);
        __syncthreads();
        const uint32_t total = block_exclusive_scan<BINS_THREADS>(cnt, lstart, subs, scratch);
#pragma unroll
        for (int j = 0; j < BIN_PER_LANE; ++j)
            if (base + j * BINS_THREADS + threadIdx.x < hi) {
                const uint32_t p = lstart[sub[j]] + rank[j];
                stage[p] = ent[j];
                ssub[p] = (uint16_t)sub[j];
            }
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < total; p += BINS_THREADS) {
            const uint32_t sb = ssub[p];
            sorted[pos[sb] + (p - lstart[sb])] = stage[p];
        }
        __syncthreads();
        for (int b = threadIdx.x; b < subs; b += BINS_THREADS) pos[b] += cnt[b];
        __syncthreads();
    }
}

// ---- runs of the accumulation -----------------------------------------------------------------------
// Bucket b (entries [start[b], start[b + 1]) of the sorted list, z of them) is cut into r = ceil(z / T) runs of floor(z / r) or
// ceil(z / r) <= T entries; one accumulation lane adds one run.  The runs are listed by DECREASING length class (ceil(z / r)):
//   * the 64 lanes of a wave then have the same trip count (+-1) however the bucket sizes are distributed, so no lane idles;
//   * the longest runs start first and the launch drains through its shortest ones;
//   * a bucket that fits one run (z <= T: every bucket of a window whose 2^(c-1) buckets outnumber the entries / T) needs no
//     merging at all -- its image is the bucket sum.  The former equal-slices form (lane t owned entries [32 t, 32 t + 32)
//     whatever buckets they fell into) paid one full addition per lane in k_msm_merge: 4 % of a proof's instructions.
// runs_cnt[len] = number of runs of class len (len <= RUN_MAX); wg_extra[g] = runs beyond the first of the buckets of workgroup g.
__device__ __forceinline__ void bucket_runs(const uint32_t* __restrict__ start, int buckets, uint32_t T, uint32_t b, uint32_t& s, uint32_t& z, uint32_t& r, uint32_t& len) {
    s = z = r = len = 0;
    if (b >= (uint32_t)buckets) return;
    s = start[b];
    z = start[b + 1] - s;
    if (!z) return;
    r = (z + T - 1) / T;
    len = (z + r - 1) / r;
}
__global__ __launch_bounds__(256) void k_msm_runs_count(const uint32_t* __restrict__ start, int buckets, uint32_t T, uint32_t* __restrict__ runs_cnt,
                                                        uint32_t* __restrict__ wg_extra) {
    ZK_LATENCY_KERNEL();
    __shared__ uint32_t h[RUN_MAX + 2];
    for (int i = threadIdx.x; i < RUN_MAX + 2; i += 256) h[i] = 0;
    __syncthreads();
    uint32_t s, z, r, len;
    bucket_runs(start, buckets, T, blockIdx.x * 256 + threadIdx.x, s, z, r, len);
    if (r) atomicAdd(&h[len], r);
    if (r > 1) atomicAdd(&h[RUN_MAX + 1], r - 1);
    __syncthreads();
    for (int i = threadIdx.x; i <= RUN_MAX; i += 256)
        if (h[i]) atomicAdd(&runs_cnt[i], h[i]);
    if (threadIdx.x == 0) wg_extra[blockIdx.x] = h[RUN_MAX + 1];
}
// one workgroup: cursor[len] = position of the first run of class len (classes in decreasing order), wg_xbase = exclusive scan of
// wg_extra, info = {runs, extra images}; runs_cnt is cleared for the next product that uses this workspace
__global__ __launch_bounds__(1024) void k_msm_runs_scan(uint32_t* __restrict__ runs_cnt, uint32_t* __restrict__ cursor, const uint32_t* __restrict__ wg_extra,
                                                        uint32_t* __restrict__ wg_xbase, int wgs, uint32_t* __restrict__ info) {
    ZK_LATENCY_KERNEL();
    __shared__ uint32_t scratch[1024];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int len = RUN_MAX; len >= 1; --len) {
            cursor[len] = run;
            run += runs_cnt[len];
            runs_cnt[len] = 0;
        }
        cursor[0] = run;
        info[0] = run;
    }
    const uint32_t extras = block_exclusive_scan<1024>(wg_extra, wg_xbase, wgs, scratch);
    if (threadIdx.x == 0) info[1] = extras;
}
__global__ __launch_bounds__(256) void k_msm_runs_emit(const uint32_t* __restrict__ start, int buckets, uint32_t T, uint32_t* __restrict__ cursor,
                                                       const uint32_t* __restrict__ wg_xbase, MsmRun* __restrict__ runs, uint32_t* __restrict__ xbase) {
    ZK_LATENCY_KERNEL();
    __shared__ uint32_t h[RUN_MAX + 1], gb[RUN_MAX + 1], scratch[256];
    for (int i = threadIdx.x; i <= RUN_MAX; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    uint32_t s, z, r, len;
    bucket_runs(start, buckets, T, b, s, z, r, len);
    const uint32_t rank = r ? atomicAdd(&h[len], r) : 0;
    // exclusive scan of the buckets' extra runs inside the workgroup
    const uint32_t x = r > 1 ? r - 1 : 0;
    scratch[threadIdx.x] = x;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = (int)threadIdx.x >= d ? scratch[threadIdx.x - d] : 0;
        __syncthreads();
        scratch[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t xb = wg_xbase[blockIdx.x] + scratch[threadIdx.x] - x;
    for (int i = threadIdx.x; i <= RUN_MAX; i += 256) gb[i] = h[i] ? atomicAdd(&cursor[i], h[i]) : 0;
    __syncthreads();
    // A bucket of many runs (a heavy bucket: evaluation-basis scalars of a structured circuit repeat one value over every gate, a
    // boolean witness fills one bucket) is written by the whole workgroup -- one lane looping over 32768 descriptors took 9 ms.
    __shared__ uint32_t big[256][6];
    __shared__ uint32_t nbig;
    if (threadIdx.x == 0) nbig = 0;
    __syncthreads();
    if (r) {
        xbase[b] = xb;
        const uint32_t pos = gb[len] + rank;
        if (r > 32) {
            const uint32_t slot = atomicAdd(&nbig, 1u);
            big[slot][0] = b; big[slot][1] = s; big[slot][2] = z; big[slot][3] = r; big[slot][4] = pos; big[slot][5] = xb;
        } else {
            uint32_t k0 = s;
            for (uint32_t j = 0; j < r; ++j) {
                const uint32_t k1 = s + (uint32_t)((uint64_t)z * (j + 1) / r);
                runs[pos + j] = MsmRun{k0, k1 - k0, j == 0 ? b : (uint32_t)buckets + xb + j - 1, k1};
                k0 = k1;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = 0; i < nbig; ++i) {
        const uint32_t bb = big[i][0], bs = big[i][1], bz = big[i][2], br = big[i][3], bpos = big[i][4], bxb = big[i][5];
        for (uint32_t j = threadIdx.x; j < br; j += 256) {
            const uint32_t k0 = bs + (uint32_t)((uint64_t)bz * j / br), k1 = bs + (uint32_t)((uint64_t)bz * (j + 1) / br);
            runs[bpos + j] = MsmRun{k0, k1 - k0, j == 0 ? bb : (uint32_t)buckets + bxb + j - 1, k1};
        }
    }
}

#else
__global__ void k_msm_hist(ScalarSrc, size_t, size_t, int, int, int, int, int, int, uint32_t, int, uint32_t*);
__global__ void k_msm_bin_totals(const uint32_t*, int, int, uint32_t*);
__global__ void k_msm_chunk_prefix(uint32_t*, int, int, const uint32_t*);
__global__ void k_msm_scan(const uint32_t*, uint32_t*, int);
__global__ void k_msm_scatter(ScalarSrc, size_t, size_t, size_t, int, int, int, int, int, int, uint32_t, int, const uint32_t*, uint64_t*);
__global__ void k_msm_bin_parts(const uint32_t*, int, uint32_t, uint32_t*);
__global__ void k_msm_bin_hist(const uint64_t*, const uint32_t*, const uint32_t*, int, int, uint32_t*);
__global__ void k_msm_bin_offsets(uint32_t*, const uint32_t*, const uint32_t*, int, int, uint32_t*);
__global__ void k_msm_bin_scatter(const uint64_t*, const uint32_t*, const uint32_t*, int, int, const uint32_t*, uint32_t*);
__global__ void k_msm_runs_count(const uint32_t*, int, uint32_t, uint32_t*, uint32_t*);
__global__ void k_msm_runs_scan(uint32_t*, uint32_t*, const uint32_t*, uint32_t*, int, uint32_t*);
__global__ void k_msm_runs_emit(const uint32_t*, int, uint32_t, uint32_t*, const uint32_t*, MsmRun*, uint32_t*);
#endif  // ZK_MSM_COMMON
