// msm_impl.hpp -- fixed-base Pippenger multi-scalar multiplication in G1 and G2 for gfx950.
//
// Replaces the SigmaG1/SigmaG2 inner products of groth16::prove
// (/root/reference/src/groth16/mod.rs:255-272,279-290): the reference performs n independent
// 256-bit double-and-add scalar multiplications (exp_encrypted_g1/g2,
// /root/reference/src/groth16/fr.rs:114-119) and folds them sequentially (Sum for G1Local/G2Local,
// fr.rs:191-198,217-223).  The sum is a group element, so any evaluation order gives the same
// affine point.
//
// MI355X-first design.  The bases are CRS points, fixed across proofs, and the GPU has 288 GB of
// HBM, so every base P_i is stored W times as T[w][i] = 2^(c w) P_i (W = floor(254/c)+1 signed
// c-bit windows).  All windows then share ONE set of 2^(c-1) buckets:
//     sum_i k_i P_i = sum_b b * ( sum_{(w,i): |digit_w(k_i)| = b} sign * T[w][i] )
// which removes the per-window bucket reductions and the serial 254-doubling Horner tail of the
// textbook algorithm.  Per MSM:
//   sort       two-level counting sort of the (window, point) digits by bucket: level 1 groups
//              8-byte records by bin (high bits of the bucket id, one workgroup per scalar chunk,
//              counters in LDS), level 2 sorts inside each bin by sub-bucket and emits the bucket
//              offsets; signed-digit recoding happens on the fly in both level-1 passes
//   runs       every bucket is cut into ceil(size / T) runs of (nearly) equal length <= T; the runs of all buckets are
//              listed by decreasing length (a counting sort over <= 128 length classes)
//   accumulate lane t adds the entries of run t (gathered 64 B / 128 B points) into a register-resident
//              accumulator in the lazy radix-2^29 form (lazy29.cuh): the 64 lanes of a wave have runs of
//              (almost) the same length, the longest runs start first, and a bucket that is one run needs nothing more
//   merge      buckets of several runs: sum of their images (a workgroup per bucket for heavy buckets)
//   reduce     sum_b b*S_b by rows and columns of the bucket index (b - 1 = hi K + lo): plain sums give the K column
//              sums C_lo and the 2^(c-1)/K row sums R_hi (2 additions per bucket, all parallel, no doubling chains),
//              sum_b b S_b = sum_lo lo C_lo + sum_hi (hi K + 1) R_hi is left for ~2 sqrt(buckets) points
// Multi-GPU partial sums: rank g of a job owns windows w = g (mod world) (the digit
// loop skips other windows), or a range of the points with every window (point_offset).
#pragma once
#include <type_traits>
#include "kernels.hpp"
#include "lazy29.cuh"
#include "quad29.cuh"

namespace zk {

#ifdef ZK_MSM_COMMON
int msm_auto_window(size_t n) {
    // Measured on MI355X (profiles/r1_window_sweep_2p20.jsonl and the r1 sweeps in DESIGN.md 4c): every
    // scalar costs one addition per window, so wider windows win until the 2^(c-1) buckets' merge /
    // running-sum tail and the sort's sub-bucket level outweigh the saved window: c = 17 (15 windows)
    // beats 16 (16 windows) by 5 % at 2^20..2^21 points; 18 has as many windows as 17, and 19/20 lose
    // again in the pipelined prover.  A narrow top window (254 - (W-1) c bits) is harmless since the
    // accumulation is balanced per lane, so small sizes just scale c with log2(n).
    // Round 2: with the row / column reduction tail (2 independent additions per bucket, both fold chains in one launch per pass)
    // c = 20 -- 13 windows -- wins for the products of 2^21 points and more (L and H + r B1 + s A at 2^20 gates): +1.3 .. 2.5 % on
    // the pipelined prover against 17 on two boxes; 19 is level, 21 / 22 lose 4 / 6 %, and a wider window for the 2^20-point
    // products (A; B in G2: 18 / 19 / 20 measured -7 / -8 / -3 %) does not pay (tools/ab_g2_window.sh).
    // Round 5: c = 20 from 2^20 points on -- the A product of a 2^20-gate proof.  With L merged into the H product the proof folds one
    // set of 2^19 buckets less, and 13 instead of 15 windows for A now pay: 100.8 against 99.5 and 102.0 against 101.5 proofs/s on two
    // boxes (profiles/r5_experiments.txt items 5, 6); c = 19 loses 7 %.
    if (n + 8 >= ((size_t)1 << 20)) return 20;
    int lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) ++lg;
    if (lg >= 17) return 17;
    if (lg >= 16) return 16;
    if (lg >= 14) return 15;
    if (lg >= 11) return 13;
    return 8;
}
// The G2 table: from 2^20 points on c = 20 as well.  An addition over Fq2 is three over Fq, so the saved windows (13 instead of 15)
// weigh more against the 2^19 bucket images of the tail, and since round 3 a bucket of <= 64 entries is ONE run of the accumulation
// with nothing to merge: same-box A/B at 2^20 gates 93.5 against 92.0 proofs/s (profiles/r3_ab.txt).  c = 20 for the n-point G1
// product (A) as well is level (92.9).
int msm_auto_window_g2(size_t n) {
    if (n + 8 >= ((size_t)1 << 20)) return 20;
    return msm_auto_window(n);
}
#endif  // ZK_MSM_COMMON

constexpr int SORT_THREADS = 1024;   // level-1 workgroups (one per scalar chunk)
constexpr int SORT2_THREADS = 256;   // level-2 histogram / offsets workgroups (several per bin)
constexpr int BINS_THREADS = 512;    // level-2 scatter workgroups
constexpr int BIN_STAGE = 8192;      // ... and the records they stage in LDS at a time
constexpr int BIN_PER_LANE = BIN_STAGE / BINS_THREADS;
constexpr int RUN_MAX = 256;         // longest run of the accumulation (length classes 1..RUN_MAX)

// one run of the accumulation: entries [k0, k0 + len) of the bucket-sorted list, all of one bucket; the image of their sum goes to
// slot `dest` (the bucket's own slot for its first run, a slot behind the buckets' for the others)
struct MsmRun { uint32_t k0, len, dest, end; };

// ---- table precompute: T[w][i] = 2^(c w) P_i ------------------------------------------------
// One lane per point: c doublings per window in Jacobian coordinates, then ONE inversion for all of the point's windows (Montgomery's
// trick along the lane's own chain: the Z of every window and the running products go through two scratch arrays, windows x points,
// coalesced over the lanes).  Round 5 inverted per window -- 12 Fermat inversions per point were three quarters of the kernel, and the
// tables are the 0.34 s of the 0.43 s a first proof over a new CRS took.  Points [first, first + count) per launch.
template <class F>
__global__ __launch_bounds__(64) void k_msm_precompute(const Aff<F>* __restrict__ pts, size_t n, size_t first, size_t count, int c, int windows,
                                                       Aff<F>* __restrict__ table, F* __restrict__ zbuf, F* __restrict__ pbuf) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const size_t i = first + t;
    const Aff<F> p = pts[i];
    table[i] = p;
    if (p.is_inf()) {
        for (int w = 1; w < windows; ++w) table[(size_t)w * n + i] = Aff<F>::infinity();
        return;
    }
    // the doublings run in the lazy radix (lazy29.cuh dbl_lazy: no conversion to and from 8 x 32 per multiplication); a window's point
    // leaves it once, fully reduced, for the shared inversion
    JacR<F> jr = jacr_load(Jac<F>::from_affine(p));
    F run = F::one();
    for (int w = 1; w < windows; ++w) {
        for (int k = 0; k < c; ++k) jr = dbl_lazy(jr);
        const Jac<F> j = jacr_store(jr);
        // (a doubling chain over a point of odd order never meets infinity; a caller's arbitrary point may: kept out of the product)
        const bool inf = j.is_inf();
        table[(size_t)w * n + i] = inf ? Aff<F>::infinity() : Aff<F>{j.X, j.Y};   // Jacobian X, Y until the second loop
        const F z = inf ? F::one() : j.Z;
        zbuf[(size_t)(w - 1) * count + t] = z;
        pbuf[(size_t)(w - 1) * count + t] = run;                                  // z_1 ... z_(w-1)
        run = run * z;
    }
    F inv = run.inv();                                                            // 1 / (z_1 ... z_(W-1))
    for (int w = windows - 1; w >= 1; --w) {
        const F zi = inv * pbuf[(size_t)(w - 1) * count + t];                     // 1 / z_w
        inv = inv * zbuf[(size_t)(w - 1) * count + t];
        const Aff<F> q = table[(size_t)w * n + i];
        if (!q.is_inf()) {
            const F zi2 = zi.sqr();
            table[(size_t)w * n + i] = Aff<F>{q.x * zi2, q.y * zi2 * zi};
        }
    }
}

template <class F>
void msm_build_table(zk_ctx* ctx, const Aff<F>* d_points, size_t n, int c, MsmTable<F>& t) {
    ZK_REQUIRE(c >= 2 && c <= MSM_MAX_C, ZK_ERR_ARG, "msm: window_bits must be in [2, 22]");
    t.c = c;
    t.windows = 254 / c + 1;
    t.n = n;
    ZK_REQUIRE((size_t)t.windows * std::max<size_t>(n, 1) < ((size_t)1 << 31), ZK_ERR_SIZE, "msm: too many points for this window size");
    t.table.alloc(std::max<size_t>((size_t)t.windows * n, 1));
    if (!n) return;
    ProfScope ps(ctx, sizeof(F) > sizeof(Fq) ? "msm_precompute_g2" : "msm_precompute_g1", (double)sizeof(Aff<F>) * n * (t.windows + 1));
    // scratch of the shared inversion: 2 x (windows - 1) field elements per point of a launch, at most ~0.8 GiB (2^20 points in G1)
    const size_t chunk = std::min<size_t>(n, ((size_t)1 << 25) / sizeof(F));
    DevBuf<F> zbuf(std::max<size_t>((size_t)(t.windows - 1) * chunk, 1)), pbuf(std::max<size_t>((size_t)(t.windows - 1) * chunk, 1));
    for (size_t first = 0; first < n; first += chunk) {
        const size_t count = std::min(chunk, n - first);
        hipLaunchKernelGGL(k_msm_precompute<F>, dim3(ceil_div(count, 64)), dim3(64), 0, ctx->stream, d_points, n, first, count, c, t.windows, t.table.p, zbuf.p, pbuf.p);
    }
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(ctx->stream));   // the scratch arrays go out of scope here (one-off set-up path)
}
template void msm_build_table<ZK_MSM_FIELD>(zk_ctx*, const Aff<ZK_MSM_FIELD>*, size_t, int, MsmTable<ZK_MSM_FIELD>&);

// ---- signed-digit recoding -------------------------------------------------------------------
// Calls f(w, mag, neg) for every owned window with a non-zero digit.  The scalar is consumed by shifting the 256-bit value right
// by c per window (8 v_alignbit): indexing its limbs with a run-time word number sent the copy to scratch and cost the level-1
// kernels 48 B of scratch and 50 registers -- four such waves per SIMD did not fit beside the bucket accumulation's waves.
template <class Fn>
__device__ __forceinline__ void for_each_digit(const Fr& k, int c, int windows, int first, int step, Fn&& f) {
    uint32_t l0 = k.l[0], l1 = k.l[1], l2 = k.l[2], l3 = k.l[3], l4 = k.l[4], l5 = k.l[5], l6 = k.l[6], l7 = k.l[7];
    uint32_t carry = 0;
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    int next_owned = first;
    for (int w = 0; w < windows; ++w) {
        const uint32_t raw = (l0 & mask) + carry;
        l0 = __builtin_amdgcn_alignbit(l1, l0, c); l1 = __builtin_amdgcn_alignbit(l2, l1, c);
        l2 = __builtin_amdgcn_alignbit(l3, l2, c); l3 = __builtin_amdgcn_alignbit(l4, l3, c);
        l4 = __builtin_amdgcn_alignbit(l5, l4, c); l5 = __builtin_amdgcn_alignbit(l6, l5, c);
        l6 = __builtin_amdgcn_alignbit(l7, l6, c); l7 >>= c;
        const uint32_t neg = raw > half;
        const uint32_t mag = neg ? (1u << c) - raw : raw;
        carry = neg;
        if (w == next_owned) {
            if (mag) f(w, mag, neg);
            next_owned += step;
        }
    }
}

// The same with the window size known at compile time: digit w is bits [C w, C w + C) of the scalar, ONE v_alignbit of two limbs whose
// numbers are constants (the recursion below is the unrolled window loop), instead of the 8 v_alignbit that move the whole value:
// 7 instead of 14 instructions per digit in both level-1 passes of the sort.  C = 20 and 17 (the windows of every product of 2^20
// points and more) are instantiated; other sizes take the generic form.
template <int C, int w, class Fn>
__device__ __forceinline__ void digit_step_c(const Fr& k, uint32_t& carry, int& next_owned, int step, Fn& f) {
    if constexpr (w < 254 / C + 1) {
        constexpr int bit = C * w, lw = bit >> 5, s = bit & 31;
        constexpr uint32_t mask = (1u << C) - 1, half = 1u << (C - 1);
        uint32_t bits = k.l[lw];
        if constexpr (s != 0) bits = lw + 1 < 8 ? __builtin_amdgcn_alignbit(k.l[lw + 1 < 8 ? lw + 1 : 7], bits, s) : bits >> s;
        const uint32_t raw = (bits & mask) + carry;
        const uint32_t neg = raw > half;
        const uint32_t mag = neg ? (1u << C) - raw : raw;
        carry = neg;
        if (w == next_owned) {
            if (mag) f(w, mag, neg);
            next_owned += step;
        }
        digit_step_c<C, w + 1>(k, carry, next_owned, step, f);
    }
}
template <int C, class Fn>
__device__ __forceinline__ void for_each_digit_c(const Fr& k, int first, int step, Fn&& f) {
    uint32_t carry = 0;
    int next_owned = first;
    digit_step_c<C, 0>(k, carry, next_owned, step, f);
}
// dispatch on the run-time window size (uniform over the launch)
template <class Fn>
__device__ __forceinline__ void for_each_digit_auto(const Fr& k, int c, int windows, int first, int step, Fn&& f) {
    if (c == 20) for_each_digit_c<20>(k, first, step, f);
    else if (c == 17) for_each_digit_c<17>(k, first, step, f);
    else for_each_digit(k, c, windows, first, step, f);
}

#include "msm_sort.hpp"   // the counting sort and the runs (shared, field-independent kernels)

// ---- bucket accumulation: one run per lane -------------------------------------------------------
// Waves per SIMD the accumulate kernel is compiled for.  G1 (the asm bodies of madd_asm.inc: 141 VGPRs) runs three -- a fourth
// (128 VGPRs, -DZK_ACC_G1_WAVES=4) buys no rate on a chip that is bound by its package power and takes the registers the other
// kernels' waves start in (profiles/r6_experiments.txt item 1); G2 (XYZZ over Fq2: 72 accumulator limbs) runs two at 256 VGPRs.
#ifndef ZK_ACC_G1_WAVES
#define ZK_ACC_G1_WAVES 3
#endif
template <class F> struct AccWaves { static constexpr int value = ZK_ACC_G1_WAVES; };
template <> struct AccWaves<Fq2> { static constexpr int value = 2; };

// register image of an accumulator as it is parked in HBM between accumulate, merge and the folds
template <class F>
struct alignas(16) AccSlot {
    typename AccOf<F>::type a;
};

// Lane t adds the entries of run t (k_msm_runs_emit) and stores the image of their sum in the run's slot.  info[0] = number of runs.
// ONE loop per field (the forms rounds 3-5 tried are on record in tools/experiments/msm_accumulate_forms_r5.hpp.txt):
//   G1  trips 1 and 2 through the generic step (the run's first point starts the accumulator, the second joins it as an affine + affine
//       addition, madd_xyzz_second); then the FAST loop over the generated whole-addition bodies of madd_asm.inc, two additions per trip.
//   G2  the next point's gather goes straight into LDS (global_load_lds_dwordx4: no destination registers -- 32 registers of a kernel
//       that sits at the 256-register limit), trips 1 and 2 peeled the same way.
template <class F>
__global__ __launch_bounds__(256, AccWaves<F>::value) void k_msm_accumulate(const Aff<F>* __restrict__ table, const uint32_t* __restrict__ sorted,
                                                       const MsmRun* __restrict__ runs, const uint32_t* __restrict__ info, AccSlot<F>* __restrict__ img) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= info[0]) return;
    const MsmRun run = runs[tid];
    const uint32_t k1 = run.k0 + run.len;
    typedef typename LazyOf<F>::type L;
    typename AccOf<F>::type acc;
    acc_clear(acc);
    uint32_t k = run.k0;
    uint32_t e = sorted[k];
    uint32_t e_next = k + 1 < k1 ? sorted[k + 1] : 0;
    if constexpr (sizeof(F) > sizeof(Fq)) {
        // ---- G2: the software pipeline runs through LDS.  Lane t's point lands in rows [j][t] of a per-workgroup stage (16 B per row
        // and lane: what the instruction writes for a wave is 64 consecutive 16-byte slots from the address in M0), is read back at the
        // top of the next trip and converted at once.
        constexpr int ROWS = (int)(sizeof(Aff<F>) / 16);
        __shared__ int4 stage[ROWS][256];
        int4* const wave_rows = &stage[0][threadIdx.x & ~63u];
        auto issue = [&](uint32_t entry) {
            const int4* src = reinterpret_cast<const int4*>(table + (entry >> 1));
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + j),
                                                 (void __attribute__((address_space(3)))*)(wave_rows + j * 256), 16, 0, 0);
        };
        // the point of entry k (its transfer was issued a trip earlier); then the transfer of entry k + 1 into the same rows
        auto fetch = [&](uint32_t& ce) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            union { Aff<F> pt; int4 q[ROWS]; } u;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) u.q[j] = stage[j][threadIdx.x];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are read before the next transfer may overwrite them
            ce = e;
            const uint32_t kn = k + 1;
            if (kn < k1) issue(e_next);
            e = e_next;
            e_next = kn + 1 < k1 ? sorted[kn + 1] : 0;
            k = kn;
            return u.pt;
        };
        issue(e);
        auto start = [&](const L& qx, const L& qy) { acc.X = qx; acc.Y = qy.norm(); acc.ZZ = acc.ZZZ = L::load(F::one()); acc.inf = false; };
        // Trips 1 and 2 peeled: the run's first point starts the accumulator, the second joins it as an affine + affine addition
        // (madd_xyzz_second: 6 of 10 multiplications).  A padding entry in front falls through to the loop.  (Its own scope and its
        // own temporaries: written with shared ones, the hot loop below went back to spilling.)
        {
            uint32_t ce;
            const Aff<F> p1 = fetch(ce);
            if (!p1.is_inf()) {
                {
                    L qx = L::load(p1.x), qy = L::load(p1.y);
                    if (ce & 1) qy = qy.neg();
                    start(qx, qy);
                }
                if (k < k1) {
                    const Aff<F> p2 = fetch(ce);
                    if (!p2.is_inf()) {
                        L qx = L::load(p2.x), qy = L::load(p2.y);
                        if (ce & 1) qy = qy.neg();
                        const int st = madd_xyzz_second(acc, qx, qy);
                        if (st == 1) acc_load(acc, jac_dbl(acc_store(acc)));
                        else if (st == 2) acc_clear(acc);
                    }
                }
            }
        }
        while (k < k1) {
            uint32_t ce;
            const Aff<F> pt = fetch(ce);
            if (pt.is_inf()) continue;                      // points at infinity: unused entries of a table
            L qx = L::load(pt.x), qy = L::load(pt.y);
            if (ce & 1) qy = qy.neg();
            if (acc.inf) { start(qx, qy); continue; }       // first finite point of the run, or the one behind a P + (-P)
            const int st = madd_xyzz_nz(acc, qx, qy);
            if (st == 1) acc_load(acc, jac_dbl(acc_store(acc)));   // same point twice in one bucket: doubling through the generic formulas (rare)
            else if (st == 2) acc_clear(acc);
        }
    } else {
        // ---- G1.  The generic step: entry k's point is in registers, the next one's gather is issued before the addition.
        Aff<F> p = table[e >> 1];
        bool fresh = false;
        auto step = [&](auto second) {
            const uint32_t kn = k + 1;
            const Aff<F> p_next = kn < k1 ? table[e_next >> 1] : Aff<F>::infinity();
            const uint32_t e_next2 = kn + 1 < k1 ? sorted[kn + 1] : 0;
            if (!p.is_inf()) {
                L qx = L::load(p.x), qy = L::load(p.y);
                if (e & 1) qy = qy.neg();
                bool done = false;
                if constexpr (decltype(second)::value) {
                    if (fresh && !acc.inf) {          // the accumulator still is the run's first point: affine + affine
                        const int st = madd_xyzz_second(acc, qx, qy);
                        if (st == 1) acc_load(acc, jac_dbl(acc_store(acc)));
                        else if (st == 2) acc.inf = true;   // P + (-P): the sum is infinity (the coordinates stay, as madd_xyzz leaves them)
                        done = true;
                    }
                }
                if (!done && !acc_madd(acc, qx, qy)) acc_load(acc, jac_dbl(acc_store(acc)));   // same point twice in one bucket (rare)
            }
            p = p_next; e = e_next; e_next = e_next2; k = kn;
        };
        if (k < k1) {
            step(std::false_type{});
            fresh = !acc.inf;
            if (k < k1) step(std::true_type{});
        }
#if ZK_MONT_ASM_ON
        // The FAST loop (madd_asm.inc, tools/gen_madd_asm.py): the whole mixed addition is one asm body whose multipliers work in place.
        // X changes place once per addition and Y changes sign, so the loop is two halves: the even body takes X from acc.X and leaves X3
        // in xb and -Y3 in acc.Y, the odd body takes both back -- no copy of the accumulator, no negation (round 5's loop: 36 + 9 moves
        // per addition).  Anything unusual makes the LANE leave the loop (kf = k) and the generic loop behind it finish the run: a table
        // entry at infinity, an empty accumulator, and a point with the accumulator's x -- then the accumulator WAS +-the point, the
        // body has overwritten it, and the sum (2 P or infinity) is rebuilt from the affine point alone.
        uint32_t kf = acc.inf ? k : k1;       // the fast loop's bound for this lane
        L xb = acc.X;
        const uint32_t lane = __lane_id();
        uint32_t ev = 0, ev_k = 0;            // a point with the accumulator's x met at position ev_k of the list: 1 = opposite points, 2 = the same point
        bool odd_state = false;               // the lane's accumulator is as the even body leaves it (X in xb, -Y in acc.Y)
        // The next point's gather is issued BEHIND the conversion of the current one, into the registers the point has just left (one
        // set of sixteen, no rotation), and has the whole addition to arrive.  The loads are unconditional: behind the end of the run
        // the entry index is clamped to the run's last entry (a valid table index whose point is then not used).
        const uint32_t klast = k1 - 1;
        while (k < kf) {
            // ---- even half: acc.X -> xb, Y -> -Y
            if (p.y.l[0] == 0 && p.is_inf()) {
                kf = k;                                                      // canonical state, entry k not consumed
            } else {
                L qx = L::load(p.x), qy = L::load(p.y);
                if (e & 1) qy = qy.neg();
                p = table[e_next >> 1];
                e = e_next;
                e_next = sorted[min(k + 2, klast)];
                uint64_t sx, sp;
                madd_asm_g1_even<FqParams>(acc.X.v, acc.Y.v, acc.ZZ.v, acc.ZZZ.v, qx.v, qy.v, xb.v, sx, sp);
                if (sx) {
                    asm volatile("" : "+s"(sx));                             // the lane test stays behind the (scalar) branch
                    if ((sx >> lane) & 1) { ev = 1 + (uint32_t)((sp >> lane) & 1); ev_k = k; kf = k + 1; }
                }
                ++k;
                odd_state = true;
            }
            // ---- odd half: xb -> acc.X, -Y -> Y
            if (k < kf) {
                if (p.y.l[0] == 0 && p.is_inf()) {
                    kf = k;                                                  // odd state, entry k not consumed
                } else {
                    L qx = L::load(p.x), qy = L::load(p.y);
                    if (e & 1) qy = qy.neg();
                    p = table[e_next >> 1];
                    e = e_next;
                    e_next = sorted[min(k + 2, klast)];
                    uint64_t sx, sp;
                    madd_asm_g1_odd<FqParams>(xb.v, acc.Y.v, acc.ZZ.v, acc.ZZZ.v, qx.v, qy.v, acc.X.v, sx, sp);
                    if (sx) {
                        asm volatile("" : "+s"(sx));
                        if ((sx >> lane) & 1) { ev = 1 + (uint32_t)((sp >> lane) & 1); ev_k = k; kf = k + 1; }
                    }
                    ++k;
                    odd_state = false;
                }
            }
        }
        if (odd_state) { acc.X = xb; acc.Y = acc.Y.neg().norm(); }
        if (ev) {
            acc_clear(acc);
            if (ev == 2) {
                const uint32_t ev_e = sorted[ev_k];
                const Aff<F> pt = table[ev_e >> 1];
                L qx = L::load(pt.x), qy = L::load(pt.y);
                if (ev_e & 1) qy = qy.neg();
                acc.X = qx; acc.Y = qy.norm(); acc.ZZ = acc.ZZZ = L::load(F::one()); acc.inf = false;
                acc_load(acc, jac_dbl(acc_store(acc)));
            }
        }
#endif
        while (k < k1) step(std::false_type{});   // lanes that left the fast loop (rare); the whole run in a build without the asm bodies
    }
    img[run.dest].a = acc;
}

// S_b = sum of the images of bucket b's runs: its own slot img[b] and the r - 1 slots from img[buckets + xbase[b]] on.  One lane per
// bucket; buckets of more than MSM_HEAVY runs (skewed digit distributions) are queued for k_msm_merge_heavy instead.  Empty buckets
// get the image of infinity here (no run wrote their slot).
constexpr uint32_t MSM_HEAVY = 96;
constexpr uint32_t MSM_HEAVY_CHUNK = 2048;   // images one workgroup of k_msm_merge_heavy sums (a bucket of 32768 runs as ONE workgroup's job took 3 ms)
// Workgroup size of the reduction tail (merge / fold / weigh).  ONE wave: while an accumulation fills the chip, a 256-lane
// workgroup of a 150..250-register kernel needs all four SIMDs of a CU to have room at the same moment, which only happens in
// the accumulation's last round (the timeline showed the previous proof's G2 tail still running 8 ms after its accumulation
// and the next proof's sort waiting behind it on the same stream); a single wave fits wherever one accumulation wave retires.
constexpr int TAIL_THREADS = 64;
// Registers of the tail kernels, as waves per SIMD they are compiled for.  A wave can only start on a SIMD that has its whole
// register allocation free: beside three waves of the G1 accumulation (3 x 152 of 512 registers) that is 56, after one of them
// retired 208 -- and beside the G2 accumulation (2 x 256) nothing until a wave retires.  A tail wave of 246..256 registers (what the
// Fq2 kernels take when left alone) therefore waits for TWO accumulation waves of one SIMD to retire together, which only happens when
// an accumulation drains: the kernel trace showed the four-wave k_msm_sum_points<Fq2> taking 1.2 ms instead of 0.28 and the G2 tail
// as a whole ending with the last accumulation of its proof.
// Same-box A/B at 2^20 gates with the G2 table at c = 20 (2^19 bucket images to fold): Fq2 tail kernels left at 246..256 registers
// 91.1 proofs/s, compiled for three waves per SIMD (168 registers, 500..1100 B of scratch) 94.3, for four (128 registers) 91.9
// (profiles/r3_experiments.txt): at 168 a tail wave fits into the 208 registers one retired G1 accumulation wave leaves, the G2
// tail no longer ends with the last accumulation of its proof and the next proof's G2 sort, queued behind it, starts in time.
#ifndef ZK_TAIL_G2_WAVES
#define ZK_TAIL_G2_WAVES 3
#endif
#ifndef ZK_TAIL_G1_WAVES
#define ZK_TAIL_G1_WAVES 3
#endif
template <class F> struct TailWaves { static constexpr int value = ZK_TAIL_G1_WAVES; };
template <> struct TailWaves<Fq2> { static constexpr int value = ZK_TAIL_G2_WAVES; };

template <class F>
__global__ __launch_bounds__(TAIL_THREADS, TailWaves<F>::value) void k_msm_merge(const uint32_t* __restrict__ start, int buckets, uint32_t T, const uint32_t* __restrict__ xbase,
                                                   AccSlot<F>* __restrict__ img, uint32_t* __restrict__ heavy, uint32_t heavy_cap) {
    ZK_LATENCY_KERNEL();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (uint32_t)buckets) return;
    const uint32_t z = start[b + 1] - start[b];
    typename AccOf<F>::type acc;
    if (!z) { acc_clear(acc); img[b].a = acc; return; }
    const uint32_t r = (z + T - 1) / T;
    if (r == 1) return;
    if (r - 1 > MSM_HEAVY) {   // queued in chunks of MSM_HEAVY_CHUNK images: (bucket, chunk) items behind heavy[2], multi-chunk buckets from the end
        const uint32_t chunks = (r + MSM_HEAVY_CHUNK - 1) / MSM_HEAVY_CHUNK, at = atomicAdd(&heavy[0], chunks);
        for (uint32_t w = 0; w < chunks; ++w) { heavy[2 + 2 * (at + w)] = b; heavy[3 + 2 * (at + w)] = w; }
        if (chunks > 1) heavy[heavy_cap - 1 - atomicAdd(&heavy[1], 1u)] = b;
        return;
    }
    const AccSlot<F>* more = img + (size_t)buckets + xbase[b];
    acc = img[b].a;
    for (uint32_t j = 0; j + 1 < r; ++j) acc = acc_add(acc, more[j].a);
    img[b].a = acc;
}

constexpr int MSM_HEAVY_THREADS = 128;
// heavy buckets.  Image j of bucket b: its own slot (j = 0) or slot j - 1 behind img[buckets + xbase[b]].
// FINAL = false: one workgroup per queued (bucket, chunk) item sums the chunk's images (lanes stride over them, tree over LDS) into the
// chunk's first image.  FINAL = true: one workgroup per multi-chunk bucket sums the chunks' first images into the bucket's slot.
// (registers as the other tail kernels, TailWaves: left alone the Fq2 instance takes 256, and -- launched behind EVERY accumulation,
// normally to find nothing to do -- waited ~1 ms for two accumulation waves of a SIMD to retire together, with the rest of the G2 tail
// and the next proof's G2 sort queued behind it)
template <class F, bool FINAL>
__global__ __launch_bounds__(MSM_HEAVY_THREADS, TailWaves<F>::value) void k_msm_merge_heavy(const uint32_t* __restrict__ start, int buckets, uint32_t T, const uint32_t* __restrict__ xbase,
                                                         AccSlot<F>* __restrict__ img, const uint32_t* __restrict__ heavy, uint32_t heavy_cap) {
    ZK_LATENCY_KERNEL();
    // 38 KiB for G2 images.  Dynamic: with a static array the compiler sees that LDS already limits the kernel to two waves per SIMD
    // and does not bother to stay within the three-wave register budget
    extern __shared__ __attribute__((aligned(16))) uint8_t heavy_smem[];
    AccSlot<F>* sh = reinterpret_cast<AccSlot<F>*>(heavy_smem);
    const uint32_t count = FINAL ? heavy[1] : heavy[0];
    for (uint32_t h = blockIdx.x; h < count; h += gridDim.x) {
        const uint32_t b = FINAL ? heavy[heavy_cap - 1 - h] : heavy[2 + 2 * h], w = FINAL ? 0 : heavy[3 + 2 * h];
        const uint32_t z = start[b + 1] - start[b], r = (z + T - 1) / T;
        AccSlot<F>* more = img + (size_t)buckets + xbase[b];
        const uint32_t lo = FINAL ? 0 : w * MSM_HEAVY_CHUNK, hi = FINAL ? r : min(lo + MSM_HEAVY_CHUNK, r), step = FINAL ? MSM_HEAVY_CHUNK : 1;
        typename AccOf<F>::type acc;
        acc_clear(acc);
        for (uint32_t j = lo + threadIdx.x * step; j < hi; j += MSM_HEAVY_THREADS * step) acc = acc_add(acc, j == 0 ? img[b].a : more[j - 1].a);
        sh[threadIdx.x].a = acc;
        __syncthreads();
        for (int d = MSM_HEAVY_THREADS / 2; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d) sh[threadIdx.x].a = acc_add(sh[threadIdx.x].a, sh[threadIdx.x + d].a);
            __syncthreads();
        }
        if (threadIdx.x == 0) { if (lo == 0) img[b].a = sh[0].a; else more[lo - 1].a = sh[0].a; }
        __syncthreads();
    }
}

// ---- sum_b b S_b over the bucket images ---------------------------------------------------------
// Bucket b of a group has index b - 1 = hi K + lo (K = 2^kbits columns, rows = 2^(c-1) / K).  With the column sums
// C_lo = sum_hi S[hi][lo] and the row sums R_hi = sum_lo S[hi][lo]:
//     sum_b b S_b = sum_lo lo C_lo + sum_hi (hi K + 1) R_hi.
// Column and row sums are plain sums: 2 additions per bucket, every one of them independent of the others (the running-sum
// form has the same count but needs a k * P of ~c bits per segment of buckets on top, which made 2^19 buckets cost more than
// the window they save).  What is left to weight is K + rows ~ 2 sqrt(buckets) points.
//
// out[a B + b] = sum_{i < f} in[(a f + i) B + b], a < A, b < B: one lane per output image, f - 1 dependent additions.
// Columns: the row index is folded (B = K); rows: the column index is folded (B = 1); group indices ride in `a`.
// One launch serves one pass of BOTH chains (column sums and row sums are independent): workgroups [0, j0.blocks) run job 0,
// the rest job 1 -- half the dependent launches of the tail, which is what a lone proof of a small circuit waits for.
struct FoldJob { const void* in; void* out; uint32_t A, f, B, blocks; };
#ifndef ZK_FOLD_G2_WAVES
#define ZK_FOLD_G2_WAVES ZK_TAIL_G2_WAVES
#endif
// G2 folds with Y, ZZ and ZZZ of the running sum in LDS between their uses.  The general addition over Fq2 holds 233 registers at
// once; capped at the tail kernels' 168 the by-value form (add_xyzz) spilled 700 B per lane -- 1.9 GB of scratch traffic per proof of
// 2^20 gates, a ninth of everything a proof moved -- and the coordinate-by-coordinate form (add_xyzz_from, ZK_FOLD_PARK=0) still 53
// dwords per addition (0.7 GB).  Parked: 15 KB of LDS per workgroup, 20 ds_read_b128 + 15 ds_write_b128 per addition, no scratch on
// the path of an ordinary addition.  Same box, 5 x 20 steps interleaved: by value 102.53, by coordinate 103.76, parked 103.49,
// by value at 2 waves per SIMD (233 registers, ZK_FOLD_G2_WAVES=2) 103.06 proofs/s; HBM bytes per proof 17.51 / 16.41 / 15.70 /
// 15.62 GB (profiles/r5_experiments.txt item 14).
#ifndef ZK_FOLD_PARK
#define ZK_FOLD_PARK 1
#endif
struct FoldPark { int4 r[3][5][TAIL_THREADS]; };   // [Y | ZZ | ZZZ][row][lane]
__device__ __forceinline__ void fpark_put(FoldPark* pk, int which, const Fp2R<FqParams>& v) {
    int4* row = &pk->r[which][0][threadIdx.x];
    row[0 * TAIL_THREADS] = make_int4(v.c0.v[0], v.c0.v[1], v.c0.v[2], v.c0.v[3]);
    row[1 * TAIL_THREADS] = make_int4(v.c0.v[4], v.c0.v[5], v.c0.v[6], v.c0.v[7]);
    row[2 * TAIL_THREADS] = make_int4(v.c0.v[8], v.c1.v[0], v.c1.v[1], v.c1.v[2]);
    row[3 * TAIL_THREADS] = make_int4(v.c1.v[3], v.c1.v[4], v.c1.v[5], v.c1.v[6]);
    row[4 * TAIL_THREADS] = make_int4(v.c1.v[7], v.c1.v[8], 0, 0);
}
__device__ __forceinline__ Fp2R<FqParams> fpark_get(const FoldPark* pk, int which) {
    asm volatile("" ::: "memory");   // a fresh read every time: the point is NOT to keep the value in registers
    const int4* row = &pk->r[which][0][threadIdx.x];
    const int4 a = row[0 * TAIL_THREADS], b = row[1 * TAIL_THREADS], c = row[2 * TAIL_THREADS], d = row[3 * TAIL_THREADS], e = row[4 * TAIL_THREADS];
    Fp2R<FqParams> v;
    v.c0.v[0] = a.x; v.c0.v[1] = a.y; v.c0.v[2] = a.z; v.c0.v[3] = a.w; v.c0.v[4] = b.x; v.c0.v[5] = b.y; v.c0.v[6] = b.z; v.c0.v[7] = b.w;
    v.c0.v[8] = c.x; v.c1.v[0] = c.y; v.c1.v[1] = c.z; v.c1.v[2] = c.w; v.c1.v[3] = d.x; v.c1.v[4] = d.y; v.c1.v[5] = d.z; v.c1.v[6] = d.w;
    v.c1.v[7] = e.x; v.c1.v[8] = e.y;
    return v;
}
// add_xyzz_from (lazy29.cuh) with the sum's Y / ZZ / ZZZ in LDS: same formulas, same order, same bounds
__device__ __forceinline__ void add_xyzz_from_parked(Fp2R<FqParams>& X, bool& inf, FoldPark* pk, const XyzzR<Fp2R<FqParams>>* q) {
    typedef Fp2R<FqParams> L;
    if (q->inf) return;
    if (inf) {
        X = q->X; fpark_put(pk, 0, q->Y); fpark_put(pk, 1, q->ZZ); fpark_put(pk, 2, q->ZZZ);
        inf = false;
        return;
    }
    L U1, P;
    {
        const L qzz = q->ZZ;
        U1 = X * qzz;
    }
    asm volatile("" ::: "memory");
    {
        const L qx = q->X;
        P = qx * fpark_get(pk, 1) - U1;
    }
    asm volatile("" ::: "memory");
    const L PP = P.sqr();
    if (PP.is_zero_mod_p()) {
        const L qy = q->Y;
        const bool same = (qy * fpark_get(pk, 2) - fpark_get(pk, 0) * q->ZZZ).sqr().is_zero_mod_p();
        if (same) {
            XyzzR<L> t = *q;
            dbl_xyzz(t);
            X = t.X; fpark_put(pk, 0, t.Y); fpark_put(pk, 1, t.ZZ); fpark_put(pk, 2, t.ZZZ);
        } else inf = true;
        return;
    }
    {
        const L qzz = q->ZZ;
        fpark_put(pk, 1, (fpark_get(pk, 1) * qzz) * PP);
    }
    asm volatile("" ::: "memory");
    const L Q = U1 * PP;
    const L PPP = P * PP;
    L S1, R;
    {
        const L qzzz = q->ZZZ;
        S1 = fpark_get(pk, 0) * qzzz;
        const L Z3 = fpark_get(pk, 2);
        const L T = Z3 * qzzz;
        asm volatile("" ::: "memory");
        const L qy = q->Y;
        R = qy * Z3 - S1;
        fpark_put(pk, 2, T * PPP);
    }
    asm volatile("" ::: "memory");
    const L X3 = (R.sqr() - PPP - (Q + Q)).norm();
    fpark_put(pk, 0, xyzz_ydiff(R, Q - X3, S1, PPP));
    X = X3;
}
template <class F> struct FoldWaves { static constexpr int value = TailWaves<F>::value; };
template <> struct FoldWaves<Fq2> { static constexpr int value = ZK_FOLD_G2_WAVES; };
template <class F>
__global__ __launch_bounds__(TAIL_THREADS, FoldWaves<F>::value) void k_msm_fold(FoldJob j0, FoldJob j1) {
    ZK_LATENCY_KERNEL();
    const bool second = blockIdx.x >= j0.blocks;
    const AccSlot<F>* in = reinterpret_cast<const AccSlot<F>*>(second ? j1.in : j0.in);
    AccSlot<F>* out = reinterpret_cast<AccSlot<F>*>(second ? j1.out : j0.out);
    const uint32_t A = second ? j1.A : j0.A, f = second ? j1.f : j0.f, B = second ? j1.B : j0.B;
    const uint32_t j = (second ? blockIdx.x - j0.blocks : blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= A * B) return;
    const uint32_t a = j / B, b = j - a * B;
    const AccSlot<F>* src = in + (size_t)a * f * B + b;
    if constexpr (sizeof(F) > sizeof(Fq) && ZK_FOLD_PARK != 0) {
        __shared__ FoldPark pk;
        Fp2R<FqParams> X = src[0].a.X;
        bool inf = src[0].a.inf;
        fpark_put(&pk, 0, src[0].a.Y); fpark_put(&pk, 1, src[0].a.ZZ); fpark_put(&pk, 2, src[0].a.ZZZ);
        for (uint32_t i = 1; i < f; ++i) add_xyzz_from_parked(X, inf, &pk, &src[(size_t)i * B].a);
        out[j].a.X = X; out[j].a.Y = fpark_get(&pk, 0); out[j].a.ZZ = fpark_get(&pk, 1); out[j].a.ZZZ = fpark_get(&pk, 2);
        out[j].a.inf = inf;
        return;
    }
    typename AccOf<F>::type acc = src[0].a;
    if constexpr (sizeof(F) > sizeof(Fq)) {
        for (uint32_t i = 1; i < f; ++i) add_xyzz_from(acc, &src[(size_t)i * B].a);
    } else {
        for (uint32_t i = 1; i < f; ++i) acc = acc_add(acc, src[(size_t)i * B].a);
    }
    out[j].a = acc;
}

// term[group][j]: j < K: lo * C[lo] (lo = j); j = K + hi: (hi K + 1) * R[hi].  One point per lane, every lane runs the same
// double-and-add (weight 0 gives infinity); k_msm_sum_points adds the K + rows terms of a group.
template <class F>
__global__ __launch_bounds__(TAIL_THREADS, TailWaves<F>::value) void k_msm_weigh(const AccSlot<F>* __restrict__ C, const AccSlot<F>* __restrict__ R, int kbits, int rows,
                                                   uint32_t wbase, Jac<F>* __restrict__ term) {
    ZK_LATENCY_KERNEL();
    const int K = 1 << kbits, g = blockIdx.y;
    const int j = blockIdx.x * TAIL_THREADS + threadIdx.x;
    if (j >= K + rows) return;
    const AccSlot<F>* src = j < K ? C + (size_t)g * K + j : R + (size_t)g * rows + (j - K);
    const uint32_t w = j < K ? (uint32_t)j : ((uint32_t)(j - K) << kbits) + 1u + wbase;   // wbase: first bucket of a bucket-range shard
    term[(size_t)g * (K + rows) + j] = jacr_store(mul_small_lazy(jacr_load(acc_store(src->a)), w));
}

// sums points: workgroup g of G adds in[g], in[g + G], ... (256 lanes, tree over LDS in the 8 x 32 form) -> out[g]
// blockIdx.y = group: its `count` inputs start at in + y count, its gridDim.x outputs at (bytes) out + y out_stride
template <class F>
__global__ __launch_bounds__(256, TailWaves<F>::value) void k_msm_sum_points(const Jac<F>* __restrict__ in, int count, Jac<F>* __restrict__ out, size_t out_stride) {
    ZK_LATENCY_KERNEL();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Jac<F>* sh = reinterpret_cast<Jac<F>*>(smem);
    in += (size_t)blockIdx.y * count;
    out = reinterpret_cast<Jac<F>*>(reinterpret_cast<uint8_t*>(out) + (size_t)blockIdx.y * out_stride);
    JacR<F> acc = jacr_load(Jac<F>::infinity());
    for (int k = blockIdx.x * 256 + threadIdx.x; k < count; k += 256 * gridDim.x) acc = add_lazy(acc, jacr_load(in[k]));
    sh[threadIdx.x] = jacr_store(acc);
    __syncthreads();
    // lanes at and behind `share` hold infinity: the tree starts at the first level that has something to add (a small product has
    // a few dozen terms, and every level is a full Jacobian addition on the critical path of a lone proof)
    const int share = min(max(count - (int)blockIdx.x * 256, 1), 256);
    int d0 = 128;
    while (d0 >= 2 && d0 >= share) d0 >>= 1;
    for (int d = d0; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sh[threadIdx.x] = jacr_store(add_lazy(jacr_load(sh[threadIdx.x]), jacr_load(sh[threadIdx.x + d])));
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

#include "msm_quad.hpp"   // the same tail with four lanes per addition, for products of few buckets

template <class F>
hipStream_t msm_run(zk_ctx* ctx, MsmWorkspace& ws, hipStream_t st, const MsmTable<F>& tab, const Fr* d_scalars, size_t n_used,
                    int rank, int world, Jac<F>* d_out, hipEvent_t acc_wait, hipEvent_t acc_done, size_t point_offset, const MsmGroups& grp, const MsmSplit& sp) {
    const bool g2 = sizeof(F) > sizeof(Fq);
    const int c = tab.c, windows = tab.windows;
    // Partial sums by BUCKET RANGE (grp.bucket_shard; world a power of two with at least 64 buckets per rank, otherwise by windows):
    // this rank's bucket set is 2^(cb - 1) buckets from blo on; every window's digits are looked at, 1 / world of them kept.
    int shard_log = 0;
    if (grp.bucket_shard && world > 1 && (world & (world - 1)) == 0 && grp.groups == 1) {
        while ((1 << shard_log) < world) ++shard_log;
        if (c - 1 - shard_log < 6) shard_log = 0;
    }
    const bool bshard = shard_log > 0;
    const int cb = c - shard_log, bpg = 1 << (cb - 1);
    const uint32_t blo = bshard ? (uint32_t)rank * (uint32_t)bpg : 0u;
    if (bshard) { rank = 0; world = 1; }       // windows: all of them
    const size_t n = tab.n;
    // grouped: `groups` products over the same bases [point_offset, point_offset + valid), scalars of group j at
    // d_scalars + j glen, result j at (bytes) d_out + j out_stride; bucket (group, |digit|) = group * 2^(c-1) + |digit| - 1
    const int groups = grp.groups;
    const size_t glen = groups > 1 ? grp.glen : std::max<size_t>(n_used, 1), gvalid = groups > 1 ? grp.valid : n_used;
    if (groups > 1) n_used = (size_t)groups * glen;
    ZK_REQUIRE(groups >= 1 && (groups == 1 || gvalid <= glen), ZK_ERR_ARG, "msm: bad grouping");
    // two scalar arrays (MsmSplit): a group's valid scalars are [0, valid1) of the first part and [split, split + n2) of the second
    ZK_REQUIRE(!sp.scalars2 || (sp.split <= gvalid && gvalid - sp.split == sp.n2 && (groups == 1 || (sp.stride1 >= sp.split && sp.stride2 >= sp.n2))),
               ZK_ERR_ARG, "msm: bad scalar split");
    ScalarSrc src;
    src.scalars = d_scalars; src.scalars2 = sp.scalars2;
    src.split = sp.scalars2 ? sp.split : ~(size_t)0;
    src.stride1 = sp.scalars2 && groups > 1 ? sp.stride1 : glen;
    src.stride2 = sp.scalars2 && groups > 1 ? sp.stride2 : 0;
    src.glen = (uint32_t)glen; src.gvalid = (uint32_t)gvalid;
    src.valid1 = sp.scalars2 ? (uint32_t)std::min(sp.valid1, sp.split) : (uint32_t)gvalid;
    ZK_REQUIRE(point_offset <= n && gvalid <= n - point_offset, ZK_ERR_ARG, "msm: more scalars than table points");
    ZK_REQUIRE(n_used < ((size_t)1 << 32) && (size_t)groups * bpg <= ((size_t)1 << 24), ZK_ERR_SIZE, "msm: too many scalars or buckets");
    const int buckets = bpg * groups;
    int owned = 0;
    for (int w = rank; w < windows; w += world) ++owned;
    if (gvalid == 0 || owned == 0) {
        static const Jac<F> inf = Jac<F>::infinity();
        for (int j = 0; j < groups; ++j)
            ZK_HIP(hipMemcpyAsync(reinterpret_cast<uint8_t*>(d_out) + (size_t)j * grp.out_stride, &inf, sizeof(inf), hipMemcpyHostToDevice, st));
        if (acc_wait) ZK_HIP(hipStreamWaitEvent(st, acc_wait, 0));
        if (acc_done) ZK_HIP(hipEventRecord(acc_done, st));
        return st;
    }
    // two-level sort: 2^8 bins per group (fewer when the window is narrow, more only to keep the sub-bucket level at 2^11 counters)
    const int sub_bits = std::min(11, std::max(0, cb - 1 - 8)), bins = (1 << (cb - 1 - sub_bits)) * groups;
    int chunks = (int)std::min<size_t>((size_t)ctx->cu_count, (n_used + SORT_THREADS - 1) / SORT_THREADS);
    if (chunks < 1) chunks = 1;
    size_t chunk_len = (n_used + chunks - 1) / chunks;
    chunks = (int)((n_used + chunk_len - 1) / chunk_len);
    const size_t entries = (size_t)owned * n_used;   // upper bound of the digits kept (a bucket-range shard keeps ~ 1 / 2^shard_log of them)
    // positions in the sorted list (start[], the scan totals, the runs) are 32-bit, and the level-1
    // counters of all bins live in LDS: refuse up front instead of wrapping silently / failing after the first launches
    ZK_REQUIRE(entries < ((size_t)1 << 32), ZK_ERR_SIZE, "msm: scalars x windows exceeds 2^32 digit records (use a wider window or fewer groups)");
    ZK_REQUIRE((size_t)bins * 4 <= 65536, ZK_ERR_SIZE, "msm: too many groups for this window size (level-1 counters exceed 64 KiB of LDS)");
    // Longest run T of the accumulation.  When the buckets are many and small (entries / buckets <= 64: every product of 2^20 points
    // and more at its automatic window) T = RUN_MAX, so that a bucket is ONE run and nothing is merged -- the lanes are then as many as
    // the non-empty buckets.  Otherwise T = msm_run_entries (32): a bucket of z entries becomes ceil(z / 32) lanes.  A product with
    // few entries (small circuits, the tails of a sharded proof) is a latency chain of T dependent additions on a fraction of the
    // chip -- 32 additions over Fq2 are 0.6 ms whether 16 or 2^16 scalars are multiplied -- so its runs are shorter, down to 4.
    uint32_t T = (uint32_t)std::max<long>(4, std::min<long>(ctx->opt_run_entries, RUN_MAX) & ~3L);
    const size_t fill = (size_t)std::max<long>(ctx->opt_small_lanes, 0);
    // (one run of at most 128 entries up to 64 entries per bucket on average -- the separate products of 2^20 .. 2^21 points --, of at
    // most RUN_MAX = 256 beyond: the merged L + H product of a proof holds 104 per bucket, and 128 would cut one bucket in a hundred)
    const size_t entries_est = entries >> shard_log;   // what the run-length rules are tuned on
    // (a bucket-range shard holds as many entries per bucket as the whole product in 1 / N of the buckets: one run per bucket would leave
    // the chip 1 / N of its lanes, so its buckets are cut into runs like a small product's -- ms per proof and rank at N = 2 / 4 / 8:
    // 5.77 / 4.06 / 3.20 -> 5.51 / 3.65 / 2.75, profiles/r6_experiments.txt item 6)
    const bool whole = entries_est / (size_t)buckets <= (size_t)std::max<long>(ctx->opt_run_whole, 0) && (size_t)buckets >= fill && !bshard;
    if (whole) T = entries_est / (size_t)buckets <= 64 ? 128 : RUN_MAX;
    else if (ctx->opt_run_fill && entries_est / T > (size_t)(g2 ? 2 : 3) * 256 * (size_t)ctx->cu_count) {
        // More runs than the chip holds lanes (3 waves per SIMD in G1, 2 in G2): every run beyond a bucket's first costs a full
        // XYZZ + XYZZ addition in the merge, so the runs are made as long as one round of lanes allows.  The A product of a 2^20-gate
        // proof (c = 17: 240 entries per bucket): T = 80 instead of 32, 2 merges per bucket instead of 7 (round 5).
        const size_t lanes = (size_t)(g2 ? 2 : 3) * 256 * (size_t)ctx->cu_count;
        T = (uint32_t)std::min<size_t>(RUN_MAX, (entries_est / lanes + 3) & ~(size_t)3);
    }
    else if (fill && entries_est / T < fill) T = (uint32_t)std::max<size_t>(4, std::min<size_t>(T, entries_est / fill) & ~(size_t)3);
    // upper bounds: a bucket of z entries has ceil(z / T) <= 1 + z / T runs, of which all but the first take an extra image slot
    const size_t max_extra = entries / T + 1, max_runs = std::min<size_t>((size_t)buckets, entries) + max_extra;
    // an accumulation that cannot fill the chip (3 waves per SIMD = 196608 lanes) is not chained behind the previous one
    if (std::min(max_runs, entries_est / std::min<size_t>(T, 32) + 1) < (size_t)std::max<long>(ctx->opt_unchain_lanes, 0)) acc_wait = nullptr;
    // rows x columns of the bucket index for the final weighted sum (see k_msm_fold)
    const int kbits = cb / 2, K = 1 << kbits, rows = bpg >> kbits;   // ceil((cb - 1) / 2) column bits
    const int wgs_w = (K + rows + TAIL_THREADS - 1) / TAIL_THREADS;
    // Few buckets: the tail is a chain of dependent additions on lanes that have nothing else to do, so four lanes share each
    // addition (msm_quad.hpp: ~3x shorter chains).  Many buckets (the products of 2^20 points and more, batches): one lane per addition.
    const bool quad = (size_t)buckets <= (size_t)std::max<long>(ctx->opt_quad_buckets, 0);
    // Whatever the size of the product, the END of its tail is a few thousand lanes in long dependency chains (the last fold passes,
    // the weighted terms, their sum): those take the quad form too.  Stand-alone at 2^20 gates the weights of the G2 product took 0.81 ms
    // and its final sum 0.39 with one lane per addition.
    const bool quad_end = ctx->opt_quad_buckets > 0;
    constexpr size_t QUAD_FOLD_JOBS = 16384;   // fold passes of at most this many output images
    const int run_wgs = (int)ceil_div(buckets, 256);

    ws.hist.ensure((size_t)chunks * bins);
    ws.total.ensure(bins);
    ws.bin_start.ensure(bins + 1);
    ws.records.ensure(entries);
    ws.start.ensure(buckets + 1);
    ws.sorted.ensure(entries + 4);
    if (!ws.runs_cnt.p) {   // cleared once; k_msm_runs_scan leaves it cleared
        ws.runs_cnt.alloc(2 * (RUN_MAX + 2) + 2);
        ZK_HIP(hipMemsetAsync(ws.runs_cnt.p, 0, ws.runs_cnt.bytes(), st));
    }
    uint32_t* const d_cursor = ws.runs_cnt.p + (RUN_MAX + 2);
    uint32_t* const d_info = ws.runs_cnt.p + 2 * (RUN_MAX + 2);
    ws.wg_extra.ensure(2 * (size_t)run_wgs);
    uint32_t* const d_wg_xbase = ws.wg_extra.p + run_wgs;
    ws.xbase.ensure(buckets);
    ws.runs.ensure(max_runs * 4);
    MsmRun* const d_runs = reinterpret_cast<MsmRun*>(ws.runs.p);
    ws.bucket_sums.ensure(((size_t)buckets + max_extra) * sizeof(AccSlot<F>));              // S_b as accumulator images | extra runs
    const size_t half = ((size_t)buckets + 1) / 2, quarter = ((size_t)buckets + 3) / 4;
    ws.fold.ensure((2 * (half + quarter) + (size_t)groups * (K + rows)) * sizeof(AccSlot<F>));   // per chain: passes 1, 3, .. | passes 2, 4, ..; then C | R
    ws.seg_sums.ensure((size_t)groups * (K + rows + wgs_w + (K + rows) / QUAD_SUM_THREADS + 2) * std::max(sizeof(Jac<F>), sizeof(AccSlot<F>)));   // terms | partial sums
    const size_t heavy_cap = 2 + 2 * ((size_t)buckets + max_extra / MSM_HEAVY_CHUNK + 1) + (size_t)buckets;   // count, count | (bucket, chunk) items | multi-chunk buckets, from the end
    ws.heavy.ensure(heavy_cap);
    AccSlot<F>* d_img = reinterpret_cast<AccSlot<F>*>(ws.bucket_sums.p);
    AccSlot<F>* const fold_base = reinterpret_cast<AccSlot<F>*>(ws.fold.p);
    AccSlot<F>* d_tmp[4] = {fold_base, fold_base + half, fold_base + half + quarter, fold_base + 2 * half + quarter};
    AccSlot<F>* d_C = fold_base + 2 * (half + quarter);
    AccSlot<F>* d_R = d_C + (size_t)groups * K;
    Jac<F>* d_seg = reinterpret_cast<Jac<F>*>(ws.seg_sums.p);
    const double pt_bytes = (double)sizeof(Aff<F>);

#ifdef ZK_MEASURE
    // measurement aid (option ablate, bit 0; ZK_MEASURE builds only): when the caller repeats the same scalars, the sorted list this
    // workspace holds from the previous call is reused -- prices the whole sort in the pipelined prover
    const bool reuse_sort = (ctx->opt_ablate & 1) && ws.sorted_for == (uint64_t)entries * 31 + (uint64_t)buckets + ((uint64_t)T << 48);
    ws.sorted_for = (uint64_t)entries * 31 + (uint64_t)buckets + ((uint64_t)T << 48);
#else
    const bool reuse_sort = false;
#endif
    if (!reuse_sort) {
        {
            ProfScope ps(ctx, "msm_hist", 32.0 * n_used + 4.0 * chunks * bins, st);
            hipLaunchKernelGGL(k_msm_hist, dim3(chunks), dim3(SORT_THREADS), (size_t)bins * 4, st, src, n_used, chunk_len, c, windows, rank, world, sub_bits,
                               groups, blo, cb - 1, ws.hist.p);
        }
        {
            ProfScope ps(ctx, "msm_offsets", 12.0 * chunks * bins, st);
            hipLaunchKernelGGL(k_msm_bin_totals, dim3(ceil_div(bins, 64)), dim3(64), 0, st, ws.hist.p, chunks, bins, ws.total.p);
            hipLaunchKernelGGL(k_msm_scan, dim3(1), dim3(1024), 0, st, ws.total.p, ws.bin_start.p, bins);
            hipLaunchKernelGGL(k_msm_chunk_prefix, dim3(ceil_div(bins, 64)), dim3(64), 0, st, ws.hist.p, chunks, bins, ws.bin_start.p);
        }
        {
            ProfScope ps(ctx, "msm_scatter", 32.0 * n_used + 8.0 * entries + 4.0 * chunks * bins, st);
            hipLaunchKernelGGL(k_msm_scatter, dim3(chunks), dim3(SORT_THREADS), (size_t)bins * 4, st, src, n_used, n, chunk_len, c, windows, rank, world,
                               sub_bits, groups, blo, cb - 1, ws.hist.p, ws.records.p);
        }
        {
            ProfScope ps(ctx, "msm_sort_bins", 20.0 * entries + 4.0 * buckets, st);
            // ~2048 level-2 workgroups in all, dealt to the bins in proportion to their records (at least BIN_STAGE records each)
            const int subs = 1 << sub_bits;
            const uint32_t target = (uint32_t)std::max<size_t>(BIN_STAGE, (entries_est + 2047) / 2048);
            const unsigned grid2 = (unsigned)(entries / target + 1) + (unsigned)bins;   // >= sum_b max(1, ceil(len_b / target))
            ws.bin_cnt.ensure((size_t)grid2 * subs);
            ws.part_start.ensure((size_t)bins + 1);
            hipLaunchKernelGGL(k_msm_bin_parts, dim3(1), dim3(1024), 0, st, ws.bin_start.p, bins, target, ws.part_start.p);
            hipLaunchKernelGGL(k_msm_bin_hist, dim3(grid2), dim3(SORT2_THREADS), (size_t)subs * 4, st, ws.records.p, ws.bin_start.p, ws.part_start.p, bins, sub_bits, ws.bin_cnt.p);
            hipLaunchKernelGGL(k_msm_bin_offsets, dim3(bins), dim3(SORT2_THREADS), 0, st, ws.bin_cnt.p, ws.bin_start.p, ws.part_start.p, bins, sub_bits, ws.start.p);
            hipLaunchKernelGGL(k_msm_bin_scatter, dim3(grid2), dim3(BINS_THREADS), (size_t)BIN_STAGE * 6 + (size_t)subs * 12, st, ws.records.p, ws.bin_start.p, ws.part_start.p, bins,
                               sub_bits, ws.bin_cnt.p, ws.sorted.p);
        }
        {
            ProfScope ps(ctx, "msm_runs", 8.0 * buckets + 12.0 * max_runs, st);
            hipLaunchKernelGGL(k_msm_runs_count, dim3(run_wgs), dim3(256), 0, st, ws.start.p, buckets, T, ws.runs_cnt.p, ws.wg_extra.p);
            hipLaunchKernelGGL(k_msm_runs_scan, dim3(1), dim3(1024), 0, st, ws.runs_cnt.p, d_cursor, ws.wg_extra.p, d_wg_xbase, run_wgs, d_info);
            hipLaunchKernelGGL(k_msm_runs_emit, dim3(run_wgs), dim3(256), 0, st, ws.start.p, buckets, T, d_cursor, d_wg_xbase, d_runs, ws.xbase.p);
        }
    }
    {
        // algorithmic bytes as SURVEY.md 8(d) prices an inner product: one 32-byte scalar and one affine point per (scalar, point)
        // pair -- 96 B in G1, 160 B in G2 -- whatever the window count.  (What this implementation actually gathers is W times
        // that: a 4-byte index and a 64 / 128-byte table entry per window and pair, plus one image per run; bench.py
        // reports that figure and the PMC-measured traffic beside the 8(d) number.)
        if (acc_wait) ZK_HIP(hipStreamWaitEvent(st, acc_wait, 0));
        {
            ProfScope ps(ctx, g2 ? "msm_accumulate_g2" : "msm_accumulate_g1", (32.0 + pt_bytes) * (double)groups * (double)gvalid, st);
            const Aff<F>* acc_table = tab.table.p + point_offset;
            const uint32_t* acc_sorted = ws.sorted.p;
            hipLaunchKernelGGL(k_msm_accumulate<F>, dim3(ceil_div(max_runs, 256)), dim3(256), 0, st, acc_table, acc_sorted, d_runs, d_info, d_img);
        }
        if (acc_done) ZK_HIP(hipEventRecord(acc_done, st));
    }
    if (grp.tail_stream && acc_done) {   // the next product of this stream (the next proof's sort) does not queue behind this tail
        ZK_HIP(hipStreamWaitEvent(grp.tail_stream, acc_done, 0));
        st = grp.tail_stream;
    }
    {
        ProfScope ps(ctx, g2 ? "msm_reduce_g2" : "msm_reduce_g1", (double)sizeof(AccSlot<F>) * (max_extra + 5.0 * buckets), st);
        ZK_HIP(hipMemsetAsync(ws.heavy.p, 0, 2 * sizeof(uint32_t), st));
        if (quad) hipLaunchKernelGGL(k_msm_merge_q<F>, dim3(ceil_div(buckets, QUAD_JOBS)), dim3(QUAD_THREADS), 0, st, ws.start.p, buckets, T, ws.xbase.p, d_img, ws.heavy.p, (uint32_t)heavy_cap);
        else hipLaunchKernelGGL(k_msm_merge<F>, dim3(ceil_div(buckets, TAIL_THREADS)), dim3(TAIL_THREADS), 0, st, ws.start.p, buckets, T, ws.xbase.p, d_img, ws.heavy.p, (uint32_t)heavy_cap);
        // heavy buckets are outliers when the average bucket is a few runs (a small grid that mostly finds nothing to do -- surplus
        // workgroups read the count and leave; a narrow top window makes 2^(top bits) buckets heavy at once); with few buckets and many
        // entries (small windows) nearly every bucket is heavy
        const unsigned heavy_grid = entries / T / (size_t)buckets > MSM_HEAVY / 2 ? (unsigned)std::min(buckets, 4096) : 256u;
        hipLaunchKernelGGL((k_msm_merge_heavy<F, false>), dim3(heavy_grid), dim3(MSM_HEAVY_THREADS), MSM_HEAVY_THREADS * sizeof(AccSlot<F>), st, ws.start.p, buckets, T, ws.xbase.p, d_img, ws.heavy.p, (uint32_t)heavy_cap);
        hipLaunchKernelGGL((k_msm_merge_heavy<F, true>), dim3(64), dim3(MSM_HEAVY_THREADS), MSM_HEAVY_THREADS * sizeof(AccSlot<F>), st, ws.start.p, buckets, T, ws.xbase.p, d_img, ws.heavy.p, (uint32_t)heavy_cap);
        // column sums C[g][lo] (fold the row index, FOLD images per lane and pass), then row sums R[g][hi] (fold the column index)
        uint32_t FOLD = 2;   // images per lane and pass (msm_fold option), a power of two
        while (FOLD * 2 <= (uint32_t)std::max<long>(2, std::min<long>(ctx->opt_fold, 64))) FOLD *= 2;
        // passes of the two chains, zipped: chain 0 = columns (count = rows, B = K, outer = groups), chain 1 = rows
        struct Chain { uint32_t count, B, outer; const AccSlot<F>* in; AccSlot<F>* fin; AccSlot<F>* tmp[2]; int tog; bool done; };
        Chain ch[2] = {{(uint32_t)rows, (uint32_t)K, (uint32_t)groups, d_img, d_C, {d_tmp[0], d_tmp[1]}, 0, false},
                       {(uint32_t)K, 1u, (uint32_t)groups * (uint32_t)rows, d_img, d_R, {d_tmp[2], d_tmp[3]}, 0, false}};
        while (!ch[0].done || !ch[1].done) {
            FoldJob job[2];
            for (int q = 0; q < 2; ++q) {
                Chain& h = ch[q];
                if (h.done) { job[q] = FoldJob{nullptr, nullptr, 0, 1, 1, 0}; continue; }
                const uint32_t f = std::min(h.count, FOLD);
                AccSlot<F>* out = h.count == f ? h.fin : h.tmp[h.tog];
                // rows (B = 1): the f images a lane adds are count / f apart, not adjacent -- out[a][b] = sum_i in[a][i count / f + b] --
                // so that consecutive lanes read consecutive images (adjacent images per lane put the lanes of a load f images apart:
                // every 16-byte access its own line, and over Fq2, where the coordinates are fetched one by one, lines were evicted
                // between their uses: 0.2 GB per proof re-read, r5_experiments.txt item 14).  The order of a sum of points does not change the sum.
                const bool wide = h.B == 1 && h.count > f;
                const uint32_t A = wide ? h.outer : h.outer * (h.count / f);
                job[q] = FoldJob{h.in, out, A, f, wide ? h.count / f : h.B, 0};
                if (h.count == f) h.done = true;
                h.count /= f; h.in = out; h.tog ^= 1;
            }
            const bool qf = quad || (quad_end && (size_t)job[0].A * job[0].B + (size_t)job[1].A * job[1].B <= QUAD_FOLD_JOBS);
            for (int q = 0; q < 2; ++q) job[q].blocks = job[q].in ? (uint32_t)ceil_div((size_t)job[q].A * job[q].B, qf ? QUAD_JOBS : TAIL_THREADS) : 0;
            if (qf && quad) hipLaunchKernelGGL((k_msm_fold_q<F, false>), dim3(job[0].blocks + job[1].blocks), dim3(QUAD_THREADS), 0, st, job[0], job[1]);
            else if (qf) hipLaunchKernelGGL((k_msm_fold_q<F, true>), dim3(job[0].blocks + job[1].blocks), dim3(QUAD_THREADS), 0, st, job[0], job[1]);
            else hipLaunchKernelGGL(k_msm_fold<F>, dim3(job[0].blocks + job[1].blocks), dim3(TAIL_THREADS), 0, st, job[0], job[1]);
        }
        if (quad || quad_end) {
            AccSlot<F>* d_term = reinterpret_cast<AccSlot<F>*>(ws.seg_sums.p);
            AccSlot<F>* d_qpart = d_term + (size_t)groups * (K + rows);
            const int terms = K + rows, parts = (terms + QUAD_SUM_THREADS - 1) / QUAD_SUM_THREADS;
            auto end = [&](auto big) {
                constexpr bool BIG = decltype(big)::value;
                hipLaunchKernelGGL((k_msm_weigh_q<F, BIG>), dim3(ceil_div(terms, QUAD_JOBS), groups), dim3(QUAD_THREADS), 0, st, d_C, d_R, kbits, rows, blo, d_term);
                if (parts > 1) {
                    hipLaunchKernelGGL((k_msm_sum_q<F, false, BIG>), dim3(parts, groups), dim3(QUAD_SUM_THREADS), 0, st, d_term, terms, d_qpart, d_out, grp.out_stride);
                    hipLaunchKernelGGL((k_msm_sum_q<F, true, BIG>), dim3(1, groups), dim3(QUAD_SUM_THREADS), 0, st, d_qpart, parts, d_qpart, d_out, grp.out_stride);
                } else {
                    hipLaunchKernelGGL((k_msm_sum_q<F, true, BIG>), dim3(1, groups), dim3(QUAD_SUM_THREADS), 0, st, d_term, terms, d_qpart, d_out, grp.out_stride);
                }
            };
            if (quad) end(std::false_type{}); else end(std::true_type{});
            ZK_HIP(hipGetLastError());
            return st;
        }
        hipLaunchKernelGGL(k_msm_weigh<F>, dim3(wgs_w, groups), dim3(TAIL_THREADS), 0, st, d_C, d_R, kbits, rows, blo, d_seg);
        // K + rows terms per group: one workgroup while each lane has at most 4 of them, otherwise two levels
        Jac<F>* d_part = d_seg + (size_t)groups * (K + rows);
        const int terms = K + rows, wgs2 = (terms + 1023) / 1024;
        if (wgs2 > 1) {
            hipLaunchKernelGGL(k_msm_sum_points<F>, dim3(wgs2, groups), dim3(256), 256 * sizeof(Jac<F>), st, d_seg, terms, d_part, (size_t)wgs2 * sizeof(Jac<F>));
            hipLaunchKernelGGL(k_msm_sum_points<F>, dim3(1, groups), dim3(256), 256 * sizeof(Jac<F>), st, d_part, wgs2, d_out, grp.out_stride);
        } else {
            hipLaunchKernelGGL(k_msm_sum_points<F>, dim3(1, groups), dim3(256), 256 * sizeof(Jac<F>), st, d_seg, terms, d_out, grp.out_stride);
        }
    }
    ZK_HIP(hipGetLastError());
    return st;
}
template hipStream_t msm_run<ZK_MSM_FIELD>(zk_ctx*, MsmWorkspace&, hipStream_t, const MsmTable<ZK_MSM_FIELD>&, const Fr*, size_t, int, int, Jac<ZK_MSM_FIELD>*, hipEvent_t, hipEvent_t, size_t, const MsmGroups&, const MsmSplit&);


#ifdef ZK_MSM_COMMON
// the staged level-2 scatter uses more than the default 64 KiB of dynamic LDS
void msm_init_attributes() {
    static bool done = false;
    if (done) return;
    ZK_HIP(hipFuncSetAttribute((const void*)k_msm_bin_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, BIN_STAGE * 6 + 2048 * 12));
    done = true;
}
#endif  // ZK_MSM_COMMON

#ifdef ZK_MSM_COMMON
#include "msm_lds.hpp"
#endif
// zk_msm_g1 with a negative window size: north_star's LDS-bucket form (msm_lds.hpp), G1 only
template <class F>
void msm_lds_dispatch(zk_ctx* ctx, hipStream_t st, const MsmTable<F>& tab, const Fr* d_scalars, size_t n, Jac<F>* d_out) {
#ifdef ZK_MSM_COMMON
    if constexpr (sizeof(F) == sizeof(Fq)) {
        DevBuf<Jac<Fq>> parts;
        if (!n) { ZK_HIP(hipMemsetAsync(d_out, 0, sizeof(Jac<Fq>), st)); return; }
        msm_lds_run_g1(ctx, st, tab, d_scalars, n, parts, d_out);
        ZK_HIP(hipStreamSynchronize(st));   // `parts` goes out of scope
        return;
    }
#endif
    throw StatusError{ZK_ERR_UNSUPPORTED, "zk_msm: the LDS-bucket form (negative window_bits) exists for G1 only"};
}

template <class F>
__global__ void k_jac_to_affine_canonical(const Jac<F>* in, Aff<F>* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pt_to_canonical(jac_to_affine(in[i]));
}

// zk_msm_g1 / zk_msm_g2: arbitrary (non-CRS) points, so the window table is built on the fly
template <class F>
void msm_host(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, int window_bits, uint64_t* out_affine) {
    ZK_REQUIRE(out_affine && (n == 0 || (points && scalars)), ZK_ERR_ARG, "zk_msm: null pointer");
    DevBuf<Aff<F>> dp(n), daff(1);
    DevBuf<Fr> ds(n), tmp(n);
    DevBuf<Jac<F>> dres(1);
    DevBuf<int> flag(1);
    hipStream_t st = ctx->stream;
    ZK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
    if (n) {
        ZK_HIP(hipMemcpyAsync(dp.p, points, n * sizeof(Aff<F>), hipMemcpyHostToDevice, st));
        ZK_HIP(hipMemcpyAsync(ds.p, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, st));
        pts_to_mont<Aff<F>>(ctx, dp.p, dp.p, n, flag.p);
        fr_to_mont(ctx, ds.p, tmp.p, n, flag.p);   // range check of the scalars (digits use the canonical integers)
    }
    int c = window_bits > 0 ? window_bits : (ctx->opt_window_bits > 0 ? (int)ctx->opt_window_bits : (sizeof(F) > sizeof(Fq) ? msm_auto_window_g2(n) : msm_auto_window(n)));
    if (window_bits < 0) c = -window_bits;   // the LDS-bucket comparator (msm_lds.hpp)
    MsmTable<F> tab;
    msm_build_table<F>(ctx, dp.p, n, c, tab);
    if (!ctx->msm_ws0) ctx->msm_ws0 = std::make_shared<MsmWorkspace>();
    if (window_bits < 0) msm_lds_dispatch<F>(ctx, st, tab, ds.p, n, dres.p);
    else msm_run<F>(ctx, *ctx->msm_ws0, st, tab, ds.p, n, 0, 1, dres.p);
    hipLaunchKernelGGL(k_jac_to_affine_canonical<F>, dim3(1), dim3(64), 0, st, dres.p, daff.p, 1);
    ZK_HIP(hipGetLastError());
    int hflag = 0;
    ZK_HIP(hipMemcpyAsync(&hflag, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipMemcpyAsync(out_affine, daff.p, sizeof(Aff<F>), hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_REQUIRE(!hflag, ZK_ERR_RANGE, "zk_msm: coordinate or scalar >= modulus");
}
template void msm_host<ZK_MSM_FIELD>(zk_ctx*, const uint64_t*, const uint64_t*, size_t, int, uint64_t*);

}  // namespace zk
