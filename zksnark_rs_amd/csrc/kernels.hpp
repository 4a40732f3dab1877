// kernels.hpp -- internal interfaces between the translation units of libzkgpu.so.
#pragma once
#include "common.hpp"

namespace zk {

// ---- field_kernels.hip ----
template <class F> void field_batch(zk_ctx*, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
void fr_to_mont(zk_ctx*, const Fr* in, Fr* out, size_t n, int* d_flag);
void fr_from_mont(zk_ctx*, const Fr* in, Fr* out, size_t n);
void fr_check_range(zk_ctx*, const Fr* in, size_t n, int* d_flag);   // *d_flag |= 2 when an element is >= r
template <class A> void pts_to_mont(zk_ctx*, const A* in, A* out, size_t n, int* d_flag);
template <class A> void pts_from_mont(zk_ctx*, const A* in, A* out, size_t n);
// *d_flag |= 4 when a finite point (Montgomery form) is not on its curve (G1: y^2 = x^3 + 3; G2: the twist)
template <class A> void pts_check_on_curve(zk_ctx*, const A* d_pts, size_t n, int* d_flag);
template <class F> void point_mul_batch(zk_ctx*, const uint64_t* points, const uint64_t* scalars, uint64_t* out, size_t n);
template <class F> void point_add_batch(zk_ctx*, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);

// ---- ntt.hip ----
// Host-side field constants (computed with the same ff.cuh code on the host)
Fr host_root_of_unity(unsigned log_n);   // w = 5^((r-1)/2^log_n), Montgomery form
Fr host_fr_from_u64(uint64_t v);
Fr host_fr_pow(Fr base, uint64_t e);

constexpr unsigned NTT_MAX_LOCAL_LOG = 11;   // 2048-element LDS tiles (64 KiB)
constexpr unsigned NTT_MAX_LOG = 24;         // <= 22: at most two passes over HBM; 23, 24: one more column pass in front (ntt.hip)

struct NttTables {
    unsigned log_n = 0;
    DevBuf<Fr> tw_fwd, tw_inv;       // w_2048^k and w_2048^-k, k < 1024 (local butterflies)
    DevBuf<int32_t> tw29_fwd, tw29_inv;   // the same in the tiles' lazy radix, 12 words per twiddle (ntt.hip k_tw29)
    DevBuf<Fr> mid_fwd, mid_inv;     // inter-pass twiddles, n entries each (log_n > 11)
    DevBuf<Fr> coset_fwd_brev;       // g^brev(pos) / n                 (prove pipeline, DIT input order; n: of the unscaled inverse transform before it)
    DevBuf<Fr> coset_inv_brev_half;  // g^-brev(pos) / (2 n) as PLAIN integers (prove pipeline, h combine: the product with a Montgomery-form value is canonical)
    Fr n_inv;                        // 1/n
};
std::shared_ptr<NttTables> ntt_get_tables(zk_ctx*, unsigned log_n);
void ntt_ensure_coset_tables(zk_ctx*, NttTables&);

// natural order in -> bit-reversed order out.  inverse: use w^-1; scale_n_inv: multiply by 1/n.
// batch: d_data holds `batch` transforms of 2^log_n elements back to back, transformed in the same launches
void ntt_dif(zk_ctx*, Fr* d_data, unsigned log_n, bool inverse, bool scale_n_inv, size_t batch = 1);
// bit-reversed order in -> natural order out.  d_pre (optional): element-wise multiplier applied
// to the input (in its bit-reversed order) as it is loaded.
void ntt_dit(zk_ctx*, Fr* d_data, unsigned log_n, bool inverse, bool scale_n_inv, const Fr* d_pre, size_t batch = 1);
void ntt_dif_pre(zk_ctx*, Fr* d_data, unsigned log_n, const Fr* table, size_t step, size_t batch);   // element i times table[i * step] on the first load
// the prove pipeline's unscaled DIF transforms with the element-wise kernels around them folded in (ntt.hip ntt_dif_fused)
struct NttFuse {
    const Fr* src_a[2] = {nullptr, nullptr};   // first-pass sources per half of the batch (null: the output array itself)
    const Fr* src_b[2] = {nullptr, nullptr};   // ... multiplied element-wise by these
    size_t half = 0;                           // transforms [0, half) use set 0, the rest set 1
    Fr* canon_out[2] = {nullptr, nullptr};     // last pass: canonical(value * canon_k) per half
    Fr canon_k;                                // Montgomery form
};
bool ntt_dif_fusable(unsigned log_n);
void ntt_dif_fused(zk_ctx*, Fr* d_out, unsigned log_n, bool inverse, size_t batch, const NttFuse&);
void bitrev_permute(zk_ctx*, const Fr* d_in, Fr* d_out, unsigned log_n);
// out[i] = a[i] * b[i]
void fr_pointwise_mul(zk_ctx*, const Fr* a, const Fr* b, Fr* out, size_t n);
// out[i] = base^i * scale (natural order powers)
void fr_powers(zk_ctx*, Fr base, Fr scale, Fr* out, size_t n);
void ntt_host(zk_ctx*, uint64_t* data, unsigned log_n, int inverse, int coset);
// qap.hip: NTT image (DIF order, size 2^log_size) of 1 / rev(t) mod x^K, t of degree d (d_cinv: 1 / its leading coefficient); synchronises
void poly_rev_inverse_ntt(zk_ctx*, const Fr* t, size_t d, const Fr* d_cinv, size_t K, unsigned log_size, DevBuf<Fr>& out);
void lazy29_batch(zk_ctx*, int field, int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, size_t n, uint64_t* out, int32_t* raw_out);

// ---- msm.hip ----
constexpr int MSM_MAX_C = 22;   // window bits; the two-level sort keeps only 2^8 + 2^(c-9) counters in LDS

// T[w][i] = 2^(c w) P_i, w < windows, i < n (affine, Montgomery)
template <class F>
struct MsmTable {
    DevBuf<Aff<F>> table;
    size_t n = 0;
    int c = 0, windows = 0;
};
struct MsmWorkspace {
    DevBuf<uint32_t> hist, total, bin_start, part_start, bin_cnt, start, sorted, heavy;
    DevBuf<uint64_t> records;
    DevBuf<uint32_t> runs_cnt, wg_extra, xbase, runs;   // runs of the accumulation (msm_impl.hpp): class counts | cursors | info, ..., 3 words per run
    DevBuf<uint8_t> bucket_sums, fold, seg_sums;
    uint64_t sorted_for = 0;             // signature of the sorted list held (option ablate, ZK_MEASURE builds)
};
int msm_auto_window(size_t n);
int msm_auto_window_g2(size_t n);
void msm_init_attributes();
template <class F>
void msm_build_table(zk_ctx*, const Aff<F>* d_points, size_t n, int c, MsmTable<F>& out);
// sum_{i < n_used} scalars[i] * P_i over the table's bases; scalars are CANONICAL Fr limbs.
// Only windows w = rank (mod world) are accumulated (multi-GPU partial sums by windows); `point_offset`
// selects the bases [point_offset, point_offset + n_used) of the table instead (partial sums by point ranges).  Everything is
// enqueued on `st`; the result (Jacobian, Montgomery) lands in d_out.
// `acc_wait` (may be null): event the bucket accumulation waits for; `acc_done` (may be null): event
// recorded right after it.  The accumulation kernels each fill every SIMD, so the pipeline chains
// them in a chosen order instead of letting them thrash each other's caches.
// Grouped form: `groups` independent products over the SAME bases in one pass (the proofs of one scalar-exchange
// round): the scalars of group j are d_scalars[j glen .. j glen + valid), its result goes to (bytes) d_out + j out_stride;
// n_used is ignored.  Each group has its own 2^(c-1) buckets in one sorted list, so the sort, the accumulation and the
// reduction tails run once per round instead of once per proof.
struct MsmGroups {
    int groups = 1;
    size_t glen = 0, valid = 0, out_stride = 0;
    hipStream_t tail_stream = nullptr;   // when set (and acc_done given): the reduction tail runs there, behind the accumulation's event
    // Partial sum of a product shared by `world` ranks BY BUCKET RANGE (world a power of two): rank g keeps the digits whose bucket
    // |digit| - 1 lies in [g 2^(c-1) / world, (g + 1) 2^(c-1) / world) -- of every window and every point -- so entries, accumulation
    // AND the per-bucket reduction tail are 1 / world of the product's.  Otherwise (rank, world) of msm_run mean Pippenger windows.
    bool bucket_shard = false;
};
// Two scalar arrays, one product (groups == 1 only): the scalars of the points [0, split) come from d_scalars, those of the points
// [split, split + n2) from scalars2 -- n_used is then split + n2.  How prove() multiplies its witness (the caller's array) and its
// quotient / r v + s u scalars (the slot's) over ONE table sum_delta-behind-xi_t-and-xi with one set of buckets: both sums only ever
// occur added together in the proof element c (prove.hip).
// Grouped form (the proofs of a scalar-exchange round): group j reads its first `split` scalars from d_scalars + j stride1 and the
// rest from scalars2 + j stride2; glen = split + (length of the second part), and the two parts have their own counts of valid scalars
// (valid1 <= split, n2 <= glen - split: what the rank's point ranges hold of each).
struct MsmSplit {
    const Fr* scalars2 = nullptr;
    size_t split = ~(size_t)0, n2 = 0;
    size_t valid1 = ~(size_t)0;        // valid scalars of the first part (default: all `split` of them)
    size_t stride1 = 0, stride2 = 0;   // grouped form only
};
// Returns the stream the result lands on (st).
template <class F>
hipStream_t msm_run(zk_ctx*, MsmWorkspace& ws, hipStream_t st, const MsmTable<F>& tab, const Fr* d_scalars, size_t n_used,
             int rank, int world, Jac<F>* d_out, hipEvent_t acc_wait = nullptr, hipEvent_t acc_done = nullptr, size_t point_offset = 0,
             const MsmGroups& grp = MsmGroups(), const MsmSplit& sp = MsmSplit());
template <class F>
void msm_host(zk_ctx*, const uint64_t* points, const uint64_t* scalars, size_t n, int window_bits, uint64_t* out_affine);

}  // namespace zk
