/* zkgpu_measure.h -- measurement and test entry points of libzkgpu.so, kept OUT of the product ABI (include/zkgpu.h).
 *
 * Nothing here replaces an interface of the reference (/root/reference has no profiling and no tuning knobs): these are the hooks of
 * bench.py (HIP-event timing of the library's own kernels for the `roofline` object), of tools/ (same-box A/B switches) and of
 * tests/test_lazy29.py (the lazy radix at its limb extremes).  The product library exports the three zk_profile_* functions and
 * zk_lazy29_batch; the tuning KEYS below are accepted only by a library built with -DZK_MEASURE (make -C zksnark_rs_amd/csrc measure;
 * zk_get_option(ctx, "measure_build") == 1) -- the product build answers ZK_ERR_UNSUPPORTED to them.
 *
 * Keys of zk_set_option in a ZK_MEASURE build (defaults are what the product build has compiled in):
 *   "profile"            0 off, 1 event-time the bucket accumulations, 2 every launch group (also accepted by the product build: bench.py)
 *   "serialize"          1: every kernel of a proof on one stream (stand-alone kernel durations)
 *   "ablate"             bit 0: reuse the previous sorted list of a workspace (repeated inputs only; prices the sort)
 *   "msm_fold"           images summed per lane and pass in the row / column sums of the MSM tail (4)
 *   "msm_run_entries"    longest run of the accumulation when buckets are cut into several runs (32)
 *   "msm_run_whole"      products with at most this many entries per bucket keep every bucket in ONE run (128)
 *   "msm_run_fill"       runs as long as one round of accumulation lanes allows (1)
 *   "msm_small_lanes", "msm_unchain_lanes"   thresholds of the small-product forms (65536, 140000)
 *   "chain_order"        order of a proof's accumulation chain (1 = A, B2, L + H)
 *   "alt_stream", "tail_stream"   the merged L + H product or its reduction tail on the idle L stream (0; profiles/r5_experiments.txt 8, 11)
 *   "ntt_fuse"           element-wise kernels folded into the DIF tile loads / stores (1)
 *   "apply_cu_reserve"   mask the inner-product streams now (tools/rccl_starvation.py)
 */
#ifndef ZKGPU_MEASURE_H
#define ZKGPU_MEASURE_H
#include "zkgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Profiling: HIP-event timing of the library's own kernels on the stream they run on.
 * ---------------------------------------------------------------------------------------- */
int zk_profile_reset(zk_ctx* ctx);
/* n_names = number of distinct kernels recorded; name(i) / stats(i) enumerate them */
int zk_profile_count(const zk_ctx* ctx);
int zk_profile_entry(const zk_ctx* ctx, int i, const char** name, double* total_ms, uint64_t* launches, double* algo_bytes);


/* The lazy radix-2^29 form of the same field arithmetic (what `Fr * Fr`, fr.rs:44-48, and bn's Fq multiply become inside
 * the bucket accumulation and the NTT tiles: csrc/lazy29.cuh, csrc/ntt.hip) on caller-supplied limb patterns, so that the
 * bounds the kernels rely on can be tested at their extremes.  Every operand is 9 signed 32-bit limbs per element, value =
 * sum v[k] 2^(29 k) (not reduced).  field: 0 = Fr, 1 = Fq.  out: canonical residues (4 words each) of the result;
 * raw_out (may be null): the result's 9 limbs before the final reduction.
 *   ZK_LAZY_MONT       a b 2^-261          (|a limbs| <= 2^30, |b limbs| < 2^29)
 *   ZK_LAZY_SQR        a a 2^-261          (|limbs| < 2^29)
 *   ZK_LAZY_MONT_DIFF  (a b - c d) 2^-261  (all |limbs| < 2^29)
 *   ZK_LAZY_NORM       a after carry propagation; ZK_LAZY_STORE  a itself        (|value| < 8 p)
 *   ZK_LAZY_FR_REDUCE  fr_reduce(a) of the NTT tiles (|value| < 2^9 r); ZK_LAZY_FR_STORE  fr_store_exact(a)   (Fr only) */
enum { ZK_LAZY_MONT = 0, ZK_LAZY_SQR = 1, ZK_LAZY_MONT_DIFF = 2, ZK_LAZY_NORM = 3, ZK_LAZY_STORE = 4, ZK_LAZY_FR_REDUCE = 5, ZK_LAZY_FR_STORE = 6 };
int zk_lazy29_batch(zk_ctx* ctx, int field, int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, size_t n,
                    uint64_t* out, int32_t* raw_out);

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_MEASURE_H */
