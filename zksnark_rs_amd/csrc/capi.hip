// capi.hip -- the extern "C" surface declared in include/zkgpu.h.
// Nothing here computes on the CPU: every entry point either launches HIP kernels or fails
// with a status code (ZK_ERR_NO_DEVICE when no GPU is visible).
#include <cstring>
#include <vector>
#include "kernels.hpp"
#include "pipeline.hpp"
#include "interp.hpp"

using namespace zk;

namespace zk {
// Re-creates the inner-product streams so that their kernels leave `per_xcd` compute units of every XCD alone (0 = the plain
// mid-priority streams of zk_ctx_create).  Used by the multi-GPU exchange (zk_mgpu_create, option comm_cu_reserve): RCCL's send /
// receive kernels run on the collectives' stream, which is not masked, and always find those units free instead of waiting ~2 ms
// (p90 5 ms) for an accumulation wave to retire (profiles/r3_rccl_starvation.txt, r4_rccl_starvation.txt).  Mask layout measured
// with tools/ubench_cumask.hip: clearing bits 0 .. 8 R - 1 takes R units from each of the 8 XCDs -- on the part it was measured on
// (256 compute units in 8 XCDs); on anything else the option is refused rather than applied to a layout nobody measured.
// The masked streams come from hipExtStreamCreateWithCUMask, which takes neither flags nor a priority: they are BLOCKING streams of
// the default priority, i.e. they synchronise implicitly with the legacy NULL stream (a caller that uses it -- torch does -- gets its
// NULL-stream work serialised against the inner products) and they no longer sit between the main streams and the tails in
// priority.  Default off (profiles/r4_rccl_starvation.txt); the plain mid-priority non-blocking streams come back with per_xcd = 0.
void ctx_reserve_cus(zk_ctx* ctx, int per_xcd) {
    per_xcd = std::max(0, std::min(per_xcd, 8));
    if (per_xcd == ctx->msm_cu_reserved) return;
    ZK_REQUIRE(per_xcd == 0 || ctx->cu_count == 256, ZK_ERR_UNSUPPORTED, "comm_cu_reserve: the compute-unit mask layout is only known for 256 units in 8 XCDs");
    ZK_HIP(hipSetDevice(ctx->device));
    ZK_HIP(hipDeviceSynchronize());
    int prio_least = 0, prio_greatest = 0;
    ZK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const int prio_mid = (prio_least + prio_greatest) / 2;
    const uint32_t words = (uint32_t)((ctx->cu_count + 31) / 32);
    std::vector<uint32_t> mask(words, 0xffffffffu);
    for (int b = 0; b < 8 * per_xcd && b < ctx->cu_count; ++b) mask[b / 32] &= ~(1u << (b % 32));
    for (int i = 0; i < zk_ctx::MSM_STREAMS; ++i) {
        if (i == 3) continue;
        hipStream_t fresh = nullptr;
        if (per_xcd) ZK_HIP(hipExtStreamCreateWithCUMask(&fresh, words, mask.data()));
        else ZK_HIP(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio_mid));
        if (ctx->msm_stream[i]) (void)hipStreamDestroy(ctx->msm_stream[i]);
        ctx->msm_stream[i] = fresh;
    }
    ctx->msm_cu_reserved = per_xcd;
}
}  // namespace zk

extern "C" {

const char* zk_strerror(int status) {
    switch (status) {
        case ZK_OK: return "ok";
        case ZK_ERR_ARG: return "invalid argument";
        case ZK_ERR_HIP: return "HIP runtime error";
        case ZK_ERR_NO_DEVICE: return "no HIP device (the product path has no CPU fallback)";
        case ZK_ERR_SIZE: return "size out of supported range";
        case ZK_ERR_DIV_BY_ZERO: return "division by zero";
        case ZK_ERR_RANGE: return "field element out of range";
        case ZK_ERR_UNSUPPORTED: return "unsupported";
        case ZK_ERR_IO: return "file missing, truncated, altered or in the wrong format";
        case ZK_ERR_COMM: return "collective failed";
        default: return "unknown status";
    }
}

void zk_ctx_destroy(zk_ctx* ctx);

int zk_ctx_create(int device_ordinal, zk_ctx** out) {
    if (!out) return ZK_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZK_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= count) return ZK_ERR_ARG;
    zk_ctx* ctx = new (std::nothrow) zk_ctx();
    if (!ctx) return ZK_ERR_HIP;
    ctx->device = device_ordinal;
    int rc = guarded(ctx, [&] {
        hipDeviceProp_t prop;
        ZK_HIP(hipGetDeviceProperties(&prop, device_ordinal));
        ctx->cu_count = prop.multiProcessorCount;
        int prio_least = 0, prio_greatest = 0;
        ZK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        // the main stream carries the NTT stage (many short dependent kernels): high priority so that it
        // is not starved by the accumulation kernels of the MSM streams
        ZK_HIP(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_greatest));
        // A and B1 (streams 2, 3) and the side stream that consumes them run at high priority so the
        // dynamic-base multiplications s*A, r*B1 start early and hide behind the other inner products
        ZK_HIP(hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, prio_greatest));
        ZK_HIP(hipStreamCreateWithPriority(&ctx->finish, hipStreamNonBlocking, prio_greatest));
        ZK_HIP(hipStreamCreateWithPriority(&ctx->main_alt, hipStreamNonBlocking, prio_greatest));
        msm_init_attributes();
        // Streams share hardware queues per priority level (4 each by default): a FIFTH stream of a level shares a queue with another
        // one and serialises with it (measured: reduction tails on two extra streams -27 %, the G2 product alternating onto a fifth
        // MSM stream -5 %, every accumulation on one extra stream -- low priority, or masked to all but 8 / 16 / 32 compute units --
        // -2.5 .. -5 %: DESIGN.md 4c).  Two levels are used:
        //   high  main / alternate main (SpMV + NTT stage), side (r, s multiples), finish (join, assembly, copy-out)
        //   mid   one stream per inner product: its sort, its accumulation and its reduction tail
        const int prio_mid = (prio_least + prio_greatest) / 2;
        for (int i = 0; i < zk_ctx::MSM_STREAMS; ++i) {
            if (i == 3) continue;   // slot 3 (B in G1) was folded into the H product
            ZK_HIP(hipStreamCreateWithPriority(&ctx->msm_stream[i], hipStreamNonBlocking, prio_mid));
        }
    });
    if (rc != ZK_OK) { zk_ctx_destroy(ctx); return rc; }   // destroys whatever streams were created before the failure
    *out = ctx;
    return ZK_OK;
}

void zk_ctx_destroy(zk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->ntt_tables.clear();
    (void)hipDeviceSynchronize();
    ctx->prove_state.reset();
    ctx->msm_ws0.reset();
    for (int i = 0; i < zk_ctx::MSM_STREAMS; ++i)
        if (ctx->msm_stream[i]) (void)hipStreamDestroy(ctx->msm_stream[i]);
    if (ctx->finish) (void)hipStreamDestroy(ctx->finish);
    if (ctx->main_alt) (void)hipStreamDestroy(ctx->main_alt);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto& pe : ctx->pending) { (void)hipEventDestroy(pe.e0); (void)hipEventDestroy(pe.e1); }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    (void)hipGetLastError();   // whatever the tear-down left in the thread's sticky error must not surface in another context's next call
    delete ctx;
}

const char* zk_last_error(const zk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

static long* option_slot(zk_ctx* ctx, const char* key) {
    if (!std::strcmp(key, "msm_window_bits")) return &ctx->opt_window_bits;
    if (!std::strcmp(key, "profile")) return &ctx->opt_profile;
    if (!std::strcmp(key, "rank_tables")) return &ctx->opt_rank_tables;
    if (!std::strcmp(key, "dense_long_division")) return &ctx->opt_long_division;
    if (!std::strcmp(key, "msm_shard_points")) return &ctx->opt_shard_points;
    if (!std::strcmp(key, "msm_quad_buckets")) return &ctx->opt_quad_buckets;
    if (!std::strcmp(key, "interp_large_log")) return &ctx->opt_interp_large_log;
    if (!std::strcmp(key, "comm_cu_reserve")) return &ctx->opt_comm_cu_reserve;
    if (!std::strcmp(key, "basis_tree_min")) return &ctx->opt_basis_tree_min;
    if (!std::strcmp(key, "merge_lh")) return &ctx->opt_merge_lh;
#ifdef ZK_MEASURE
    // measurement switches (tools/ab_*.sh, bench.py --opt / --serialize): not part of the product build
    if (!std::strcmp(key, "serialize")) return &ctx->opt_serialize;
    if (!std::strcmp(key, "ablate")) return &ctx->opt_ablate;
    if (!std::strcmp(key, "msm_fold")) return &ctx->opt_fold;
    if (!std::strcmp(key, "msm_small_lanes")) return &ctx->opt_small_lanes;
    if (!std::strcmp(key, "msm_unchain_lanes")) return &ctx->opt_unchain_lanes;
    if (!std::strcmp(key, "chain_order")) return &ctx->opt_chain_order;
    if (!std::strcmp(key, "msm_run_entries")) return &ctx->opt_run_entries;
    if (!std::strcmp(key, "msm_run_whole")) return &ctx->opt_run_whole;
    if (!std::strcmp(key, "msm_run_fill")) return &ctx->opt_run_fill;
    if (!std::strcmp(key, "alt_stream")) return &ctx->opt_alt_stream;
    if (!std::strcmp(key, "ntt_fuse")) return &ctx->opt_ntt_fuse;
    if (!std::strcmp(key, "tail_stream")) return &ctx->opt_tail_stream;
#endif
    return nullptr;
}
int zk_set_option(zk_ctx* ctx, const char* key, long value) {
    if (!ctx || !key) return ZK_ERR_ARG;
#ifdef ZK_MEASURE
    // measurement build: mask the inner-product streams NOW (what zk_mgpu_create does over a multi-rank RCCL communicator), so that
    // tools/rccl_starvation.py can price the reservation on one GPU
    if (!std::strcmp(key, "apply_cu_reserve")) return guarded(ctx, [&] { zk::ctx_reserve_cus(ctx, (int)value); });
#endif
    long* s = option_slot(ctx, key);
    if (!s) return ZK_ERR_UNSUPPORTED;
    *s = value;
    return ZK_OK;
}
long zk_get_option(const zk_ctx* ctx, const char* key) {
    if (!ctx || !key) return -1;
    if (!std::strcmp(key, "measure_build")) {   // 1: the library was compiled with ZK_MEASURE and accepts the measurement switches
#ifdef ZK_MEASURE
        return 1;
#else
        return 0;
#endif
    }
    long* s = option_slot(const_cast<zk_ctx*>(ctx), key);
    return s ? *s : -1;
}

int zk_ntt_fr(zk_ctx* ctx, uint64_t* data, unsigned log_n, int inverse, int coset) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { ntt_host(ctx, data, log_n, inverse, coset); ctx->resolve_profile(); });
}
int zk_interpolate_fr(zk_ctx* ctx, const uint64_t* roots, const uint64_t* values, size_t n, uint64_t* coeffs) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { interp_host(ctx, roots, values, n, coeffs); ctx->resolve_profile(); });
}
int zk_msm_g1(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, int window_bits, uint64_t out[ZK_G1_WORDS]) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { msm_host<Fq>(ctx, points, scalars, n, window_bits, out); ctx->resolve_profile(); });
}
int zk_msm_g2(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, size_t n, int window_bits, uint64_t out[ZK_G2_WORDS]) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { msm_host<Fq2>(ctx, points, scalars, n, window_bits, out); ctx->resolve_profile(); });
}
int zk_lazy29_batch(zk_ctx* ctx, int field, int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, size_t n, uint64_t* out, int32_t* raw_out) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { lazy29_batch(ctx, field, op, a, b, c, d, n, out, raw_out); });
}
int zk_fr_batch(zk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { field_batch<Fr>(ctx, op, a, b, out, n); });
}
int zk_fq_batch(zk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { field_batch<Fq>(ctx, op, a, b, out, n); });
}
int zk_g1_mul_batch(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t* out, size_t n) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { point_mul_batch<Fq>(ctx, points, scalars, out, n); });
}
int zk_g2_mul_batch(zk_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t* out, size_t n) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { point_mul_batch<Fq2>(ctx, points, scalars, out, n); });
}
int zk_g1_add_batch(zk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { point_add_batch<Fq>(ctx, a, b, out, n); });
}
int zk_g2_add_batch(zk_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { point_add_batch<Fq2>(ctx, a, b, out, n); });
}

// ---- QAP / CRS / prove: see pipeline.hip ---------------------------------------------------
int zk_qap_upload_sparse(zk_ctx* ctx, const zk_qap_sparse_desc* desc, zk_qap** out) {
    if (!ctx || !desc || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = qap_upload_sparse(ctx, *desc); });
}
/* the window size msm tables of `count` points are built with when the option msm_window_bits is 0 (host code) */
int zk_msm_auto_window(size_t count) { return msm_auto_window(count); }
int zk_msm_auto_window_g2(size_t count) { return msm_auto_window_g2(count); }
int zk_qap_upload_sparse_integers(zk_ctx* ctx, const zk_qap_sparse_desc* desc, size_t n, zk_qap** out) {
    if (!ctx || !desc || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = qap_upload_sparse_integers(ctx, *desc, n); });
}
int zk_qap_upload_sparse_roots(zk_ctx* ctx, const zk_qap_sparse_desc* desc, const uint64_t* roots, size_t n, zk_qap** out) {
    if (!ctx || !desc || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = qap_upload_sparse_roots(ctx, *desc, roots, n); });
}
int zk_qap_upload_dense(zk_ctx* ctx, const uint64_t* u, const uint64_t* v, const uint64_t* w, const uint64_t* t,
                        size_t m, size_t n, size_t input, zk_qap** out) {
    if (!ctx || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = qap_upload_dense(ctx, u, v, w, t, m, n, input); });
}
void zk_qap_free(zk_qap* qap) { qap_free(qap); }

int zk_crs_upload(zk_ctx* ctx, const zk_crs_desc* desc, zk_crs** out) {
    if (!ctx || !desc || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = crs_upload(ctx, *desc); });
}
int zk_setup(zk_ctx* ctx, const zk_qap* qap, const uint64_t trapdoor[20], zk_crs** out) {
    if (!ctx || !qap || !trapdoor || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = crs_setup(ctx, *qap, trapdoor); });
}
int zk_crs_dims(const zk_crs* crs, size_t* n, size_t* m, size_t* input) {
    if (!crs) return ZK_ERR_ARG;
    crs_dims(*crs, n, m, input);
    return ZK_OK;
}
int zk_qap_save(zk_ctx* ctx, const zk_qap* qap, const char* path) {
    if (!ctx || !qap || !path) return ZK_ERR_ARG;
    return guarded(ctx, [&] { qap_save(ctx, *qap, path); });
}
int zk_qap_load(zk_ctx* ctx, const char* path, zk_qap** out) {
    if (!ctx || !path || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = qap_load(ctx, path); });
}
int zk_proof_save(const uint8_t proof[ZK_PROOF_BYTES], const char* path) {
    if (!proof || !path) return ZK_ERR_ARG;
    return guarded(nullptr, [&] { proof_save(proof, path); });
}
int zk_proof_load(const char* path, uint8_t proof_out[ZK_PROOF_BYTES]) {
    if (!proof_out || !path) return ZK_ERR_ARG;
    return guarded(nullptr, [&] { proof_load(path, proof_out); });
}
int zk_crs_save(zk_ctx* ctx, const zk_crs* crs, const char* path) {
    if (!ctx || !crs || !path) return ZK_ERR_ARG;
    return guarded(ctx, [&] { crs_save(ctx, *crs, path); });
}
int zk_crs_load(zk_ctx* ctx, const char* path, zk_crs** out) {
    if (!ctx || !path || !out) return ZK_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] { *out = crs_load(ctx, path); });
}
int zk_crs_download(zk_ctx* ctx, const zk_crs* crs, const zk_crs_out* out) {
    if (!ctx || !crs || !out) return ZK_ERR_ARG;
    return guarded(ctx, [&] { crs_download(ctx, *crs, *out); });
}
void zk_crs_free(zk_crs* crs) { crs_free(crs); }

int zk_prove(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const uint64_t* weights, size_t m,
             const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]) {
    if (!ctx || !crs || !qap || !weights || !r || !s || !proof_out) return ZK_ERR_ARG;
    return guarded(ctx, [&] { prove_host(ctx, *crs, *qap, weights, m, r, s, proof_out); ctx->resolve_profile(); });
}
int zk_prove_dev(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                 const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]) {
    if (!ctx || !crs || !qap || !d_weights || !r || !s || !proof_out) return ZK_ERR_ARG;
    return guarded(ctx, [&] { prove_dev(ctx, *crs, *qap, (const Fr*)d_weights, m, r, s, proof_out, 0, 1, nullptr); ctx->resolve_profile(); });
}
int zk_prove_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                    const uint64_t r[4], const uint64_t s[4], int* ticket) {
    if (!ctx || !crs || !qap || !d_weights || !r || !s || !ticket) return ZK_ERR_ARG;
    return guarded(ctx, [&] { *ticket = prove_submit(ctx, *crs, *qap, (const Fr*)d_weights, m, r, s, 0, 1, nullptr); });
}
int zk_prove_submit_host(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const uint64_t* weights, size_t m,
                         const uint64_t r[4], const uint64_t s[4], int* ticket) {
    if (!ctx || !crs || !qap || (!weights && m) || !r || !s || !ticket) return ZK_ERR_ARG;
    return guarded(ctx, [&] { *ticket = prove_submit_host(ctx, *crs, *qap, weights, m, r, s); });
}
int zk_host_alloc(size_t bytes, void** out) {
    if (!out) return ZK_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? ZK_OK : ZK_ERR_HIP;
}
void zk_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}
int zk_prove_partial_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                            const uint64_t r[4], const uint64_t s[4], int rank, int world, void* d_partial_out, int* ticket) {
    if (!ctx || !crs || !qap || !d_weights || !r || !s || !d_partial_out || !ticket || world < 1 || rank < 0 || rank >= world) return ZK_ERR_ARG;
    return guarded(ctx, [&] { *ticket = prove_submit(ctx, *crs, *qap, (const Fr*)d_weights, m, r, s, rank, world, d_partial_out); });
}
int zk_prove_batch_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, int count, const void* const* d_weights, const size_t* m,
                          const uint64_t* r, const uint64_t* s, int* ticket) {
    if (!ctx || !crs || !qap || !d_weights || !m || !r || !s || !ticket) return ZK_ERR_ARG;
    return guarded(ctx, [&] { *ticket = prove_batch_submit(ctx, *crs, *qap, count, d_weights, m, r, s); });
}
int zk_prove_batch_wait(zk_ctx* ctx, int ticket, int count, uint8_t* proofs_out) {
    if (!ctx || !proofs_out) return ZK_ERR_ARG;
    return guarded(ctx, [&] { prove_batch_wait(ctx, ticket, count, proofs_out); });
}
int zk_prove_exchange_elems(const zk_qap* qap, int world, size_t elems_out[4]) {
    if (!qap || world < 1 || !elems_out) return ZK_ERR_ARG;
    if (qap->dense) return ZK_ERR_UNSUPPORTED;
    prove_exchange_elems(*qap, world, elems_out);
    return ZK_OK;
}
int zk_prove_scalars_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                            const uint64_t r[4], const uint64_t s[4], int world,
                            void* d_l, void* d_v, void* d_u, void* d_h, int* ticket) {
    if (!ctx || !crs || !qap || !d_weights || !r || !s || world < 1 || !d_l || !d_v || !d_u || !d_h || !ticket) return ZK_ERR_ARG;
    return guarded(ctx, [&] {
        Fr* xout[4] = {(Fr*)d_l, (Fr*)d_v, (Fr*)d_u, (Fr*)d_h};
        *ticket = prove_submit(ctx, *crs, *qap, (const Fr*)d_weights, m, r, s, 0, world, nullptr, xout);
    });
}
int zk_prove_scalars_submit_host(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const uint64_t* weights, size_t m,
                                 const uint64_t r[4], const uint64_t s[4], int world,
                                 void* d_l, void* d_v, void* d_u, void* d_h, int* ticket) {
    if (!ctx || !crs || !qap || !weights || !r || !s || world < 1 || !d_l || !d_v || !d_u || !d_h || !ticket) return ZK_ERR_ARG;
    return guarded(ctx, [&] {
        Fr* xout[4] = {(Fr*)d_l, (Fr*)d_v, (Fr*)d_u, (Fr*)d_h};
        *ticket = prove_submit_host(ctx, *crs, *qap, weights, m, r, s, world, xout);
    });
}
int zk_prove_msm_submit(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, int sets, int rank, int world,
                        const void* d_l, const void* d_v, const void* d_u, const void* d_h, void* d_partials_out, int* ticket) {
    if (!ctx || !crs || !qap || sets < 0 || world < 1 || rank < 0 || rank >= world || !d_l || !d_v || !d_u || !d_h || !d_partials_out || !ticket)
        return ZK_ERR_ARG;
    return guarded(ctx, [&] {
        *ticket = prove_msm_submit(ctx, *crs, *qap, sets, rank, world, (const Fr*)d_l, (const Fr*)d_v, (const Fr*)d_u, (const Fr*)d_h, d_partials_out);
    });
}
int zk_prove_wait(zk_ctx* ctx, int ticket, uint8_t* proof_out) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { prove_wait(ctx, ticket, proof_out); });
}
int zk_prove_partial(zk_ctx* ctx, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                     const uint64_t r[4], const uint64_t s[4], int rank, int world, void* d_partial_out) {
    if (!ctx || !crs || !qap || !d_weights || !r || !s || !d_partial_out || world < 1 || rank < 0 || rank >= world) return ZK_ERR_ARG;
    return guarded(ctx, [&] { prove_dev(ctx, *crs, *qap, (const Fr*)d_weights, m, r, s, nullptr, rank, world, d_partial_out); ctx->resolve_profile(); });
}
int zk_prove_combine(zk_ctx* ctx, const zk_crs* crs, const void* d_partials, int world,
                     const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]) {
    if (!ctx || !crs || !d_partials || world < 1 || !r || !s || !proof_out) return ZK_ERR_ARG;
    return guarded(ctx, [&] { prove_combine(ctx, *crs, d_partials, world, r, s, proof_out); ctx->resolve_profile(); });
}

int zk_profile_reset(zk_ctx* ctx) {
    if (!ctx) return ZK_ERR_ARG;
    return guarded(ctx, [&] { ZK_HIP(hipDeviceSynchronize()); ctx->resolve_profile(-2); ctx->prof.clear(); });
}
int zk_profile_count(const zk_ctx* ctx) { return ctx ? (int)ctx->prof.size() : 0; }
int zk_profile_entry(const zk_ctx* ctx, int i, const char** name, double* total_ms, uint64_t* launches, double* algo_bytes) {
    if (!ctx || i < 0 || i >= (int)ctx->prof.size()) return ZK_ERR_ARG;
    auto it = ctx->prof.begin();
    std::advance(it, i);
    if (name) *name = it->first.c_str();
    if (total_ms) *total_ms = it->second.total_ms;
    if (launches) *launches = it->second.launches;
    if (algo_bytes) *algo_bytes = it->second.algo_bytes;
    return ZK_OK;
}

}  // extern "C"
