// common.hpp -- context, error plumbing, device buffers and HIP-event profiling shared by the
// translation units of libzkgpu.so.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/zkgpu.h"
#include "../../include/zkgpu_measure.h"   // the measurement / test entry points the library also exports (not part of the product ABI)
#include "ec.cuh"

namespace zk {

struct HipError {
    hipError_t code;
    std::string where;
};

#define ZK_HIP(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) throw ::zk::HipError{_e, std::string(#expr) + " @ " + __FILE__ + ":" + std::to_string(__LINE__)}; \
    } while (0)

struct StatusError {
    int status;
    std::string msg;
};
#define ZK_REQUIRE(cond, status, msg)                          \
    do {                                                       \
        if (!(cond)) throw ::zk::StatusError{(status), (msg)}; \
    } while (0)

// RAII device allocation
// hipFuncSetAttribute applies to the device that is current at the call, so "once" means once per DEVICE, not once per process: with a
// process-wide flag only the first device of a multi-context process got the larger dynamic LDS limit.  Thread-safe.
struct PerDeviceOnce {
    std::mutex mu;
    uint64_t done = 0;
    template <class Fn>
    void run(int device, Fn&& fn) {
        std::lock_guard<std::mutex> g(mu);
        const uint64_t bit = (uint64_t)1 << (device & 63);
        if (done & bit) return;
        fn();
        done |= bit;
    }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) ZK_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    }
    // grow-only: keeps the allocation when it is already large enough
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

struct ProfEntry {
    double total_ms = 0;
    uint64_t launches = 0;
    double algo_bytes = 0;
};
struct PendingEvent {
    std::string name;
    hipEvent_t e0, e1;
    double bytes;
    int slot;     // in-flight proof the launch belongs to (-1: none)
};

struct NttTables;  // ntt.hip
struct MsmWorkspace;  // kernels.hpp
struct ProveState;    // prove.hip

}  // namespace zk

struct zk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;    // side stream for the r/s-only fixed-base multiplications
    hipStream_t finish = nullptr;  // join + assembly + copy-out of a proof
    hipStream_t main_alt = nullptr;  // second main stream: odd-numbered proof slots run their SpMV / NTT stage here
    std::shared_ptr<zk::ProveState> prove_state;
    hipEvent_t submit_wait_evt = nullptr;   // consumed by the next prove_submit / prove_msm_submit: its first kernels wait for this event (comm.hip)
    int cur_slot = -1;
    std::string last_error;
    long opt_window_bits = 0;
    long opt_profile = 0;
    long opt_shard_points = 0;    // multi-GPU partial sums: 0 = by Pippenger windows, 1 = by point ranges, 2 = by bucket ranges (MsmGroups::bucket_shard)
    long opt_long_division = 0;   // dense form: always use the reference's long division (A/B check of the Newton form)
    long opt_rank_tables = 1;     // multi-GPU scalar exchange: window tables of this rank's point ranges only (prove_msm_submit)
    long opt_comm_cu_reserve = 0; // multi-GPU exchange over RCCL: compute units per XCD the inner-product streams leave free, so that the collectives' kernels never wait for an accumulation wave to retire (comm.hip, capi.hip ctx_reserve_cus); 0 = none
    int msm_cu_reserved = 0;      // what the inner-product streams are currently masked to leave free (per XCD)
    // tuning values: fixed in the product build, settable in a ZK_MEASURE build (capi.hip: option_slot)
    long opt_serialize = 0;       // 1: every kernel of a proof on one stream (stand-alone kernel timings)
    long opt_ablate = 0;          // bit 0: reuse the previous sorted list of a workspace (repeated inputs only; prices the sort)
    long opt_small_lanes = 65536; // inner products with fewer accumulation lanes than this take shorter runs (msm_impl.hpp)
    long opt_unchain_lanes = 140000; // inner products with fewer accumulation lanes than this are not chained behind the previous accumulation (2^16 gates: lone proof 2.83 -> 2.55 ms, two in flight level)
    long opt_alt_stream = 0;      // merged L + H product: odd tickets run it on the idle L stream (measurement switch; measured -1.4 %, profiles/r5_experiments.txt item 8)
    long opt_tail_stream = 0;     // merged L + H product: its reduction tail on the idle L stream (measurement switch)
    long opt_ntt_fuse = 1;        // roots-of-unity form, two-pass sizes: element-wise kernels folded into the DIF tile loads / stores (ntt_dif_fused); measurement switch
    long opt_merge_lh = 1;        // prove: L (witness over sum_delta) and H + r B1 + s A as ONE inner product over the table xi_t | xi | sum_delta (one bucket set, one tail); 0 = two products
    long opt_chain_order = 1;     // order of the accumulation chain of a proof: 0 = (L,) B2, A, HB; 1 = (L,) A, B2, HB; 2 = B2, (L,) HB, A.  Round 5, L merged into HB: 1 = 102.2 against 101.2 (0) and 100.8 (2) proofs/s, profiles/r5_experiments.txt item 5
    long opt_fold = 4;            // images summed per lane and pass in the row / column sums of the MSM tail
    long opt_run_entries = 32;    // longest run of the bucket accumulation when buckets are cut into several runs (multiple of 4)
    long opt_run_fill = 1;        // products cut into several runs per bucket: runs as long as one round of accumulation lanes allows (fewer merges)
    long opt_run_whole = 128;     // products with at most this many entries per bucket (and enough buckets) keep every bucket in ONE run
    long opt_basis_tree_min = 16384; // integer-roots QAP over a powers-only CRS: from this many gates on the Lagrange-basis points come from the transposed interpolation tree (gbasis.hip), below from the n^2 inner products (basis.hip)
    long opt_interp_large_log = 20; // interp.hip: trees of at least 2^this elements per level take the form that halves the upward transforms
    long opt_quad_buckets = 65536; // inner products of at most this many buckets run their reduction tail with four lanes per addition (msm_quad.hpp)
    std::map<std::string, zk::ProfEntry> prof;
    std::vector<zk::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<unsigned, std::shared_ptr<zk::NttTables>> ntt_tables;
    static constexpr int MSM_STREAMS = 5;        // slots: 0 = B in G2, 1 = L, 2 = A, 3 = unused, 4 = H + r B1 + s A
    std::shared_ptr<zk::MsmWorkspace> msm_ws0;   // workspace of the stand-alone zk_msm_* entry points
    hipStream_t msm_stream[MSM_STREAMS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int cu_count = 256;

    hipEvent_t get_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e;
        ZK_HIP(hipEventCreate(&e));
        return e;
    }
    // resolves the pending event pairs of one in-flight proof (-1: launches outside any proof; -2: all);
    // call after the work they bracket is known to be complete
    void resolve_profile(int slot = -1) {
        std::vector<zk::PendingEvent> keep;
        for (auto& pe : pending) {
            if (slot >= -1 && pe.slot != slot) { keep.push_back(pe); continue; }
            if (hipEventQuery(pe.e1) != hipSuccess) { keep.push_back(pe); continue; }   // still in flight (a slot released without waiting)
            float ms = 0;
            if (hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) {
                auto& e = prof[pe.name];
                e.total_ms += ms;
                e.launches += 1;
                e.algo_bytes += pe.bytes;
            }
            event_pool.push_back(pe.e0);
            event_pool.push_back(pe.e1);
        }
        pending.swap(keep);
    }
};

namespace zk {

// Brackets a kernel launch with HIP events on the launching stream when profiling is on.
struct ProfScope {
    zk_ctx* ctx;
    hipStream_t st;
    PendingEvent pe;
    bool on;
    // option profile: 0 off, 1 the bucket accumulations only (the dominant kernel; what bench.py's roofline needs -- event pairs
    // around all ~150 launches of a proof cost 1.3 % of the pipelined rate), 2 every launch group
    ProfScope(zk_ctx* c, const char* name, double algo_bytes, hipStream_t s = nullptr)
        : ctx(c), st(s ? s : c->stream), on(c->opt_profile >= 2 || (c->opt_profile == 1 && !std::strncmp(name, "msm_accumulate", 14))) {
        if (on) {
            pe.name = name;
            pe.bytes = algo_bytes;
            pe.slot = c->cur_slot;
            pe.e0 = ctx->get_event();
            pe.e1 = ctx->get_event();
            (void)hipEventRecord(pe.e0, st);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(pe.e1, st);
            ctx->pending.push_back(pe);
        }
    }
};

static inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Runs `fn`, mapping exceptions to zk_status codes; never lets anything cross the C ABI.
template <class Fn>
int guarded(zk_ctx* ctx, Fn&& fn) {
    try {
        if (ctx) ZK_HIP(hipSetDevice(ctx->device));
        (void)hipGetLastError();   // a call starts clean: the thread's sticky error may stem from another context's tear-down or a caller's own HIP use
        fn();
        return ZK_OK;
    } catch (const HipError& e) {
        if (ctx) ctx->last_error = std::string(hipGetErrorString(e.code)) + " in " + e.where;
        return ZK_ERR_HIP;
    } catch (const StatusError& e) {
        if (ctx) ctx->last_error = e.msg;
        return e.status;
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->last_error = "host allocation failed";
        return ZK_ERR_HIP;
    } catch (const std::exception& e) {
        if (ctx) ctx->last_error = e.what();
        return ZK_ERR_ARG;
    } catch (...) {
        if (ctx) ctx->last_error = "unknown error";
        return ZK_ERR_ARG;
    }
}

}  // namespace zk
