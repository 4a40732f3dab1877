// comm.hip -- several GPUs behind the C ABI (SURVEY.md 8b "one context owns 1..8 devices", 8e).
//
// The reference's groth16::prove (/root/reference/src/groth16/mod.rs:213-217) is one call on one thread; its inner
// products (mod.rs:255-272, 279-290) are sums of independent terms, so they shard.  One process per GPU (the launch model of
// torch.distributed.run and of MPI); what crosses the GPUs goes through a zk_comm:
//   * RCCL over xGMI (zk_comm_init: ncclCommInitRank on the context's device; byte all-to-all = grouped ncclSend/ncclRecv,
//     byte all-gather = ncclAllGather -- group elements cannot be all-reduced), or
//   * a caller-supplied transport (zk_comm_init_custom: function pointers; the CPU tests plug gloo in here).
// On top of it zk_mgpu runs the scalar-exchange prover as a software pipeline (formerly Python, distributed.py): in a round
// of `world` proofs rank j runs the SpMV / NTT stage of proof j only, four equal-split all-to-alls hand every rank the
// scalars that multiply ITS point range of the four inner products, the rank accumulates them for all proofs of the
// round in grouped MSMs, one more all-to-all returns the 768-byte partial sums to the proofs' owners.  The stages
// themselves are called through a zk_mgpu_backend table: the default one is this library's GPU entry points
// (zk_prove_scalars_submit / zk_prove_msm_submit / zk_prove_wait / zk_prove_combine); tests may substitute CPU stand-ins
// so that the pipeline logic and the collectives' order run under world-size-2 gloo without a GPU.
#include <rccl/rccl.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <thread>
#include <functional>
#include <vector>
#include "pipeline.hpp"

namespace zk { struct GpuBackend; }

struct zk_comm {
    zk_ctx* ctx = nullptr;          // null for a custom transport created without a device
    int rank = 0, world = 1;
    bool custom = false;
    zk_comm_ops ops{};
    // The handle and the flag are touched by two threads (zk_comm_abort is documented as callable while another thread waits for a
    // collective): whoever exchanges the handle out owns the ONE ncclCommAbort / ncclCommDestroy of it.
    std::atomic<ncclComm_t> nccl{nullptr};
    bool loopback = false;          // TIMING ONLY (ZK_COMM_LOOPBACK=1): several ranks played by device-to-device copies on this stream
    hipStream_t stream = nullptr;   // collectives run here, never on the compute streams
    int* d_flag = nullptr;          // barrier / max-reduce scratch (device)
    long timeout_ms = 120000;       // every host wait for a collective is bounded by this (zk_comm_set_timeout; ZK_COMM_TIMEOUT_MS)
    std::atomic<bool> aborted{false};   // a wait timed out or the caller gave up: the RCCL communicator is gone, every later call is refused
    // Enqueue paths hold this while they make their RCCL calls with the handle they loaded; zk_comm_abort takes it (for at most 200 ms:
    // an enqueue that blocks inside RCCL must not keep the abort away for ever) before it exchanges the handle out, so a communicator is
    // not aborted under a thread that is between its load and its last ncclSend / ncclRecv.
    std::timed_mutex enqueue_mu;
    // zk_mgpu provers that hold this communicator: zk_comm_destroy while users > 0 only marks it, the last zk_mgpu_destroy frees it --
    // either order of the two destroy calls is safe (the header documents "the communicator outlives its provers" anyway)
    std::atomic<int> users{0};
    std::atomic<bool> destroy_pending{false};
};

struct zk_mgpu {
    zk_comm* comm = nullptr;
    zk_mgpu_backend be{};
    bool own_backend = false;
    zk::GpuBackend* gpu = nullptr;
    size_t elems[4] = {0, 0, 0, 0};
    // two sets of exchange buffers (round k uses set k % 2): scalars out / in for L, V, U, H and the partial-sum blobs
    void* send[2][4] = {};
    void* recv[2][4] = {};
    void* part_send[2] = {};
    void* part_recv[2] = {};
    struct Round { uint64_t r[4], s[4]; int t_scalars = -1, t_msm = -1; bool ip_done = false; };
    std::deque<Round> rounds;       // pushed and not yet popped, oldest first
    size_t first = 0;               // round number of rounds.front()
    std::string last_error;
    bool failed = false;            // a stage or a collective failed: the pipeline state is unknown, every later call is refused
    bool counted = false;           // this prover is counted in comm->users
    bool cu_reserved = false;       // zk_mgpu_create masked the inner-product streams (comm_cu_reserve): undone by zk_mgpu_destroy
    // stream-ordered hand-overs (the library's own GPU stages over the library's own transport, see inner_products): events of round
    // k at index k % 3 -- scalars written, scalars exchanged, inner products done -- and the pinned landing places of the range flag
    // and of the proof
    bool ordered = false;
    hipEvent_t ev_scal[3] = {}, ev_a2a[3] = {}, ev_msm[3] = {};
    bool a2a_recorded[3] = {};
    int* h_flag = nullptr;          // 3 ints, pinned
    uint8_t* h_proof = nullptr;     // ZK_PROOF_BYTES, pinned
};

namespace zk {

#define ZK_NCCL(expr)                                                                                    \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) throw ::zk::StatusError{ZK_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(_r)}; \
    } while (0)

// ---- transport ----------------------------------------------------------------------------------
static void comm_live(const zk_comm* c);
static void comm_all_to_all(zk_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    comm_live(c);
    if (c->custom) {
        ZK_REQUIRE(c->ops.all_to_all(c->ops.user, d_send, d_recv, bytes_per_rank) == 0, ZK_ERR_COMM, "custom all_to_all failed");
        return;
    }
    std::lock_guard<std::timed_mutex> enq(c->enqueue_mu);
    ncclComm_t nc = c->nccl.load();
    if (!nc) {   // one rank, no communicator (loopback: the copies a `world`-rank exchange would receive, same sizes)
        ZK_HIP(hipMemcpyAsync(d_recv, d_send, bytes_per_rank * (c->loopback ? (size_t)c->world : 1), hipMemcpyDeviceToDevice, c->stream));
        return;
    }
    // chunk g of d_send goes to rank g; chunk j of d_recv comes from rank j.  xGMI is point to point (7 links per GPU): the
    // grouped send / recv pairs run on all links at once, which is what an equal-split exchange wants.
    ZK_NCCL(ncclGroupStart());
    for (int peer = 0; peer < c->world; ++peer) {
        ZK_NCCL(ncclSend((const uint8_t*)d_send + (size_t)peer * bytes_per_rank, bytes_per_rank, ncclUint8, peer, nc, c->stream));
        ZK_NCCL(ncclRecv((uint8_t*)d_recv + (size_t)peer * bytes_per_rank, bytes_per_rank, ncclUint8, peer, nc, c->stream));
    }
    ZK_NCCL(ncclGroupEnd());
}
static void comm_all_gather(zk_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    comm_live(c);
    if (c->custom) {
        ZK_REQUIRE(c->ops.all_gather(c->ops.user, d_send, d_recv, bytes_per_rank) == 0, ZK_ERR_COMM, "custom all_gather failed");
        return;
    }
    std::lock_guard<std::timed_mutex> enq(c->enqueue_mu);
    ncclComm_t nc = c->nccl.load();
    if (!nc) {
        for (int g = 0; g < (c->loopback ? c->world : 1); ++g)
            ZK_HIP(hipMemcpyAsync((uint8_t*)d_recv + (size_t)g * bytes_per_rank, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, c->stream));
        return;
    }
    ZK_NCCL(ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8, nc, c->stream));
}
// Tears the RCCL communicator down without waiting for its peers (ncclCommAbort makes the collectives' kernels exit), so that a
// rank whose peer died or never arrived gets its stream back instead of hanging in a device synchronisation for ever.
static void comm_abort(zk_comm* c) {
    c->aborted.store(true);      // enqueue paths that start from now on are refused (comm_live)
    std::unique_lock<std::timed_mutex> enq(c->enqueue_mu, std::chrono::milliseconds(200));   // one that is under way finishes its calls first (bounded)
    ncclComm_t nc = c->nccl.exchange(nullptr);   // a time-out's abort racing the caller's: one of them gets the handle
    if (nc) (void)ncclCommAbort(nc);
}
// Host wait for everything enqueued on the collectives' stream, BOUNDED: polls the stream (a spin for the first 2 ms -- the normal
// case is a few hundred microseconds --, then 50 us sleeps) and, when timeout_ms pass, aborts the communicator and reports
// ZK_ERR_COMM.  A custom transport completes inside the caller's callback.
// A drained stream is not proof of a completed collective: ncclCommAbort (zk_comm_abort from another thread, or this function's own
// time-out) makes RCCL's kernels LEAVE, and the stream completes over partial or untouched receive buffers.  So the flag is read
// again after the stream reports success, and an aborted wait answers ZK_ERR_COMM whatever the stream says.
static void comm_sync(zk_comm* c) {
    if (c->custom) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) {
            if (c->aborted.load()) throw StatusError{ZK_ERR_COMM, "zk_comm: the communicator was aborted while a collective was waited for; what it received is not valid"};
            return;
        }
        if (q != hipErrorNotReady) throw HipError{q, "hipStreamQuery (collectives' stream)"};
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (c->timeout_ms > 0 && ms > (double)c->timeout_ms) {
            comm_abort(c);
            (void)hipStreamSynchronize(c->stream);   // returns once the aborted kernels have left
            throw StatusError{ZK_ERR_COMM, "zk_comm: a collective did not complete within " + std::to_string(c->timeout_ms) + " ms (a peer is missing or hung); the communicator was aborted"};
        }
        if (ms > 2.0) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
static void comm_live(const zk_comm* c) {
    ZK_REQUIRE(!c->aborted.load(), ZK_ERR_COMM, "zk_comm: the communicator was aborted (an earlier collective timed out); create a new one");
}

// ---- the default backend: this library's GPU stages -----------------------------------------------
struct GpuBackend {
    zk_ctx* ctx;
    const zk_crs* crs;
    const zk_qap* qap;
    bool host_witness = false;   // the next scalars_submit gets a host pointer (zk_mgpu_push_host)
};
static int gpu_elems(void* u, int world, size_t out[4]) { return zk_prove_exchange_elems(((GpuBackend*)u)->qap, world, out); }
static void* gpu_alloc(void* u, size_t bytes) {
    void* p = nullptr;
    (void)hipSetDevice(((GpuBackend*)u)->ctx->device);
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, bytes ? bytes : 1);
    return p;
}
static void gpu_free(void*, void* p) { (void)hipFree(p); }
static int gpu_scalars(void* u, const void* w, size_t m, const uint64_t r[4], const uint64_t s[4], int world, void* const send[4], int* t) {
    GpuBackend* g = (GpuBackend*)u;
    if (g->host_witness) return zk_prove_scalars_submit_host(g->ctx, g->crs, g->qap, (const uint64_t*)w, m, r, s, world, send[0], send[1], send[2], send[3], t);
    return zk_prove_scalars_submit(g->ctx, g->crs, g->qap, w, m, r, s, world, send[0], send[1], send[2], send[3], t);
}
static int gpu_msm(void* u, int sets, int rank, int world, void* const recv[4], void* part, int* t) {
    GpuBackend* g = (GpuBackend*)u;
    return zk_prove_msm_submit(g->ctx, g->crs, g->qap, sets, rank, world, recv[0], recv[1], recv[2], recv[3], part, t);
}
static int gpu_wait(void* u, int t) { return zk_prove_wait(((GpuBackend*)u)->ctx, t, nullptr); }
static int gpu_combine(void* u, const void* part_recv, int world, const uint64_t r[4], const uint64_t s[4], uint8_t* proof) {
    GpuBackend* g = (GpuBackend*)u;
    return zk_prove_combine(g->ctx, g->crs, part_recv, world, r, s, proof);
}

// ---- the pipeline ---------------------------------------------------------------------------------
// Round k: A(k) = scalars of this rank's proof k; B(k) = wait A(k), four all-to-alls, grouped inner products over this rank's
// points for the `world` proofs of the round; F(k) = wait B(k), all-to-all of the partial sums, assembly of proof k.
// Issue order (every rank the same, so the collectives pair up): push(j) = A(j), then B(j-1); pop() = F(oldest), after
// B(oldest) if the caller did not push ahead.  With two rounds pushed ahead of every pop this is the schedule measured in
// round 1: the scalars run two rounds ahead of the inner products (under the chip-filling accumulations the SpMV / NTT stage
// only progresses in the gaps) and the inner products of round k+1 are queued before those of round k end.
static void be_check(zk_mgpu* g, int rc, const char* what) {
    if (rc != 0) throw StatusError{rc, std::string("zk_mgpu: ") + what + " failed"};
}
//
// Hand-overs between the stages.  With a caller's transport or caller's stages (the gloo / CPU stand-in tests) they go through the
// host: wait for the ticket, run the collective, wait for it, submit the next stage.  With the library's own stages over the
// library's own transport (g->ordered) nothing between A, B and F blocks the host: the collectives' stream waits for an event
// recorded behind the scalars, the inner products' streams wait for an event recorded behind the all-to-alls, the partial-sum
// exchange and the assembly follow the inner products' event on the collectives' stream, and the scalars of round k + 2 wait for the
// exchange of round k (which reads the send buffers they overwrite).  Tickets are released as soon as their work is enqueued
// (prove_release); the one host synchronisation of a round is at the end of zk_mgpu_pop, which returns the proof bytes.
static void inner_products(zk_mgpu* g, size_t k) {
    zk_mgpu::Round& R = g->rounds[k - g->first];
    if (R.ip_done) return;
    const int set = (int)(k & 1), world = g->comm->world;
    if (g->ordered) {
        zk_comm* c = g->comm;
        zk_ctx* ctx = g->gpu->ctx;
        const int e = (int)(k % 3);
        ZK_HIP(hipStreamWaitEvent(c->stream, g->ev_scal[e], 0));
        for (int a = 0; a < 4; ++a) comm_all_to_all(c, g->send[set][a], g->recv[set][a], g->elems[a] / world * 32);
        ZK_HIP(hipEventRecord(g->ev_a2a[e], c->stream));
        g->a2a_recorded[e] = true;
        ctx->submit_wait_evt = g->ev_a2a[e];
        const int rc = g->be.msm_submit(g->be.user, world, c->rank, world, g->recv[set], g->part_send[set], &R.t_msm);
        ctx->submit_wait_evt = nullptr;
        be_check(g, rc, "msm_submit");
        ZK_HIP(hipEventRecord(g->ev_msm[e], prove_ticket_stream(ctx, R.t_msm)));
        prove_release(ctx, R.t_msm, nullptr);
        R.t_msm = -1;
        R.ip_done = true;
        return;
    }
    be_check(g, g->be.wait(g->be.user, R.t_scalars), "wait (scalars)");
    R.t_scalars = -1;
    for (int a = 0; a < 4; ++a) comm_all_to_all(g->comm, g->send[set][a], g->recv[set][a], g->elems[a] / world * 32);
    comm_sync(g->comm);
    be_check(g, g->be.msm_submit(g->be.user, world, g->comm->rank, world, g->recv[set], g->part_send[set], &R.t_msm), "msm_submit");
    R.ip_done = true;
}

}  // namespace zk

using namespace zk;

static int comm_guard(zk_comm* c, std::string* err, const std::function<void()>& fn) {
    try {
        if (c && c->ctx) ZK_HIP(hipSetDevice(c->ctx->device));
        fn();
        return ZK_OK;
    } catch (const HipError& e) {
        if (err) *err = std::string(hipGetErrorString(e.code)) + " in " + e.where;
        if (c && c->ctx) c->ctx->last_error = std::string(hipGetErrorString(e.code)) + " in " + e.where;
        return ZK_ERR_HIP;
    } catch (const StatusError& e) {
        if (err) *err = e.msg;
        if (c && c->ctx) c->ctx->last_error = e.msg;
        return e.status;
    } catch (...) {
        if (err) *err = "unknown error";
        return ZK_ERR_ARG;
    }
}

extern "C" {

int zk_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int zk_comm_unique_id(uint8_t id_out[ZK_COMM_ID_BYTES]) {
    static_assert(ZK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id_out) return ZK_ERR_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return ZK_ERR_COMM;
    std::memcpy(id_out, id.internal, ZK_COMM_ID_BYTES);
    return ZK_OK;
}

int zk_comm_init(zk_ctx* ctx, const uint8_t id[ZK_COMM_ID_BYTES], int rank, int world, zk_comm** out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return ZK_ERR_ARG;
    *out = nullptr;
    zk_comm* c = new (std::nothrow) zk_comm();
    if (!c) return ZK_ERR_HIP;
    c->ctx = ctx; c->rank = rank; c->world = world;
    int rc = comm_guard(c, nullptr, [&] {
        // The collectives' stream is created at the HIGH priority level.  Streams of one level share ~4 hardware queues, and a fifth
        // stream of a level serialises with another: the mid level holds the four inner-product streams, and of the
        // high level's main, alternate main, side and finish streams the side stream has no work in the multi-GPU pipeline (the
        // assembly of a round runs on THIS stream, behind the partial-sum exchange) -- so four streams of the level are in use
        // whatever the mode.  RCCL's send / receive kernels are then dispatched ahead of pending accumulation workgroups.
        int prio_least = 0, prio_greatest = 0;
        ZK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        ZK_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest));
        ZK_HIP(hipMalloc((void**)&c->d_flag, 64));
        // ZK_COMM_FORCE_RCCL=1: a one-rank communicator too goes through RCCL (self send / recv, all-gather, all-reduce) -- the
        // only way to execute this file's RCCL calls on a one-GPU box (tests/test_gpu_bench.py)
        if (const char* tmo = std::getenv("ZK_COMM_TIMEOUT_MS")) {
            // a malformed value must not silently become 0 = "wait for ever"
            char* end = nullptr;
            errno = 0;
            const long v = std::strtol(tmo, &end, 10);
            ZK_REQUIRE(end != tmo && *end == '\0' && errno == 0 && v >= 0, ZK_ERR_ARG, "zk_comm_init: ZK_COMM_TIMEOUT_MS must be a non-negative integer (milliseconds; 0 = unbounded)");
            c->timeout_ms = v;
        }
        const char* force = std::getenv("ZK_COMM_FORCE_RCCL");
        // ZK_COMM_LOOPBACK=1 (bench.py --emulate-world): rank 0 of `world` ranks with copies in place of the collectives -- one rank's
        // work of a `world`-GPU run through the same code path (stream-ordered hand-overs included).  The sums it forms are NOT
        // proofs, so the switch exists in the measurement build only (ZK_MEASURE; zk_mgpu_pop then hands out zero bytes); the
        // product library refuses the variable instead of ignoring it -- a stray setting must not change what a prover returns.
        const char* loop = std::getenv("ZK_COMM_LOOPBACK");
        const bool want_loop = world > 1 && loop && loop[0] == '1';
#ifndef ZK_MEASURE
        ZK_REQUIRE(!want_loop, ZK_ERR_UNSUPPORTED, "zk_comm_init: ZK_COMM_LOOPBACK is a measurement switch (libzkgpu_measure.so); unset it");
#endif
        if (want_loop && rank == 0) {
            c->loopback = true;
        } else if (world > 1 || (force && force[0] == '1' && id)) {
            ZK_HIP(hipSetDevice(ctx->device));
            ncclUniqueId nid;
            std::memcpy(nid.internal, id, ZK_COMM_ID_BYTES);
            // ncclCommInitRank blocks until every rank has joined: bounded like the collectives (timeout_ms).  The call runs on
            // a helper thread; when the bound passes this rank gives up, and the helper -- which cannot be cancelled inside RCCL's
            // bootstrap -- aborts whatever communicator it may still get and ends on its own.
            struct InitJob { std::mutex m; std::condition_variable cv; bool done = false, abandoned = false; ncclResult_t res = ncclSuccess; ncclComm_t comm = nullptr; };
            auto job = std::make_shared<InitJob>();
            const int device = ctx->device;
            std::thread([job, world, nid, rank, device] {
                (void)hipSetDevice(device);
                ncclComm_t nc = nullptr;
                const ncclResult_t r = ncclCommInitRank(&nc, world, nid, rank);
                std::lock_guard<std::mutex> lk(job->m);
                job->res = r; job->comm = nc; job->done = true;
                if (job->abandoned && nc) (void)ncclCommAbort(nc);
                job->cv.notify_all();
            }).detach();
            {
                std::unique_lock<std::mutex> lk(job->m);
                if (c->timeout_ms > 0) {
                    if (!job->cv.wait_for(lk, std::chrono::milliseconds(c->timeout_ms), [&] { return job->done; })) {
                        job->abandoned = true;
                        throw StatusError{ZK_ERR_COMM, "zk_comm_init: ncclCommInitRank did not return within " + std::to_string(c->timeout_ms) + " ms (a rank is missing)"};
                    }
                } else {
                    job->cv.wait(lk, [&] { return job->done; });
                }
                if (job->res != ncclSuccess) throw StatusError{ZK_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(job->res)};
                c->nccl.store(job->comm);
            }
        }
    });
    if (rc != ZK_OK) { zk_comm_destroy(c); return rc; }
    *out = c;
    return ZK_OK;
}

int zk_comm_init_custom(zk_ctx* ctx, const zk_comm_ops* ops, int rank, int world, zk_comm** out) {
    if (!out || !ops || !ops->all_to_all || !ops->all_gather || world < 1 || rank < 0 || rank >= world) return ZK_ERR_ARG;
    zk_comm* c = new (std::nothrow) zk_comm();
    if (!c) return ZK_ERR_HIP;
    c->ctx = ctx; c->rank = rank; c->world = world; c->custom = true; c->ops = *ops;
    *out = c;
    return ZK_OK;
}

void zk_comm_destroy(zk_comm* c) {
    if (!c) return;
    if (c->users.load() > 0) { c->destroy_pending.store(true); return; }   // a prover still holds it: freed by the last zk_mgpu_destroy
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (ncclComm_t nc = c->nccl.exchange(nullptr)) (void)ncclCommDestroy(nc);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
int zk_comm_rank(const zk_comm* c) { return c ? c->rank : -1; }
int zk_comm_world(const zk_comm* c) { return c ? c->world : 0; }
/* ranks of the RCCL communicator behind this zk_comm as RCCL itself counts them (ncclCommCount); 0 = no RCCL communicator
 * (one rank without ZK_COMM_FORCE_RCCL, a caller's transport, the loop-back of the measurement build, or aborted) */
int zk_comm_rccl_ranks(const zk_comm* c) {
    ncclComm_t nc = c ? c->nccl.load() : nullptr;
    if (!nc) return 0;
    int n = 0;
    return ncclCommCount(nc, &n) == ncclSuccess ? n : 0;
}
/* bound (ms) of every host wait for a collective of this communicator: barrier, max, all-to-all, all-gather, zk_mgpu_pop.
 * 0 = wait for ever.  Default 120000, or the environment variable ZK_COMM_TIMEOUT_MS at zk_comm_init. */
int zk_comm_set_timeout(zk_comm* c, long ms) {
    if (!c || ms < 0) return ZK_ERR_ARG;
    c->timeout_ms = ms;
    return ZK_OK;
}
/* gives up on the peers: the RCCL communicator is aborted (its kernels leave the GPU), later collectives answer ZK_ERR_COMM.
 * May be called from another thread while a collective of this communicator is being waited for. */
int zk_comm_abort(zk_comm* c) {
    if (!c) return ZK_ERR_ARG;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    comm_abort(c);
    return ZK_OK;
}

/* Every rank blocks until all have arrived (an all-gather of one flag per rank). */
int zk_comm_barrier(zk_comm* c) {
    if (!c) return ZK_ERR_ARG;
    if (c->custom) return c->ops.barrier ? c->ops.barrier(c->ops.user) : ZK_ERR_UNSUPPORTED;
    return comm_guard(c, nullptr, [&] {
        comm_live(c);
        {
            std::lock_guard<std::timed_mutex> enq(c->enqueue_mu);
            if (ncclComm_t nc = c->nccl.load()) ZK_NCCL(ncclAllReduce(c->d_flag, c->d_flag + 1, 1, ncclInt32, ncclSum, nc, c->stream));
        }
        comm_sync(c);
    });
}
/* *value = max over the ranks of *value (the bench's "time of the slowest rank") */
int zk_comm_max_f64(zk_comm* c, double* value) {
    if (!c || !value) return ZK_ERR_ARG;
    if (c->custom) return c->ops.max_f64 ? c->ops.max_f64(c->ops.user, value) : ZK_ERR_UNSUPPORTED;
    return comm_guard(c, nullptr, [&] {
        comm_live(c);
        {
            std::lock_guard<std::timed_mutex> enq(c->enqueue_mu);
            ncclComm_t nc = c->nccl.load();
            if (!nc) return;
            double* d = reinterpret_cast<double*>(c->d_flag) + 2;
            ZK_HIP(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
            ZK_NCCL(ncclAllReduce(d, d + 1, 1, ncclDouble, ncclMax, nc, c->stream));
            ZK_HIP(hipMemcpyAsync(value, d + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        comm_sync(c);
    });
}
/* Byte collectives on device buffers, for callers that drive the stages themselves; complete on return. */
int zk_comm_all_to_all(zk_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    if (!c || !d_send || !d_recv) return ZK_ERR_ARG;
    return comm_guard(c, nullptr, [&] { comm_all_to_all(c, d_send, d_recv, bytes_per_rank); comm_sync(c); });
}
int zk_comm_all_gather(zk_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    if (!c || !d_send || !d_recv) return ZK_ERR_ARG;
    return comm_guard(c, nullptr, [&] { comm_all_gather(c, d_send, d_recv, bytes_per_rank); comm_sync(c); });
}

/* ---- latency form: ONE proof over all ranks (every rank holds the same witness) ---- */
int zk_mgpu_prove_sharded(zk_ctx* ctx, zk_comm* c, const zk_crs* crs, const zk_qap* qap, const void* d_weights, size_t m,
                          const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[ZK_PROOF_BYTES]) {
    if (!ctx || !c || !crs || !qap || !d_weights || !r || !s || !proof_out) return ZK_ERR_ARG;
    void *part = nullptr, *all = nullptr;
    int rc = comm_guard(c, nullptr, [&] {
        ZK_HIP(hipMalloc(&part, ZK_PARTIAL_BYTES));
        ZK_HIP(hipMalloc(&all, (size_t)c->world * ZK_PARTIAL_BYTES));
        int st = zk_prove_partial(ctx, crs, qap, d_weights, m, r, s, c->rank, c->world, part);
        ZK_REQUIRE(st == ZK_OK, st, ctx->last_error);
        comm_all_gather(c, part, all, ZK_PARTIAL_BYTES);
        comm_sync(c);
        st = zk_prove_combine(ctx, crs, all, c->world, r, s, proof_out);
        ZK_REQUIRE(st == ZK_OK, st, ctx->last_error);
    });
    if (part) (void)hipFree(part);
    if (all) (void)hipFree(all);
    return rc;
}

/* ---- throughput form: the scalar-exchange pipeline ---- */
static int mgpu_create(zk_comm* c, const zk_mgpu_backend* be, GpuBackend* gpu, zk_mgpu** out) {
    zk_mgpu* g = new (std::nothrow) zk_mgpu();
    if (!g) { delete gpu; return ZK_ERR_HIP; }
    g->comm = c; g->be = *be; g->gpu = gpu;
    c->users.fetch_add(1); g->counted = true;
    int rc = comm_guard(c, &g->last_error, [&] {
        be_check(g, g->be.elems(g->be.user, c->world, g->elems), "elems");
        for (int set = 0; set < 2; ++set) {
            for (int a = 0; a < 4; ++a) {
                ZK_REQUIRE(g->elems[a] % (size_t)c->world == 0, ZK_ERR_ARG, "zk_mgpu: exchange arrays must split evenly");
                g->send[set][a] = g->be.alloc(g->be.user, g->elems[a] * 32);
                g->recv[set][a] = g->be.alloc(g->be.user, g->elems[a] * 32);
                ZK_REQUIRE(g->send[set][a] && g->recv[set][a], ZK_ERR_HIP, "zk_mgpu: buffer allocation failed");
            }
            g->part_send[set] = g->be.alloc(g->be.user, (size_t)c->world * ZK_PARTIAL_BYTES);
            g->part_recv[set] = g->be.alloc(g->be.user, (size_t)c->world * ZK_PARTIAL_BYTES);
            ZK_REQUIRE(g->part_send[set] && g->part_recv[set], ZK_ERR_HIP, "zk_mgpu: buffer allocation failed");
        }
    });
    if (rc == ZK_OK && gpu && !c->custom && c->stream && !std::getenv("ZK_MGPU_HOST_HANDOVER")) {
        rc = comm_guard(c, &g->last_error, [&] {
            for (int e = 0; e < 3; ++e) {
                ZK_HIP(hipEventCreateWithFlags(&g->ev_scal[e], hipEventDisableTiming));
                ZK_HIP(hipEventCreateWithFlags(&g->ev_a2a[e], hipEventDisableTiming));
                ZK_HIP(hipEventCreateWithFlags(&g->ev_msm[e], hipEventDisableTiming));
            }
            ZK_HIP(hipHostMalloc((void**)&g->h_flag, 3 * sizeof(int)));
            ZK_HIP(hipHostMalloc((void**)&g->h_proof, ZK_PROOF_BYTES));
            std::memset(g->h_flag, 0, 3 * sizeof(int));
            g->ordered = true;
        });
    }
    if (rc != ZK_OK) { zk_mgpu_destroy(g); return rc; }
    *out = g;
    return ZK_OK;
}

int zk_mgpu_create(zk_ctx* ctx, zk_comm* c, const zk_crs* crs, const zk_qap* qap, zk_mgpu** out) {
    if (!ctx || !c || !crs || !qap || !out) return ZK_ERR_ARG;
    if (qap->dense) return ZK_ERR_UNSUPPORTED;
    GpuBackend* gpu = new (std::nothrow) GpuBackend{ctx, crs, qap};
    if (!gpu) return ZK_ERR_HIP;
    zk_mgpu_backend be{gpu, gpu_elems, gpu_alloc, gpu_free, gpu_scalars, gpu_msm, gpu_wait, gpu_combine};
    // over RCCL the inner-product streams leave comm_cu_reserve units per XCD to the collectives' kernels (a straggling exchange on
    // one rank stalls every peer); a lone rank or a caller's transport keeps the whole chip
    const bool reserve = c->nccl.load() && ctx->opt_comm_cu_reserve > 0;   // an RCCL communicator (several ranks, or one with ZK_COMM_FORCE_RCCL)
    if (reserve) {
        const int rc = comm_guard(c, nullptr, [&] { ctx_reserve_cus(ctx, (int)ctx->opt_comm_cu_reserve); });
        if (rc != ZK_OK) { delete gpu; return rc; }
    }
    const int rc = mgpu_create(c, &be, gpu, out);
    if (rc == ZK_OK) (*out)->cu_reserved = reserve;
    return rc;
}
int zk_mgpu_create_custom(zk_comm* c, const zk_mgpu_backend* be, zk_mgpu** out) {
    if (!c || !be || !out || !be->elems || !be->alloc || !be->free || !be->scalars_submit || !be->msm_submit || !be->wait || !be->combine) return ZK_ERR_ARG;
    return mgpu_create(c, be, nullptr, out);
}

void zk_mgpu_destroy(zk_mgpu* g) {
    if (!g) return;
    if (g->gpu) {   // whatever is still enqueued (released tickets, collectives) must end before the buffers go
        (void)hipSetDevice(g->gpu->ctx->device);
        // ... but a collective whose peer hangs never ends: the bounded wait first (it aborts the communicator when the time-out
        // passes, after which the device synchronisation below returns), as zkgpu.h promises for every host wait of a collective
        if (g->comm && !g->comm->custom && g->comm->stream) {
            try { comm_sync(g->comm); } catch (...) {}
        }
        (void)hipDeviceSynchronize();
    }
    if (g->gpu && g->cu_reserved) {
        try { ctx_reserve_cus(g->gpu->ctx, 0); } catch (...) {}
    }
    for (auto& R : g->rounds) {   // tickets still held
        if (R.t_msm >= 0) (void)g->be.wait(g->be.user, R.t_msm);
        if (R.t_scalars >= 0) (void)g->be.wait(g->be.user, R.t_scalars);
    }
    for (int e = 0; e < 3; ++e)
        for (hipEvent_t ev : {g->ev_scal[e], g->ev_a2a[e], g->ev_msm[e]})
            if (ev) (void)hipEventDestroy(ev);
    if (g->h_flag) (void)hipHostFree(g->h_flag);
    if (g->h_proof) (void)hipHostFree(g->h_proof);
    for (int set = 0; set < 2; ++set) {
        for (int a = 0; a < 4; ++a) {
            if (g->send[set][a]) g->be.free(g->be.user, g->send[set][a]);
            if (g->recv[set][a]) g->be.free(g->be.user, g->recv[set][a]);
        }
        if (g->part_send[set]) g->be.free(g->be.user, g->part_send[set]);
        if (g->part_recv[set]) g->be.free(g->be.user, g->part_recv[set]);
    }
    delete g->gpu;
    zk_comm* const held = g->counted ? g->comm : nullptr;
    delete g;
    if (held && held->users.fetch_sub(1) == 1 && held->destroy_pending.load()) zk_comm_destroy(held);
}
const char* zk_mgpu_last_error(const zk_mgpu* g) { return g ? g->last_error.c_str() : "null prover"; }

static int mgpu_push(zk_mgpu* g, const void* d_weights, size_t m, const uint64_t r[4], const uint64_t s[4], bool host) {
    if (!g || !d_weights || !r || !s) return ZK_ERR_ARG;
    if (g->failed) { g->last_error = "zk_mgpu: an earlier stage or collective failed; destroy the prover"; return ZK_ERR_COMM; }
    if (g->rounds.size() >= 3) { g->last_error = "zk_mgpu_push: three rounds in flight (call zk_mgpu_pop first)"; return ZK_ERR_ARG; }
    const int rc = comm_guard(g->comm, &g->last_error, [&] {
        const size_t k = g->first + g->rounds.size();
        // the send buffers of set k % 2 were last used by round k - 2: its exchange must be issued (it normally is, from push(k - 1))
        if (g->rounds.size() >= 2) inner_products(g, k - 2);
        zk_mgpu::Round R;
        std::memcpy(R.r, r, 32); std::memcpy(R.s, s, 32);
        if (g->gpu) g->gpu->host_witness = host;
        if (g->ordered) {
            zk_ctx* ctx = g->gpu->ctx;
            const int e = (int)(k % 3), e2 = (int)((k + 1) % 3);   // (k - 2) % 3 == (k + 1) % 3
            // ... and complete before the scalars of this round overwrite what it sends: ordered on the device, not by the host
            if (k >= 2 && g->a2a_recorded[e2]) ctx->submit_wait_evt = g->ev_a2a[e2];
            const int st = g->be.scalars_submit(g->be.user, d_weights, m, r, s, g->comm->world, g->send[k & 1], &R.t_scalars);
            ctx->submit_wait_evt = nullptr;
            be_check(g, st, "scalars_submit");
            // the range flag's copy to pinned memory is enqueued by prove_release on the ticket's stream: the event is recorded BEHIND
            // it, so that whatever follows ev_scal (the exchange, the inner products, the pop's synchronisation) also follows the copy
            hipStream_t fin = prove_ticket_stream(ctx, R.t_scalars);
            prove_release(ctx, R.t_scalars, &g->h_flag[e]);
            ZK_HIP(hipEventRecord(g->ev_scal[e], fin));
            R.t_scalars = -1;
        } else {
            be_check(g, g->be.scalars_submit(g->be.user, d_weights, m, r, s, g->comm->world, g->send[k & 1], &R.t_scalars), "scalars_submit");
        }
        g->rounds.push_back(R);
        if (g->rounds.size() >= 2) inner_products(g, k - 1);   // A(k), then B(k - 1): the scalars run ahead of the inner products
    });
    // A failed stage leaves tickets, buffers and -- worst -- the other ranks' collectives in an unknown state: the prover refuses
    // every later call (the peers block in their collective until the caller tears the job down; RCCL has no poison message).
    if (rc != ZK_OK) g->failed = true;
    return rc;
}

int zk_mgpu_push(zk_mgpu* g, const void* d_weights, size_t m, const uint64_t r[4], const uint64_t s[4]) {
    return mgpu_push(g, d_weights, m, r, s, false);
}
// the witness in HOST memory (page-locked from zk_host_alloc: the copy overlaps the previous rounds' inner products); a custom
// backend receives the pointer as it is
int zk_mgpu_push_host(zk_mgpu* g, const uint64_t* weights, size_t m, const uint64_t r[4], const uint64_t s[4]) {
    return mgpu_push(g, weights, m, r, s, true);
}

int zk_mgpu_pop(zk_mgpu* g, uint8_t proof_out[ZK_PROOF_BYTES]) {
    if (!g || !proof_out) return ZK_ERR_ARG;
    if (g->failed) { g->last_error = "zk_mgpu: an earlier stage or collective failed; destroy the prover"; return ZK_ERR_COMM; }
    if (g->rounds.empty()) { g->last_error = "zk_mgpu_pop: nothing pushed"; return ZK_ERR_ARG; }
    const int rc = comm_guard(g->comm, &g->last_error, [&] {
        const size_t k = g->first;
        // the collectives must be issued in round order on every rank: B(k), then B(k+1) if it was pushed, then F(k)
        inner_products(g, k);
        if (g->rounds.size() >= 2) inner_products(g, k + 1);
        zk_mgpu::Round R = g->rounds.front();
        const int set = (int)(k & 1), world = g->comm->world;
        if (g->ordered) {
            zk_comm* c = g->comm;
            const int e = (int)(k % 3);
            ZK_HIP(hipStreamWaitEvent(c->stream, g->ev_msm[e], 0));
            comm_all_to_all(c, g->part_send[set], g->part_recv[set], ZK_PARTIAL_BYTES);
            prove_combine_on(g->gpu->ctx, *g->gpu->crs, g->part_recv[set], world, R.r, R.s, c->stream, g->h_proof);
            comm_sync(c);   // the round's one host synchronisation (bounded: zk_comm_set_timeout)
            g->rounds.pop_front();
            g->first = k + 1;
            g->gpu->ctx->resolve_profile(-2);
            ZK_REQUIRE(!g->h_flag[e], ZK_ERR_RANGE, "prove: witness element >= r");
            if (c->loopback) std::memset(proof_out, 0, ZK_PROOF_BYTES);   // timing run: what was assembled is not a proof
            else std::memcpy(proof_out, g->h_proof, ZK_PROOF_BYTES);
            return;
        }
        be_check(g, g->be.wait(g->be.user, R.t_msm), "wait (inner products)");
        g->rounds.front().t_msm = -1;
        comm_all_to_all(g->comm, g->part_send[set], g->part_recv[set], ZK_PARTIAL_BYTES);
        comm_sync(g->comm);
        g->rounds.pop_front();
        g->first = k + 1;
        be_check(g, g->be.combine(g->be.user, g->part_recv[set], world, R.r, R.s, proof_out), "combine");
    });
    if (rc != ZK_OK && rc != ZK_ERR_RANGE) g->failed = true;   // a witness out of range fails its own proof only: the round completed on every rank
    return rc;
}

}  // extern "C"
