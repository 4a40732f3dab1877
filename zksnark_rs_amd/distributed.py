"""Multi-GPU prove.  Two protocols over torch.distributed (backend "nccl" is RCCL on ROCm):

prove_exchange_stream -- the inner products of EVERY proof are sharded over the ranks by point ranges, and the SpMV /
NTT stage is not repeated: in a round of `world` proofs rank j runs that stage for proof j only, an all-to-all hands
every rank the scalars that multiply its own points (4 n / world x 32 B per proof and rank), the rank accumulates
them for all proofs of the round, a second all-to-all returns the 768-byte partial sums to the owners.  Per-GPU
work per round is one proof's worth whatever the world size.

prove_sharded / prove_sharded_stream -- the latency form: one proof at a time, every rank repeats the NTT stage and
accumulates its share (windows or point ranges), one all-gather of 768-byte partial sums:

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  Every rank holds the CRS
and the QAP, recomputes the cheap NTT stage, accumulates only the Pippenger windows
w = rank (mod world) of the inner products (zk_prove_partial), all-gathers the partial sums
as raw bytes (RCCL cannot reduce group elements) and finishes locally (zk_prove_combine).
SURVEY.md 8e.  The same function drives the CPU `gloo` test with a stand-in backend.
"""
import ctypes as C
import os

import numpy as np

from . import PARTIAL_BYTES, PROOF_BYTES, ZkError, _lib, fr_to_limbs


# ---- the C ABI's own multi-GPU layer (csrc/comm.hip): RCCL inside libzkgpu.so, no torch collective on the data path ----
def _fr(x):
    a = np.ascontiguousarray(fr_to_limbs(x) if isinstance(x, int) else x, dtype=np.uint64)
    return a, a.ctypes.data_as(_lib.u64p)


class Comm:
    """zk_comm.  Comm(ctx, rank, world, id_bytes) = RCCL (zk_comm_init; rank 0 draws the id with Comm.unique_id() and ships it by any
    channel); Comm(ctx_or_None, rank, world, ops=CommOps) = caller-supplied transport (zk_comm_init_custom)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * _lib.COMM_ID_BYTES)()
        rc = _lib.load().zk_comm_unique_id(buf)
        if rc != 0:
            raise ZkError(rc)
        return bytes(buf)

    def __init__(self, ctx, rank, world, id_bytes=None, ops=None):
        self.lib = _lib.load()
        self.ctx, self.rank, self.world, self._ops = ctx, rank, world, ops
        p = C.c_void_p()
        cptr = ctx.ptr if ctx is not None else None
        if ops is not None:
            rc = self.lib.zk_comm_init_custom(cptr, C.byref(ops), rank, world, C.byref(p))
        else:
            idb = (C.c_uint8 * _lib.COMM_ID_BYTES).from_buffer_copy(id_bytes) if id_bytes is not None else None
            rc = self.lib.zk_comm_init(cptr, idb, rank, world, C.byref(p))
        if rc != 0:
            raise ZkError(rc, self.lib.zk_last_error(cptr).decode() if cptr else "")
        self.ptr = p

    def _check(self, rc):
        if rc != 0:
            raise ZkError(rc, self.lib.zk_last_error(self.ctx.ptr).decode() if self.ctx is not None else "")

    def barrier(self):
        self._check(self.lib.zk_comm_barrier(self.ptr))

    def max_f64(self, v):
        d = C.c_double(v)
        self._check(self.lib.zk_comm_max_f64(self.ptr, C.byref(d)))
        return d.value

    def all_to_all(self, d_send, d_recv, bytes_per_rank):
        self._check(self.lib.zk_comm_all_to_all(self.ptr, C.c_void_p(d_send), C.c_void_p(d_recv), bytes_per_rank))

    def all_gather(self, d_send, d_recv, bytes_per_rank):
        self._check(self.lib.zk_comm_all_gather(self.ptr, C.c_void_p(d_send), C.c_void_p(d_recv), bytes_per_rank))

    def rccl_ranks(self):
        """ranks of the RCCL communicator as RCCL counts them (0: none -- one rank, a caller's transport, aborted)"""
        return int(self.lib.zk_comm_rccl_ranks(self.ptr))

    def set_timeout(self, ms):
        self._check(self.lib.zk_comm_set_timeout(self.ptr, int(ms)))

    def abort(self):
        self.lib.zk_comm_abort(self.ptr)

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.zk_comm_destroy(self.ptr)
            self.ptr = None


def bootstrap_comm(ctx, rank, world, addr=None, port=None):
    """One process per GPU launched by torch.distributed.run / torchrun: rank 0 draws the RCCL id and publishes it through a
    TCP key-value store on MASTER_ADDR : MASTER_PORT + 1 (a store, not a collective -- torch.distributed is never
    initialised); every rank then joins the communicator inside libzkgpu.so."""
    if world == 1:
        return Comm(ctx, 0, 1)
    from datetime import timedelta
    from torch.distributed import TCPStore
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + 1)
    store = TCPStore(addr, port, world, rank == 0, timeout=timedelta(seconds=300))
    if rank == 0:
        store.set("zk_comm_id", Comm.unique_id())
    comm = Comm(ctx, rank, world, bytes(store.get("zk_comm_id")))
    comm._store = store      # keeps rank 0's server alive as long as the communicator
    return comm


def gloo_comm(ctx, dist, rank, world):
    """A zk_comm whose transport is torch.distributed's gloo backend, with the device buffers of the GPU backend staged through
    the host (zk_comm_init_custom).  For functional runs of the C pipeline with several ranks on ONE GPU, where RCCL refuses
    to form a communicator; never the measured configuration."""
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    D2H, H2D = 2, 1

    def staged(fn, send, recv, n_send, n_recv):
        hs, hr = np.empty(n_send, np.uint8), np.empty(n_recv, np.uint8)
        if hip.hipMemcpy(hs.ctypes.data, send, n_send, D2H) != 0:
            return 1
        fn(torch.from_numpy(hr), torch.from_numpy(hs))
        return 1 if hip.hipMemcpy(recv, hr.ctypes.data, n_recv, H2D) != 0 else 0

    def a2a(user, send, recv, per_rank):
        return staged(dist.all_to_all_single, send, recv, per_rank * world, per_rank * world)

    def gather(user, send, recv, per_rank):
        return staged(dist.all_gather_into_tensor, send, recv, per_rank, per_rank * world)

    def barrier(user):
        dist.barrier()
        return 0

    def max_f64(user, val):
        t = torch.tensor([val[0]], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        val[0] = float(t.item())
        return 0

    ops = _lib.CommOps(None, _lib.A2A_FN(a2a), _lib.A2A_FN(gather), _lib.BARRIER_FN(barrier), _lib.MAXF64_FN(max_f64))
    return Comm(ctx, rank, world, ops=ops)


def loopback_comm(ctx, world):
    """TIMING ONLY: rank 0 of a `world`-rank communicator whose collectives are device-to-device copies of the right size on
    this GPU (chunk g of the send buffer lands in chunk g of the receive buffer).  The C pipeline then does exactly one rank's
    work of a `world`-GPU run -- one SpMV / NTT stage per round, `world` groups of inner products over 1/world of the points,
    one assembly -- but the proofs it returns are NOT valid (bench.py --emulate-world)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    D2D = 3

    def a2a(user, send, recv, per_rank):
        return 1 if hip.hipMemcpy(recv, send, per_rank * world, D2D) != 0 else 0

    def gather(user, send, recv, per_rank):
        for g in range(world):
            if hip.hipMemcpy(recv + g * per_rank, send, per_rank, D2D) != 0:
                return 1
        return 0

    def max_f64(user, val):
        return 0

    ops = _lib.CommOps(None, _lib.A2A_FN(a2a), _lib.A2A_FN(gather), _lib.BARRIER_FN(lambda user: 0), _lib.MAXF64_FN(max_f64))
    return Comm(ctx, 0, world, ops=ops)


class MgpuProver:
    """zk_mgpu: the scalar-exchange prover as a pipeline inside the library.  push() hands in this rank's proof of the next
    round, pop() returns this rank's proof of the oldest round; both are collective (same call sequence on every rank)."""

    def __init__(self, ctx, comm, crs=None, qap=None, backend=None):
        self.lib = _lib.load()
        self.comm, self._backend, self._keep = comm, backend, (crs, qap)
        p = C.c_void_p()
        if backend is not None:
            rc = self.lib.zk_mgpu_create_custom(comm.ptr, C.byref(backend), C.byref(p))
        else:
            rc = self.lib.zk_mgpu_create(ctx.ptr, comm.ptr, crs.ptr, qap.ptr, C.byref(p))
        if rc != 0:
            raise ZkError(rc, self.lib.zk_last_error(ctx.ptr).decode() if ctx is not None else "")
        self.ptr = p

    def _check(self, rc):
        if rc != 0:
            raise ZkError(rc, self.lib.zk_mgpu_last_error(self.ptr).decode())

    def push(self, d_weights_ptr, m, r, s):
        r_, rp = _fr(r)
        s_, sp = _fr(s)
        self._check(self.lib.zk_mgpu_push(self.ptr, C.c_void_p(d_weights_ptr), m, rp, sp))

    def pop(self):
        out = np.zeros(PROOF_BYTES, dtype=np.uint8)
        self._check(self.lib.zk_mgpu_pop(self.ptr, out.ctypes.data_as(_lib.u8p)))
        return out.tobytes()

    def prove_stream(self, jobs, ahead=2):
        """jobs: iterable of (d_weights_ptr, m, r, s), this rank's proof of each round; yields this rank's proofs in order with
        `ahead` rounds pushed beyond the one being popped (2 = the pipelined schedule, 0 = one round at a time)."""
        pending = 0
        for j in jobs:
            self.push(*j)
            pending += 1
            if pending > ahead:
                yield self.pop()
                pending -= 1
        while pending:
            yield self.pop()
            pending -= 1

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.zk_mgpu_destroy(self.ptr)
            self.ptr = None


def prove_sharded_abi(ctx, comm, crs, qap, d_weights_ptr, m, r, s):
    """zk_mgpu_prove_sharded: one proof over all ranks (latency form); the same bytes on every rank."""
    r_, rp = _fr(r)
    s_, sp = _fr(s)
    out = np.zeros(PROOF_BYTES, dtype=np.uint8)
    ctx._check(ctx.lib.zk_mgpu_prove_sharded(ctx.ptr, comm.ptr, crs.ptr, qap.ptr, C.c_void_p(d_weights_ptr), m, rp, sp, out.ctypes.data_as(_lib.u8p)))
    return out.tobytes()


# ---- the same protocols driven from Python over torch.distributed (round 1; kept as the fallback transport of bench.py) ----


class GpuProver:
    """partial()/combine() on one device through the C ABI; buffers are torch CUDA tensors."""

    def __init__(self, ctx, crs, qap, d_weights, m):
        import torch
        self.torch = torch
        self.ctx, self.crs, self.qap, self.d_weights, self.m = ctx, crs, qap, d_weights, m

    def new_buffer(self, nbytes):
        return self.torch.zeros(nbytes, dtype=self.torch.uint8, device="cuda")

    def partial(self, rank, world, r, s, out):
        self.ctx.prove_partial(self.crs, self.qap, self.d_weights.data_ptr(), self.m, r, s, rank, world, out.data_ptr())

    def partial_submit(self, rank, world, r, s, out):
        return self.ctx.prove_partial_submit(self.crs, self.qap, self.d_weights.data_ptr(), self.m, r, s, rank, world, out.data_ptr())

    def partial_wait(self, ticket):
        self.ctx.prove_wait(ticket, partial=True)

    def gather_done(self):
        self.torch.cuda.current_stream().synchronize()   # the all-gather runs on torch's stream

    def combine(self, gathered, world, r, s):
        self.gather_done()
        return self.ctx.prove_combine(self.crs, gathered.data_ptr(), world, r, s)


def all_gather_bytes(dist, gathered, part):
    """all-gather of the per-rank byte blobs.  RCCL ("nccl") gathers device tensors directly; with the gloo
    backend (CPU tests, and functional runs of several ranks on one GPU) device tensors are staged through
    the host."""
    if part.is_cuda and dist.get_backend() == "gloo":
        host = gathered.cpu()
        dist.all_gather_into_tensor(host, part.cpu())
        gathered.copy_(host)
    else:
        dist.all_gather_into_tensor(gathered, part)


def prove_sharded(prover, dist, rank, world, r, s, buffers=None):
    """One proof with the inner products sharded over `world` ranks.  Returns the 259 proof bytes
    (identical on every rank)."""
    if buffers is None:
        buffers = (prover.new_buffer(PARTIAL_BYTES), prover.new_buffer(world * PARTIAL_BYTES))
        prover.gather_done()
    part, gathered = buffers
    prover.partial(rank, world, r, s, part)
    if world > 1:
        all_gather_bytes(dist, gathered, part)
    else:
        gathered.copy_(part)
    return prover.combine(gathered, world, r, s)


def prove_sharded_stream(prover, dist, rank, world, jobs, depth=4):
    """Pipelined prove_sharded over a sequence of (r, s) jobs: the partial sums of proof k+1 are enqueued
    before proof k's all-gather and final assembly, so the GPU never idles on the collective or on the
    latency-bound tail.  `depth` proofs are in flight (the C ABI allows ZK_MAX_IN_FLIGHT = 4).  Yields the proof bytes in order."""
    bufs = [(prover.new_buffer(PARTIAL_BYTES), prover.new_buffer(world * PARTIAL_BYTES)) for _ in range(depth)]
    prover.gather_done()   # torch zero-fills the new buffers on its own stream; the library writes them on others
    inflight = []

    def finish(item):
        ticket, (part, gathered), r, s = item
        prover.partial_wait(ticket)
        if world > 1:
            all_gather_bytes(dist, gathered, part)
        else:
            gathered.copy_(part)
        return prover.combine(gathered, world, r, s)

    for k, (r, s) in enumerate(jobs):
        if len(inflight) == depth:
            yield finish(inflight.pop(0))
        buf = bufs[k % depth]
        inflight.append((prover.partial_submit(rank, world, r, s, buf[0]), buf, r, s))
    while inflight:
        yield finish(inflight.pop(0))


# ---- scalar exchange ------------------------------------------------------------------------------
def all_to_all_bytes(dist, out, inp):
    """Equal-split all-to-all of byte tensors (chunk g of `inp` goes to rank g; chunk j of `out` comes from rank j)."""
    if inp.is_cuda and dist.get_backend() == "gloo":
        host = out.cpu()
        dist.all_to_all_single(host, inp.cpu())
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp)


class GpuExchangeProver(GpuProver):
    """The scalar-exchange stages on one device through the C ABI (zk_prove_scalars_submit / zk_prove_msm_submit)."""

    def exchange_buffers(self, world):
        """(send, recv, part_send, part_recv): four byte tensors each way (L, V, U, H scalars) + the partial-sum blobs."""
        elems = self.ctx.prove_exchange_elems(self.qap, world)
        send = [self.new_buffer(32 * e) for e in elems]
        recv = [self.new_buffer(32 * e) for e in elems]
        return send, recv, self.new_buffer(world * PARTIAL_BYTES), self.new_buffer(world * PARTIAL_BYTES)

    def scalars_submit(self, r, s, world, send):
        return self.ctx.prove_scalars_submit(self.crs, self.qap, self.d_weights.data_ptr(), self.m, r, s, world, [t.data_ptr() for t in send])

    def msm_submit(self, sets, rank, world, recv, part_send):
        return self.ctx.prove_msm_submit(self.crs, self.qap, sets, rank, world, [t.data_ptr() for t in recv], part_send.data_ptr())

    def wait(self, ticket):
        self.ctx.prove_wait(ticket, partial=True)

    def comm_done(self):
        self.torch.cuda.current_stream().synchronize()   # the collectives run on torch's stream

    def combine_own(self, part_recv, world, r, s):
        return self.ctx.prove_combine(self.crs, part_recv.data_ptr(), world, r, s)


def prove_exchange_stream(prover, dist, rank, world, jobs):
    """jobs: this rank's (r, s) per round -- in round k every rank owns one proof, so len(jobs) rounds make
    world * len(jobs) proofs.  Yields this rank's proof bytes round by round.  Software pipeline: while the inner
    products of round k run, those of round k+1 are already queued (its scalars were computed during round k-1 and
    exchanged at the start of round k) and the scalars of round k+2 are being computed, so neither the collectives nor
    the host waits leave the GPU idle.  With world == 1 (or dist None) the exchanges
    are local copies."""
    rounds = len(jobs)
    bufs = [prover.exchange_buffers(world) for _ in range(2)]
    prover.comm_done()   # torch zero-fills the new buffers on its own stream; the library writes them on others
    single = world == 1 or dist is None

    def exchange(outs, ins):
        for o, i in zip(outs, ins):
            if single:
                o.copy_(i)
            else:
                all_to_all_bytes(dist, o, i)
        prover.comm_done()

    def scalars(k):
        return prover.scalars_submit(jobs[k][0], jobs[k][1], world, bufs[k % 2][0])

    def inner_products(k, ticket_scalars):
        send, recv, part_send, _ = bufs[k % 2]
        prover.wait(ticket_scalars)
        exchange(recv, send)
        return prover.msm_submit(world, rank, world, recv, part_send)

    if rounds == 0:
        return
    # the scalars run two rounds ahead of the inner products: under the chip-filling accumulations of round k the
    # SpMV / NTT stage of a later round only progresses in the gaps, and the inner products of round k+1 must be
    # queued (sorted) before those of round k end
    t_a = {k: scalars(k) for k in range(min(2, rounds))}
    t_b = {0: inner_products(0, t_a.pop(0))}
    for k in range(rounds):
        if k + 2 < rounds:
            t_a[k + 2] = scalars(k + 2)     # writes the send buffers of round k, whose exchange completed in the previous iteration
        if k + 1 < rounds:
            t_b[k + 1] = inner_products(k + 1, t_a.pop(k + 1))
        _, _, part_send, part_recv = bufs[k % 2]
        prover.wait(t_b.pop(k))
        exchange([part_recv], [part_send])
        yield prover.combine_own(part_recv, world, jobs[k][0], jobs[k][1])
