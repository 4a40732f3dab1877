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
from . import PARTIAL_BYTES


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
