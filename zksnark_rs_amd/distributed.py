"""Multi-GPU prove: MSM windows sharded over ranks, one all-gather of 768-byte partial sums.

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
