"""-m gpu: bench.py's contract line at a small size, for N = 1 and for the N = 2 code path (two ranks on ONE
GPU over gloo: a functional check of the scalar-exchange and the window/point-sharded provers, their pipelined drivers
and the replicas leg -- bench.py asserts that all of them give the same proof bytes; RCCL itself needs one GPU per rank and is exercised by the driver's multi-GPU runs)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def last_json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_bench_line_single_gpu():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--log-n", "12"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    d = last_json_line(res.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] > 0 and d["unit"] == "proofs/s"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source_commit"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    # the timed region cycles through distinct (witness, r, s); every proof is compared with a synchronous single-GPU proof of the
    # same inputs made in the untimed set-up
    c = d["config"]
    assert c["witness_sets"] >= 4 and len(set(c["proof_shas"])) == c["witness_sets"] and c["proof_sha"] == c["proof_shas"][0]
    # set 0's proof passed groth16::verify in the untimed set-up (the pairing check shares no code with the prover), and the line
    # carries the steady-state period beside ms_per_step (null when too few proofs were timed to have one)
    assert c["verified"] is True and "steady_state_ms_per_proof" in d
    # whole-proof HBM rate from counters: present with its source, or null (no counter pass for this size) -- never typed in
    assert "hbm_measured_GBps_whole_proof" in d and "hbm_algorithmic_GBps_whole_proof" in d
    if d["hbm_measured_GBps_whole_proof"] is not None:
        assert d["hbm_measured_source"]["file"].startswith("profiles/") and "commit" in d["hbm_measured_source"]
    v = d["roofline"].get("valu")
    if v:       # the sustained issue rates come from the micro-benchmark's committed output
        assert v["peak_measured_source"].startswith("profiles/")


def run_two_ranks(extra, env=None, port_base=29600, backend="gloo"):
    port = port_base + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", backend] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return last_json_line(res.stdout)


@pytest.mark.parametrize("mode,shard,transport", [("exchange", "points", "zk-gloo"), ("shard", "points", "zk-gloo"), ("shard", "windows", "zk-gloo"),
                                                  ("shard", "buckets", "zk-gloo"), ("exchange", "points", "torch"), ("shard", "points", "torch")])
def test_bench_line_two_ranks_on_one_gpu(mode, shard, transport):
    """zk-gloo: the pipeline and collectives inside libzkgpu.so (zk_mgpu_*, zk_comm_* with a caller-supplied gloo transport);
    torch: the round-1 Python driver over torch.distributed.  With the library's pipeline ONE line carries all three legs -- the
    scalar exchange, the window-sharded form (BASELINE config 5) and the replicas -- each with its rate and its byte equality
    against a single-GPU proof of the same inputs."""
    d = run_two_ranks(["--log-n", "12", "--mode", mode, "--shard", shard, "--transport", transport])
    assert KEYS <= set(d)
    if mode == "exchange":
        assert ("inside libzkgpu.so" in d["config"]["parallelism"]) == (transport == "zk-gloo")
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if mode == "shard" else "weak") and d["value"] > 0
    assert d["replicas"]["value"] > 0 and d["replicas"]["scaling"] == "weak" and d["replicas"]["bytes_equal_to_single_gpu_prove"]
    assert "degraded" not in d and d["wait_bound_s"] > 0 and d["rccl_ranks"] == 0    # gloo / torch transports: no RCCL communicator behind the line
    if transport == "zk-gloo":
        for leg, scaling in (("exchange", "weak"), ("window_shard", "strong"), ("bucket_shard", "strong")):
            assert d[leg]["value"] > 0 and d[leg]["ms_per_step"] > 0 and d[leg]["scaling"] == scaling and d[leg]["bytes_equal_to_single_gpu_prove"], leg
        assert d["xgmi_bytes_sent_per_rank_per_round"] > 0 and d["config"]["witness_sets"] >= 4
        primary = d["exchange"] if mode == "exchange" else d["window_shard"] if shard == "windows" else d["bucket_shard"] if shard == "buckets" else d["shard"]
        assert primary["value"] == d["value"]
        # BASELINE config 5 as a top-level object of the N > 1 line: one proof at a time, strong scaling
        c5 = d["config5"]
        assert c5["scaling"] == "strong" and c5["form"] in ("bucket_shard", "window_shard") and c5["value"] == max(d["bucket_shard"]["value"], d["window_shard"]["value"])
        assert c5["window_shard_value"] == d["window_shard"]["value"] and c5["bucket_shard_value"] == d["bucket_shard"]["value"]


def test_bench_falls_back_to_independent_provers():
    """A sharded protocol that fails in the warm-up (injected on both ranks) degrades to replicas and says so."""
    d = run_two_ranks(["--log-n", "10"], env={"ZK_BENCH_TEST_FAIL_EXCHANGE": "1"}, port_base=29900)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "injected failure" in d["degraded"] and d["config"]["parallelism"] == "replicas x2"
    assert "error" in d["exchange"] and d["replicas"]["value"] == d["value"]


def test_bench_survives_a_peer_that_never_arrives():
    """One rank hangs in the warm-up of the exchange (it never makes its collective calls): the other rank's wait is bounded, both
    agree through the rendezvous store, the line reports the time-out and the replicas' rate -- the job ends instead of hanging."""
    d = run_two_ranks(["--log-n", "10", "--timeout", "25"], env={"ZK_BENCH_TEST_HANG_EXCHANGE": "1"}, port_base=30300)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "replicas x2"
    assert "did not complete" in d["degraded"] or "timed out" in d["degraded"] or "another rank" in d["degraded"]
    assert d["replicas"]["bytes_equal_to_single_gpu_prove"]


def test_bench_degrades_when_the_rccl_communicator_cannot_be_created():
    """The driver's own command line (--transport zk over RCCL) with two ranks on ONE GPU: RCCL refuses a communicator whose ranks
    share a device (or never completes it) -- the bootstrap's bounded wait ends in the degraded path, and the line still carries the
    replicas' rate and their byte equality.  What a broken fabric or a missing peer looks like on a real node."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU (two ranks must collide on it)")
    d = run_two_ranks(["--log-n", "10", "--timeout", "40"], port_base=30700, backend="nccl")
    assert d["n_gpus"] == 2 and d["value"] > 0 and "degraded" in d and d["config"]["parallelism"] == "replicas x2"
    assert d["rccl_ranks"] == 0 and d["replicas"]["bytes_equal_to_single_gpu_prove"]
