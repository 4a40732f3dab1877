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
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--log-n", "12"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    d = last_json_line(res.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["unit"] == "proofs/s"
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])


@pytest.mark.parametrize("mode,shard,transport", [("exchange", "points", "zk-gloo"), ("shard", "points", "zk-gloo"), ("shard", "windows", "zk-gloo"),
                                                  ("exchange", "points", "torch"), ("shard", "points", "torch")])
def test_bench_line_two_ranks_on_one_gpu(mode, shard, transport):
    """zk-gloo: the pipeline and collectives inside libzkgpu.so (zk_mgpu_*, zk_comm_* with a caller-supplied gloo transport);
    torch: the round-1 Python driver over torch.distributed"""
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--log-n", "12", "--backend", "gloo", "--mode", mode, "--shard", shard, "--transport", transport]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    d = last_json_line(res.stdout)
    assert KEYS <= set(d)
    if mode == "exchange":
        assert ("inside libzkgpu.so" in d["config"]["parallelism"]) == (transport == "zk-gloo")
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if mode == "shard" else "weak") and d["value"] > 0
    assert d["replicas"]["value"] > 0 and d["replicas"]["scaling"] == "weak"


def test_bench_falls_back_to_independent_provers():
    """A sharded protocol that fails in the warm-up (injected on both ranks) degrades to replicas and says so."""
    port = 29900 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--log-n", "10", "--backend", "gloo"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, ZK_BENCH_TEST_FAIL_EXCHANGE="1"))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    d = last_json_line(res.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert "injected failure" in d["degraded"] and d["config"]["parallelism"] == "replicas x2" and "replicas" not in d
