"""The C++ host API (include/zksnark.hpp) mirrors the reference crate's groth16::{setup, prove, verify},
ASTParser, QAP and FrLocal over the C ABI.  tests/cpp/reference_tests.cpp restates the reference's own
BN254 end-to-end tests (lib.rs:156-190, fr.rs:248-416) against it.

not gpu: the program compiles and links against libzkgpu.so (no device call is made).
gpu:     it runs on the device and every test prints "ok"."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp")
LIBDIR = os.path.join(ROOT, "zksnark_rs_amd")


def build(out_dir):
    exe = os.path.join(str(out_dir), "reference_tests")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", LIBDIR, "-lzkgpu", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_cpp_api_compiles_and_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libzkgpu.so")):
        pytest.skip("libzkgpu.so not built")
    try:
        exe = build(tmp_path)
    except subprocess.CalledProcessError as e:
        pytest.fail("g++ failed:\n" + e.stderr[-3000:])
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_reference_tests_through_cpp_api(tmp_path):
    exe = build(tmp_path)
    env = dict(os.environ, ZK_TEST_TMP=str(tmp_path))
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "zk")], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    for name in ("simple_circuit_test", "single_mult_honest_bn", "bn_encrypt_quad_test", "bn_encrypt_cubic_test",
                 "bn_encrypt_deg_15_test", "root_representation_over_any_roots", "error_behaviour"):
        assert "ok " + name in res.stdout, res.stdout + res.stderr


# ---- the Rust shim's call sequences (bindings/rust/src/gpu.rs), executed from C ----
SHIM_SRC = os.path.join(ROOT, "tests", "cpp", "rust_shim_sequence.c")


def build_shim(out_dir):
    exe = os.path.join(str(out_dir), "rust_shim_sequence")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SHIM_SRC, "-o", exe,
           "-L", LIBDIR, "-lzkgpu", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lamdhip64"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_rust_shim_sequence_compiles_and_links(tmp_path):
    """plain C against include/zkgpu.h: every entry point and struct layout the shim binds exists with that signature"""
    try:
        exe = build_shim(tmp_path)
    except subprocess.CalledProcessError as e:
        pytest.fail("gcc failed:\n" + e.stderr[-3000:])
    assert os.path.exists(exe)
    # the extern block of the shim names nothing the header does not declare
    import re
    shim = open(os.path.join(ROOT, "bindings", "rust", "src", "gpu.rs")).read()
    header = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    for fn in re.findall(r"\bfn (zk_[a-z0-9_]+)\(", shim):
        assert re.search(r"\b%s\s*\(" % fn, header), fn
    # the byte conversions run before the first device call: checked here on the CPU as well
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "zk", "simple.zk")] + g2_generator_packed(), capture_output=True, text=True, timeout=120)
    assert "ok byte_conversions" in res.stdout, res.stdout + res.stderr


def g2_generator_packed():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyref
    (x0, x1), (y0, y1) = pyref.G2_GEN
    return ["%0128x" % (c1 * pyref.Q + c0) for c0, c1 in ((x0, x1), (y0, y1))]     # bn's Fq2 packing


@pytest.mark.gpu
def test_rust_shim_sequence_on_the_gpu(tmp_path):
    exe = build_shim(tmp_path)
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "zk", "simple.zk")] + g2_generator_packed(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    for name in ("byte_conversions", "setup", "prove", "verify", "prove_stream", "from_root_rep", "from_root_rep_integers", "from_root_rep_any", "multi_gpu_world_1"):
        assert "ok " + name in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_one_rank_communicator_through_rccl(tmp_path):
    """ZK_COMM_FORCE_RCCL=1: a one-rank zk_comm is a real RCCL communicator (ncclCommInitRank with one rank, grouped self
    send / recv, all-gather, all-reduce) -- the only way to execute the RCCL calls of csrc/comm.hip on a one-GPU box.  The C
    program's multi-GPU section must still give zk_prove's bytes; then the same from a process that has torch (and its own
    bundled librccl) loaded, as bench.py does."""
    exe = build_shim(tmp_path)
    env = dict(os.environ, ZK_COMM_FORCE_RCCL="1")
    res = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "zk", "simple.zk")] + g2_generator_packed(), capture_output=True, text=True,
                         timeout=600, env=env)
    assert res.returncode == 0 and "ok multi_gpu_world_1" in res.stdout, res.stdout + res.stderr
    code = r"""
import numpy as np, torch
import zksnark_rs_amd as zk
from zksnark_rs_amd.distributed import Comm, MgpuProver
from zksnark_rs_amd.circuits import chain_rows, chain_weights
ctx = zk.Context(0)
comm = Comm(ctx, 0, 1, Comm.unique_id())
a = torch.arange(4096, dtype=torch.int32, device="cuda"); b = torch.zeros_like(a)
torch.cuda.synchronize()
comm.all_to_all(a.data_ptr(), b.data_ptr(), a.numel() * 4); assert torch.equal(a, b)
b.zero_(); torch.cuda.synchronize()
comm.all_gather(a.data_ptr(), b.data_ptr(), a.numel() * 4); assert torch.equal(a, b)
comm.barrier(); assert comm.max_f64(2.5) == 2.5
log_n = 10
m, l, u, v, w = chain_rows(log_n)
rng = zk.SplitMix64(5)
wts = chain_weights(log_n, rng.fr(), [rng.fr() for _ in range(1 << log_n)])
qap = ctx.qap_sparse(log_n, m, l, u, v, w)
crs = ctx.setup(qap, zk.ints_to_limbs([rng.fr() for _ in range(5)]))
r, s = rng.fr(), rng.fr()
want = ctx.prove(crs, qap, wts, r, s)
dw = torch.from_numpy(np.ascontiguousarray(wts).view(np.int64)).cuda(); torch.cuda.synchronize()
mp = MgpuProver(ctx, comm, crs, qap)
got = list(mp.prove_stream([(dw.data_ptr(), m, r, s)] * 4, ahead=2))
assert got == [want] * 4
mp.close()
assert comm.rccl_ranks() == 1
# option comm_cu_reserve: over an RCCL communicator the inner-product streams leave compute units to the collectives; same bytes,
# and zk_mgpu_destroy gives the streams the whole chip back (a lone prove afterwards still works)
ctx.set_option("comm_cu_reserve", 2)
mp = MgpuProver(ctx, comm, crs, qap)
assert list(mp.prove_stream([(dw.data_ptr(), m, r, s)] * 3, ahead=2)) == [want] * 3
mp.close()
assert ctx.prove(crs, qap, wts, r, s) == want
comm.set_timeout(5000)
comm.abort()
try:
    comm.barrier()
    raise SystemExit("an aborted communicator answered a collective")
except zk.ZkError as e:
    assert e.status == zk._lib.ZK_ERR_COMM
comm.close()
print("ok rccl world 1")
"""
    res = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0 and "ok rccl world 1" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
