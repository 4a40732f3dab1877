"""pytest configuration: the `gpu` marker and shared fixtures.

`-m "not gpu"`: oracle vs the reference's golden vectors, host logic, ABI surface (no GPU needed).
`-m gpu`     : parity of the HIP path (through the C ABI) against the oracle.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_build():
    """Builds oracle/_build/{oracle_kats,liboracle.so} with g++ (test infrastructure)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "oracle", "_build")


@pytest.fixture(scope="session")
def orc(oracle_build):
    import oracle_lib
    return oracle_lib.load(os.path.join(oracle_build, "liboracle.so"))


@pytest.fixture(scope="session")
def ctx():
    import zksnark_rs_amd as zk
    c = zk.Context(0)      # raises if libzkgpu.so is missing or no GPU is visible: no CPU fallback
    yield c
    c.close()
