"""-m "not gpu": the C-ABI library loads and exports every symbol include/zkgpu.h declares, the
binding covers exactly that set, and the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from zksnark_rs_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert getattr(lib, n) is not None
    assert sorted(_lib.SIGNATURES) == names


def test_strerror_and_null_arguments():
    from zksnark_rs_amd import _lib
    lib = _lib.load()
    assert lib.zk_strerror(0) == b"ok"
    assert b"division" in lib.zk_strerror(_lib.ZK_ERR_DIV_BY_ZERO)
    assert lib.zk_ctx_create(0, None) == _lib.ZK_ERR_ARG
    assert lib.zk_ntt_fr(None, None, 3, 0, 0) == _lib.ZK_ERR_ARG
    lib.zk_ctx_destroy(None)
    lib.zk_qap_free(None)
    lib.zk_crs_free(None)


def test_no_cpu_fallback_without_gpu():
    """Without a visible GPU the context cannot be created; nothing computes on the CPU instead."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible")
    except ImportError:
        pass
    import zksnark_rs_amd as zk
    with pytest.raises(zk.ZkError) as e:
        zk.Context(0)
    assert e.value.status == zk._lib.ZK_ERR_NO_DEVICE


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under zksnark_rs_amd/ or include/ may import, include,
    link or call it."""
    banned = ("oracle_lib", "liboracle", "oracle/", "orc_", "pyref", "namespace orc")
    for base in ("zksnark_rs_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "_build" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cuh", ".h", ".cpp")) or f == "Makefile":
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    for b in banned:
                        assert b not in text, (os.path.join(dp, f), b)


def test_bench_mirrors_the_window_rule():
    """bench.py counts additions with its own copy of the window rule (msm_window); it must be the library's (host code, no GPU)"""
    import importlib.util
    from zksnark_rs_amd import _lib
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lib = _lib.load()
    for count in [1, 2, 15, 16, 100, 2047, 2048, 16383, 16384, 65535, 65536, 131071, 131072, 1 << 20, (1 << 21) - 9, (1 << 21) - 8, (1 << 21) - 1,
                  1 << 21, 1 << 22, 1 << 24]:
        assert bench.msm_window(count) == lib.zk_msm_auto_window(count), count
        assert bench.msm_window(count, 0, True) == lib.zk_msm_auto_window_g2(count), count
    assert bench.msm_window(1 << 21, 2017) == 20 and bench.msm_window(1 << 20, 2017) == 17 and bench.msm_window(1 << 20, 172018, True) == 17
