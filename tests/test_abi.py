"""-m "not gpu": the C-ABI library loads and exports every symbol include/zkgpu.h (the product ABI) and include/zkgpu_measure.h (the
measurement / test entry points) declare, the binding covers exactly that set, the product header carries no measurement switch, and
the product refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("zkgpu.h", "zkgpu_measure.h")):
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


MEASUREMENT_KEYS = ("serialize", "ablate", "msm_fold", "msm_run_entries", "msm_run_whole", "msm_run_fill", "msm_small_lanes", "msm_unchain_lanes",
                    "chain_order", "alt_stream", "tail_stream", "ntt_fuse", "apply_cu_reserve")


def test_product_header_exports_no_measurement_switch():
    """include/zkgpu.h is the product ABI -- no profiling entry point, no test hook, no tuning key, none of the
    closed experiments (lone_graph, g2_affine); those live in include/zkgpu_measure.h or are gone.  The option table of the product
    build (csrc/capi.hip outside #ifdef ZK_MEASURE) accepts none of the measurement keys either."""
    product = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    for banned in ("zk_profile_", "zk_lazy29_batch", "lone_graph", "g2_affine") + tuple('"%s"' % k for k in MEASUREMENT_KEYS):
        assert banned not in product, banned
    only_measure = set(declared_symbols(("zkgpu_measure.h",))) - set(declared_symbols(("zkgpu.h",)))
    assert only_measure == {"zk_profile_reset", "zk_profile_count", "zk_profile_entry", "zk_lazy29_batch"}
    capi = open(os.path.join(ROOT, "zksnark_rs_amd", "csrc", "capi.hip")).read()
    slot = capi[capi.index("static long* option_slot"):capi.index("int zk_set_option")]
    product_part = slot[:slot.index("#ifdef ZK_MEASURE")]
    measure_part = slot[slot.index("#ifdef ZK_MEASURE"):]
    for k in MEASUREMENT_KEYS:
        assert '"%s"' % k not in product_part, k
        if k != "apply_cu_reserve":
            assert '"%s"' % k in measure_part, k
    for gone in ("lone_graph", "g2_affine"):
        assert gone not in capi


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


# ---- the Rust binding against the header -------------------------------------------------------------------
# bindings/rust/src/gpu.rs cannot be compiled here (no rustc), so nothing mechanical kept its extern "C" block and #[repr(C)]
# structs in step with include/zkgpu.h.  Both sides are parsed into the same canonical description -- per function: name, return
# type, and per argument the integer width / signedness or "pointer to <pointee>"; per struct: field names, order and types -- and
# compared.  const-ness is not part of the ABI and is ignored; a parameter added, dropped, reordered or resized on one side fails.
C_INT_TYPES = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "long": "long", "size_t": "usize", "double": "f64",
               "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "char": "char", "void": "void"}
RUST_TYPES = {"c_int": "i32", "c_uint": "u32", "c_long": "long", "std::os::raw::c_long": "long", "usize": "usize", "f64": "f64", "c_double": "f64",
              "u8": "u8", "u32": "u32", "u64": "u64", "i32": "i32", "c_char": "char", "c_void": "void"}


def _camel(snake):                       # zk_qap_sparse_desc -> ZkQapSparseDesc
    return "".join(p.capitalize() for p in snake.split("_"))


def _c_type(decl):
    """'const uint64_t r[4]' / 'zk_ctx** out' / 'size_t m' -> (canonical type, name)"""
    decl = decl.strip()
    ptr = decl.count("*")
    m = re.search(r"\[[^\]]*\]\s*$", decl)
    if m:                                  # an array parameter is a pointer
        ptr += 1
        decl = decl[:m.start()]
    decl = decl.replace("*", " ")
    words = [w for w in decl.split() if w not in ("const", "struct")]
    name = words.pop() if len(words) > 1 else ""
    base = " ".join(words)
    base = C_INT_TYPES.get(base, _camel(base) if base.startswith("zk_") else base)
    return "*" * ptr + base, name


def _rust_type(t):
    t = t.strip()
    ptr = 0
    while True:
        m = re.match(r"\*(const|mut)\s+(.*)$", t)
        if not m:
            break
        ptr += 1
        t = m.group(2).strip()
    return "*" * ptr + RUST_TYPES.get(t, t)


def _split_args(text):
    return [a for a in (x.strip() for x in text.split(",")) if a and a != "void"]


def parse_c_header(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    funcs, structs = {}, {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(zk_[a-z0-9_]+)\s*;", text, flags=re.S):
        fields = []
        for f in m.group(1).split(";"):
            f = f.strip()
            if not f or "(" in f:          # callback tables are not mirrored by the shim
                if "(" in f:
                    fields.append(("fnptr", re.search(r"\(\s*\*\s*([a-z0-9_]+)\s*\)", f).group(1)))
                continue
            ftype, rest = re.match(r"((?:const\s+)?(?:unsigned\s+)?[A-Za-z0-9_]+)\s*(.*)$", f).groups()
            for item in rest.split(","):   # `const uint64_t *a, *b;` declares several
                t, name = _c_type(ftype + " " + item.strip())
                fields.append((t, name))
        structs[_camel(m.group(2))] = fields
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*zk_[a-z0-9_]+\s*;", "", text, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(zk_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body):
        ret, _ = _c_type(m.group(1).strip() + " x")
        funcs[m.group(2)] = (ret, [_c_type(a)[0] for a in _split_args(m.group(3))])
    return funcs, structs


def parse_rust_binding(text):
    text = re.sub(r"//[^\n]*", "", text)
    funcs, structs = {}, {}
    for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn\s+(zk_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", blk.group(1), flags=re.S):
            args = [_rust_type(a.split(":", 1)[1]) for a in _split_args(m.group(2))]
            funcs[m.group(1)] = (_rust_type(m.group(3)) if m.group(3) else "void", args)
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub\s+struct\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in _split_args(m.group(2)):
            name, t = f.split(":", 1)
            fields.append((_rust_type(t), name.replace("pub", "").strip()))
        structs[m.group(1)] = fields
    return funcs, structs


def _binding_sources():
    header = open(os.path.join(ROOT, "include", "zkgpu.h")).read()
    rust = open(os.path.join(ROOT, "bindings", "rust", "src", "gpu.rs")).read()
    return header, rust


def compare_abi(header, rust):
    """list of mismatches between the header's prototypes / structs and the binding's (empty = in step)"""
    cf, cs = parse_c_header(header)
    rf, rs = parse_rust_binding(rust)
    bad = []
    for name, (rret, rargs) in sorted(rf.items()):
        if name not in cf:
            bad.append("%s: bound in gpu.rs, not declared in zkgpu.h" % name)
            continue
        cret, cargs = cf[name]
        if cret != rret:
            bad.append("%s: returns %s in the header, %s in gpu.rs" % (name, cret, rret))
        if len(cargs) != len(rargs):
            bad.append("%s: %d parameters in the header, %d in gpu.rs" % (name, len(cargs), len(rargs)))
            continue
        for i, (a, b) in enumerate(zip(cargs, rargs)):
            if a != b:
                bad.append("%s: parameter %d is %s in the header, %s in gpu.rs" % (name, i, a, b))
    for name, rfields in sorted(rs.items()):
        if rfields and rfields[0][1] == "_p":          # opaque handles
            continue
        if name not in cs:
            bad.append("struct %s: in gpu.rs, not in zkgpu.h" % name)
            continue
        if cs[name] != rfields:
            bad.append("struct %s: fields %s in the header, %s in gpu.rs" % (name, cs[name], rfields))
    return bad, cf, rf, rs


def test_rust_binding_matches_the_header():
    header, rust = _binding_sources()
    bad, cf, rf, rs = compare_abi(header, rust)
    assert not bad, "\n".join(bad)
    assert len(rf) >= 29 and {"zk_prove", "zk_setup", "zk_verify", "zk_mgpu_pop", "zk_crs_upload"} <= set(rf)
    assert {"ZkCrsDesc", "ZkCrsOut", "ZkSparseRows", "ZkQapSparseDesc"} <= set(rs)
    assert len(cf) >= 80                                   # the parser saw the whole header, not a fragment


def test_rust_binding_guard_catches_drift():
    """the guard itself: a parameter added to one side only, a resized integer, a swapped struct field and a dropped one must all fail"""
    header, rust = _binding_sources()
    assert compare_abi(header, rust)[0] == []
    h2 = header.replace("int zk_prove_wait(zk_ctx* ctx, int ticket,", "int zk_prove_wait(zk_ctx* ctx, int ticket, int flags,")
    assert h2 != header and any("zk_prove_wait" in b for b in compare_abi(h2, rust)[0])
    r2 = rust.replace("fn zk_prove_wait(ctx: *mut ZkCtx, ticket: c_int,", "fn zk_prove_wait(ctx: *mut ZkCtx, ticket: usize,")
    assert r2 != rust and any("zk_prove_wait: parameter 1" in b for b in compare_abi(header, r2)[0])
    r3 = rust.replace("n: usize, m: usize, input: usize,\n    alpha_g1", "m: usize, n: usize, input: usize,\n    alpha_g1")
    assert r3 != rust and any("struct ZkCrsDesc" in b for b in compare_abi(header, r3)[0])
    r4 = rust.replace("fn zk_mgpu_pop(p: *mut ZkMgpu, proof_out: *mut u8) -> c_int;", "fn zk_mgpu_pop(p: *mut ZkMgpu) -> c_int;")
    assert r4 != rust and any("zk_mgpu_pop" in b for b in compare_abi(header, r4)[0])
    r5 = rust.replace("fn zk_comm_set_timeout(c: *mut ZkComm, ms: std::os::raw::c_long)", "fn zk_comm_set_timeout(c: *mut ZkComm, ms: c_int)")
    assert r5 != rust and any("zk_comm_set_timeout" in b for b in compare_abi(header, r5)[0])


def test_bench_reads_this_rounds_profile_files():
    """bench.py's roofline fields are read from committed counter files, never typed in: they must be the newest round's (VERDICT r4
    weak 6: a two-round-old micro-benchmark file fed the line) and exist (host code, no GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    newest = max(int(m.group(1)) for m in (re.match(r"r(\d+)_", f) for f in os.listdir(os.path.join(ROOT, "profiles"))) if m)
    for name in [bench.PMC_FILES[20], bench.PMC_FILES[16], bench.PMC_ACC_FILES[20], bench.PMC_ACC_FILES[16], bench.UBENCH_FILE]:
        assert name.startswith("r%d_" % newest), name
        assert os.path.getsize(os.path.join(ROOT, "profiles", name)) > 0, name
    assert bench.ubench_sustained() is not None and bench.pmc_acc("msm_accumulate_g1", 20) is not None
