"""ctypes binding of libzkgpu.so (the C ABI in include/zkgpu.h).

There is no CPU fallback: if the HIP extension is missing or no GPU is visible this module
raises instead of silently computing something else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKGPU_LIB") or os.path.join(_HERE, "libzkgpu.so")   # ZKGPU_LIB: experiment builds only

ZK_OK = 0
ZK_ERR_ARG, ZK_ERR_HIP, ZK_ERR_NO_DEVICE, ZK_ERR_SIZE, ZK_ERR_DIV_BY_ZERO, ZK_ERR_RANGE, ZK_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
ZK_ERR_IO, ZK_ERR_COMM = -8, -9
COMM_ID_BYTES = 128
PROOF_BYTES = 259
PARTIAL_BYTES = 768
MAX_IN_FLIGHT = 4      # ZK_MAX_IN_FLIGHT
MAX_BATCH = 64          # ZK_MAX_BATCH

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class SparseRows(C.Structure):
    _fields_ = [("ptr", u64p), ("gate", u32p), ("val", u64p)]


class QapSparseDesc(C.Structure):
    _fields_ = [("log_n", C.c_uint), ("m", C.c_size_t), ("input", C.c_size_t),
                ("u", SparseRows), ("v", SparseRows), ("w", SparseRows)]


class CrsDesc(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("input", C.c_size_t),
                ("alpha_g1", u64p), ("beta_g1", u64p), ("delta_g1", u64p),
                ("xi_g1", u64p), ("sum_gamma_g1", u64p), ("sum_delta_g1", u64p), ("xi_t_g1", u64p),
                ("beta_g2", u64p), ("gamma_g2", u64p), ("delta_g2", u64p), ("xi_g2", u64p)]


class CrsOut(C.Structure):
    _fields_ = [("alpha_g1", u64p), ("beta_g1", u64p), ("delta_g1", u64p), ("xi_g1", u64p),
                ("sum_gamma_g1", u64p), ("sum_delta_g1", u64p), ("xi_t_g1", u64p),
                ("beta_g2", u64p), ("gamma_g2", u64p), ("delta_g2", u64p), ("xi_g2", u64p)]


# zk_comm_ops / zk_mgpu_backend: tables of C callbacks (tests plug gloo and CPU stand-ins in here)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
MAXF64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double))


class CommOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_to_all", A2A_FN), ("all_gather", A2A_FN), ("barrier", BARRIER_FN), ("max_f64", MAXF64_FN)]


ELEMS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_size_t))
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
SCALARS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int))
MSM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_int))
WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
COMBINE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, u64p, u64p, u8p)


class MgpuBackend(C.Structure):
    _fields_ = [("user", C.c_void_p), ("elems", ELEMS_FN), ("alloc", ALLOC_FN), ("free", FREE_FN), ("scalars_submit", SCALARS_FN),
                ("msm_submit", MSM_FN), ("wait", WAIT_FN), ("combine", COMBINE_FN)]


# every symbol include/zkgpu.h declares: (restype, argtypes)
SIGNATURES = {
    "zk_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "zk_ctx_destroy": (None, [C.c_void_p]),
    "zk_strerror": (C.c_char_p, [C.c_int]),
    "zk_last_error": (C.c_char_p, [C.c_void_p]),
    "zk_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_long]),
    "zk_get_option": (C.c_long, [C.c_void_p, C.c_char_p]),
    "zk_ntt_fr": (C.c_int, [C.c_void_p, u64p, C.c_uint, C.c_int, C.c_int]),
    "zk_interpolate_fr": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, u64p]),
    "zk_msm_g1": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, C.c_int, u64p]),
    "zk_msm_g2": (C.c_int, [C.c_void_p, u64p, u64p, C.c_size_t, C.c_int, u64p]),
    "zk_lazy29_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i32p, i32p, i32p, i32p, C.c_size_t, u64p, i32p]),
    "zk_fr_batch": (C.c_int, [C.c_void_p, C.c_int, u64p, u64p, u64p, C.c_size_t]),
    "zk_fq_batch": (C.c_int, [C.c_void_p, C.c_int, u64p, u64p, u64p, C.c_size_t]),
    "zk_g1_mul_batch": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_size_t]),
    "zk_g2_mul_batch": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_size_t]),
    "zk_g1_add_batch": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_size_t]),
    "zk_g2_add_batch": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.c_size_t]),
    "zk_qap_upload_sparse": (C.c_int, [C.c_void_p, C.POINTER(QapSparseDesc), C.POINTER(C.c_void_p)]),
    "zk_qap_upload_sparse_integers": (C.c_int, [C.c_void_p, C.POINTER(QapSparseDesc), C.c_size_t, C.POINTER(C.c_void_p)]),
    "zk_qap_upload_sparse_roots": (C.c_int, [C.c_void_p, C.POINTER(QapSparseDesc), u64p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "zk_qap_upload_dense": (C.c_int, [C.c_void_p, u64p, u64p, u64p, u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "zk_qap_free": (None, [C.c_void_p]),
    "zk_qap_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "zk_qap_kind": (C.c_int, [C.c_void_p]),
    "zk_qap_weighted_sum": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_int, u64p]),
    "zk_qap_download_dense": (C.c_int, [C.c_void_p, C.c_void_p, u64p, u64p, u64p, u64p]),
    "zk_circuit_parse": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "zk_circuit_free": (None, [C.c_void_p]),
    "zk_circuit_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "zk_circuit_rows": (C.c_int, [C.c_void_p, C.c_int, u64p, u32p, u64p, C.POINTER(C.c_size_t)]),
    "zk_circuit_weights": (C.c_int, [C.c_void_p, u64p, C.c_size_t, u64p, C.c_size_t]),
    "zk_circuit_last_error": (C.c_char_p, [C.c_void_p]),
    "zk_msm_auto_window": (C.c_int, [C.c_size_t]),
    "zk_msm_auto_window_g2": (C.c_int, [C.c_size_t]),
    "zk_circuit_qap": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "zk_circuit_qap_sparse": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "zk_crs_upload": (C.c_int, [C.c_void_p, C.POINTER(CrsDesc), C.POINTER(C.c_void_p)]),
    "zk_setup": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.POINTER(C.c_void_p)]),
    "zk_crs_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "zk_crs_download": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CrsOut)]),
    "zk_crs_free": (None, [C.c_void_p]),
    "zk_prove": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64p, C.c_size_t, u64p, u64p, u8p]),
    "zk_crs_save": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p]),
    "zk_crs_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "zk_prove_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, u8p]),
    "zk_prove_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.POINTER(C.c_int)]),
    "zk_prove_submit_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.POINTER(C.c_int)]),
    "zk_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "zk_host_free": (None, [C.c_void_p]),
    "zk_prove_wait": (C.c_int, [C.c_void_p, C.c_int, u8p]),
    "zk_prove_partial_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.c_int, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_int)]),
    "zk_prove_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.c_int, C.c_int, C.c_void_p]),
    "zk_prove_batch_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), u64p, u64p,
                                        C.POINTER(C.c_int)]),
    "zk_prove_batch_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_int, u8p]),
    "zk_prove_exchange_elems": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    "zk_prove_scalars_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "zk_prove_scalars_submit_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "zk_prove_msm_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "zk_prove_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, u64p, u64p, u8p]),
    "zk_qap_save": (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p]),
    "zk_qap_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "zk_proof_save": (C.c_int, [u8p, C.c_char_p]),
    "zk_proof_load": (C.c_int, [C.c_char_p, u8p]),
    "zk_device_count": (C.c_int, []),
    "zk_comm_unique_id": (C.c_int, [u8p]),
    "zk_comm_init": (C.c_int, [C.c_void_p, u8p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_comm_init_custom": (C.c_int, [C.c_void_p, C.POINTER(CommOps), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "zk_comm_destroy": (None, [C.c_void_p]),
    "zk_comm_rank": (C.c_int, [C.c_void_p]),
    "zk_comm_world": (C.c_int, [C.c_void_p]),
    "zk_comm_rccl_ranks": (C.c_int, [C.c_void_p]),
    "zk_comm_set_timeout": (C.c_int, [C.c_void_p, C.c_long]),
    "zk_comm_abort": (C.c_int, [C.c_void_p]),
    "zk_comm_barrier": (C.c_int, [C.c_void_p]),
    "zk_comm_max_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "zk_comm_all_to_all": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zk_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "zk_mgpu_prove_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p, u8p]),
    "zk_mgpu_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "zk_mgpu_create_custom": (C.c_int, [C.c_void_p, C.POINTER(MgpuBackend), C.POINTER(C.c_void_p)]),
    "zk_mgpu_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p]),
    "zk_mgpu_push_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, u64p, u64p]),
    "zk_mgpu_pop": (C.c_int, [C.c_void_p, u8p]),
    "zk_mgpu_destroy": (None, [C.c_void_p]),
    "zk_mgpu_last_error": (C.c_char_p, [C.c_void_p]),
    "zk_verify": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_size_t, u8p, C.POINTER(C.c_int)]),
    "zk_pairing": (C.c_int, [u64p, u64p, u64p]),
    "zk_profile_reset": (C.c_int, [C.c_void_p]),
    "zk_profile_count": (C.c_int, [C.c_void_p]),
    "zk_profile_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
}

_lib = None


def load():
    """Loads libzkgpu.so and types every entry point.  Raises if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "zksnark_rs_amd: HIP extension %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is deliberately no CPU fallback)" % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME libamdhip64.so.7).  Two HIP runtimes in
    # one process cannot both see the GPU, so when torch is installed let it load first: libzkgpu.so
    # then binds to the runtime that is already resident (device memory / streams / RCCL interop).
    if os.environ.get("ZKGPU_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
