"""The reference's .zk front end behind the C ABI (host code, as in the reference).

    ASTParser::try_parse      circuit/mod.rs:224-527   -> Circuit(code)
    circuit::weights          circuit/mod.rs:529-637   -> Circuit.weights(inputs)
    QAP::from(root_rep)       fr.rs:140-173            -> Circuit.qap(ctx)   (Lagrange interpolation on the GPU)
"""
import ctypes as C

import numpy as np

from . import _lib, ZkError, ints_to_limbs, Qap


class ParseErr(ValueError):
    """ParseErr::SyntaxErr / ParseErr::StructureErr (circuit/ast.rs:290-293)."""


class Circuit:
    def __init__(self, code):
        self.lib = _lib.load()
        p = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.lib.zk_circuit_parse(code.encode(), C.byref(p), err, len(err))
        if rc != 0:
            raise ParseErr(err.value.decode())
        self.ptr = p
        m, n, l, n_in = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.lib.zk_circuit_dims(p, C.byref(m), C.byref(n), C.byref(l), C.byref(n_in))
        self.m, self.n, self.input, self.n_in = m.value, n.value, l.value, n_in.value

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.zk_circuit_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rows(self, which):
        """DummyRep rows of u (0), v (1) or w (2): (ptr[m+1], gate[nnz] 0-based, val[nnz,4])."""
        nnz = C.c_size_t()
        self.lib.zk_circuit_rows(self.ptr, which, None, None, None, C.byref(nnz))
        ptr = np.zeros(self.m + 1, np.uint64)
        gate = np.zeros(max(nnz.value, 1), np.uint32)
        val = np.zeros((max(nnz.value, 1), 4), np.uint64)
        self.lib.zk_circuit_rows(self.ptr, which, ptr.ctypes.data_as(_lib.u64p), gate.ctypes.data_as(_lib.u32p),
                                 val.ctypes.data_as(_lib.u64p), C.byref(nnz))
        return ptr, gate[:nnz.value], val[:nnz.value]

    def weights(self, inputs):
        """inputs: ints or (n_in, 4) limbs in `in` order -> (m, 4) witness, [1] first."""
        a = ints_to_limbs(list(inputs)) if not isinstance(inputs, np.ndarray) else np.ascontiguousarray(inputs, dtype=np.uint64)
        out = np.zeros((self.m, 4), np.uint64)
        rc = self.lib.zk_circuit_weights(self.ptr, a.ctypes.data_as(_lib.u64p), a.shape[0], out.ctypes.data_as(_lib.u64p), self.m)
        if rc != 0:
            raise ParseErr(self.lib.zk_circuit_last_error(self.ptr).decode())
        return out

    def qap(self, ctx):
        p = C.c_void_p()
        ctx._check(self.lib.zk_circuit_qap(ctx.ptr, self.ptr, C.byref(p)))
        q = Qap(ctx, p, self.lib.zk_qap_free)
        q.n, q.m, q.input, q.dense = self.n, self.m, self.input, True
        return q

    def qap_sparse(self, ctx):
        """The same QAP kept as rows over the integer roots 1..n (no interpolation, any size)."""
        p = C.c_void_p()
        ctx._check(self.lib.zk_circuit_qap_sparse(ctx.ptr, self.ptr, C.byref(p)))
        q = Qap(ctx, p, self.lib.zk_qap_free)
        q.n, q.m, q.input, q.dense, q.roots = self.n, self.m, self.input, False, "integers"
        return q


def qap_download_dense(ctx, qap):
    u = np.zeros((qap.m, qap.n, 4), np.uint64)
    v = np.zeros_like(u)
    w = np.zeros_like(u)
    t = np.zeros((qap.n + 1, 4), np.uint64)
    ctx._check(ctx.lib.zk_qap_download_dense(ctx.ptr, qap.ptr, u.ctypes.data_as(_lib.u64p), v.ctypes.data_as(_lib.u64p),
                                             w.ctypes.data_as(_lib.u64p), t.ctypes.data_as(_lib.u64p)))
    return u, v, w, t
