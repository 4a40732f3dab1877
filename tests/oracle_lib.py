"""ctypes binding of oracle/_build/liboracle.so (TEST INFRASTRUCTURE -- the CPU oracle).

Mirrors the product binding (zksnark_rs_amd/__init__.py) so parity tests read the same on both
sides.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

from zksnark_rs_amd import _lib as L
from zksnark_rs_amd import fr_to_limbs

u64p, u32p, u8p = L.u64p, L.u32p, L.u8p


def _p(a):
    return a.ctypes.data_as(u64p)


class Oracle:
    def __init__(self, path):
        self.lib = C.CDLL(path)
        self.lib.orc_time_prove_sparse.restype = C.c_double
        self.lib.orc_time_prove_sparse_mt.restype = C.c_double

    @staticmethod
    def _chk(rc):
        if rc != 0:
            raise RuntimeError("oracle status %d" % rc)

    def _batch(self, fn, op, a, b):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64).reshape(-1, 4))
        b = a if b is None else np.ascontiguousarray(np.asarray(b, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros_like(a)
        rc = fn(op, _p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
        return rc, out

    def fr_batch(self, op, a, b=None):
        return self._batch(self.lib.orc_fr_batch, {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], a, b)

    def fq_batch(self, op, a, b=None):
        return self._batch(self.lib.orc_fq_batch, {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], a, b)

    def _pt2(self, fn, words, a, b, bwords):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64).reshape(-1, words))
        b = np.ascontiguousarray(np.asarray(b, dtype=np.uint64).reshape(-1, bwords))
        out = np.zeros_like(a)
        self._chk(fn(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0])))
        return out

    def g1_mul_batch(self, p, s): return self._pt2(self.lib.orc_g1_mul_batch, 8, p, s, 4)
    def g2_mul_batch(self, p, s): return self._pt2(self.lib.orc_g2_mul_batch, 16, p, s, 4)
    def g1_add_batch(self, a, b): return self._pt2(self.lib.orc_g1_add_batch, 8, a, b, 8)
    def g2_add_batch(self, a, b): return self._pt2(self.lib.orc_g2_add_batch, 16, a, b, 16)

    def enc_base_g1(self):
        out = np.zeros(8, np.uint64); self.lib.orc_enc_base_g1(_p(out)); return out

    def enc_base_g2(self):
        out = np.zeros(16, np.uint64); self.lib.orc_enc_base_g2(_p(out)); return out

    def root_of_unity(self, log_n):
        out = np.zeros(4, np.uint64); self.lib.orc_root_of_unity(C.c_uint(log_n), _p(out)); return out

    def dft_fr(self, data, root, inverse=False):
        a = np.ascontiguousarray(np.asarray(data, dtype=np.uint64).reshape(-1, 4))
        r = np.ascontiguousarray(root, dtype=np.uint64)
        out = np.zeros_like(a)
        self._chk(self.lib.orc_dft_fr(_p(a), C.c_size_t(a.shape[0]), _p(r), int(inverse), _p(out)))
        return out

    def ntt_fr(self, data, inverse=False, coset=False):
        a = np.array(data, dtype=np.uint64, order="C").reshape(-1, 4)
        log_n = a.shape[0].bit_length() - 1
        self._chk(self.lib.orc_ntt_fr(_p(a), C.c_uint(log_n), int(inverse), int(coset)))
        return a

    def _msm(self, fn, words, pts, sc, window_bits):
        p = np.ascontiguousarray(np.asarray(pts, dtype=np.uint64).reshape(-1, words))
        s = np.ascontiguousarray(np.asarray(sc, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros(words, np.uint64)
        self._chk(fn(_p(p), _p(s), C.c_size_t(p.shape[0]), int(window_bits), _p(out)))
        return out

    def msm_g1(self, pts, sc, window_bits=0): return self._msm(self.lib.orc_msm_g1, 8, pts, sc, window_bits)
    def msm_g2(self, pts, sc, window_bits=0): return self._msm(self.lib.orc_msm_g2, 16, pts, sc, window_bits)

    # ---- protocol ----
    @staticmethod
    def _crs_out(arrs):
        return L.CrsOut(**{k: _p(v) for k, v in arrs.items()})

    def setup_sparse(self, desc, trapdoor, n, m, input, faithful):
        from zksnark_rs_amd import Context
        arrs = Context.crs_arrays(n, m, input)
        td = np.ascontiguousarray(trapdoor, dtype=np.uint64)
        out = self._crs_out(arrs)
        self._chk(self.lib.orc_setup_sparse(C.byref(desc), _p(td), int(faithful), C.byref(out)))
        return arrs

    def setup_dense(self, u, v, w, t, input, trapdoor):
        from zksnark_rs_amd import Context
        m, n = u.shape[0], u.shape[1]
        arrs = Context.crs_arrays(n, m, input)
        td = np.ascontiguousarray(trapdoor, dtype=np.uint64)
        out = self._crs_out(arrs)
        self._chk(self.lib.orc_setup_dense(_p(u), _p(v), _p(w), _p(t), C.c_size_t(m), C.c_size_t(n), C.c_size_t(input), _p(td), C.byref(out)))
        return arrs

    def prove_sparse(self, desc, crs_desc, weights, r, s, faithful):
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros(259, np.uint8)
        self._chk(self.lib.orc_prove_sparse(C.byref(desc), C.byref(crs_desc), _p(w), C.c_size_t(w.shape[0]),
                                            _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), int(faithful), out.ctypes.data_as(u8p)))
        return out.tobytes()

    def time_prove_sparse(self, desc, crs_desc, weights, r, s, faithful, reps=1):
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros(259, np.uint8)
        sec = self.lib.orc_time_prove_sparse(C.byref(desc), C.byref(crs_desc), _p(w), C.c_size_t(w.shape[0]),
                                             _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), int(faithful), int(reps), out.ctypes.data_as(u8p))
        return sec, out.tobytes()

    def unit_costs(self):
        """(Fr multiply-add, Fr inversion, G1 scalar mul, G2 scalar mul) in seconds on one host thread."""
        out = (C.c_double * 4)()
        self.lib.orc_unit_costs(out)
        return tuple(out)

    def time_prove_sparse_mt(self, desc, crs_desc, weights, r, s, threads, reps=1):
        """NTT + Pippenger path on `threads` host threads; returns (seconds per proof, proof bytes)."""
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros(259, np.uint8)
        sec = self.lib.orc_time_prove_sparse_mt(C.byref(desc), C.byref(crs_desc), _p(w), C.c_size_t(w.shape[0]),
                                                _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), int(threads), int(reps), out.ctypes.data_as(u8p))
        return sec, out.tobytes()

    def prove_dense(self, u, v, w, t, input, crs_desc, weights, r, s):
        m, n = u.shape[0], u.shape[1]
        wt = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros(259, np.uint8)
        rc = self.lib.orc_prove_dense(_p(u), _p(v), _p(w), _p(t), C.c_size_t(m), C.c_size_t(n), C.c_size_t(input), C.byref(crs_desc),
                                      _p(wt), C.c_size_t(wt.shape[0]), _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), out.ctypes.data_as(u8p))
        if rc != 0:
            return rc
        return out.tobytes()

    def trapdoor_proof_sparse(self, desc, trapdoor, weights, r, s):
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        td = np.ascontiguousarray(trapdoor, dtype=np.uint64)
        out = np.zeros(259, np.uint8)
        self._chk(self.lib.orc_trapdoor_proof_sparse(C.byref(desc), _p(td), _p(w), C.c_size_t(w.shape[0]),
                                                     _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), out.ctypes.data_as(u8p)))
        return out.tobytes()

    def trapdoor_proof_integers(self, desc, n, trapdoor, weights, r, s):
        """closed-form proof for sparse rows over the roots 1..n (gate index g = root g + 1)"""
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        td = np.ascontiguousarray(trapdoor, dtype=np.uint64)
        out = np.zeros(259, np.uint8)
        self._chk(self.lib.orc_trapdoor_proof_integers(C.byref(desc), C.c_size_t(n), _p(td), _p(w), C.c_size_t(w.shape[0]),
                                                       _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), out.ctypes.data_as(u8p)))
        return out.tobytes()

    def trapdoor_proof_dense(self, u, v, w, t, input, trapdoor, weights, r, s):
        m, n = u.shape[0], u.shape[1]
        wt = np.ascontiguousarray(np.asarray(weights, dtype=np.uint64).reshape(-1, 4))
        td = np.ascontiguousarray(trapdoor, dtype=np.uint64)
        out = np.zeros(259, np.uint8)
        self._chk(self.lib.orc_trapdoor_proof_dense(_p(u), _p(v), _p(w), _p(t), C.c_size_t(m), C.c_size_t(n), C.c_size_t(input), _p(td),
                                                    _p(wt), C.c_size_t(wt.shape[0]), _p(fr_to_limbs(r)), _p(fr_to_limbs(s)), out.ctypes.data_as(u8p)))
        return out.tobytes()

    # ---- .zk front end ----
    def zk_qap_dense(self, code):
        m, n, l, n_in = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._chk(self.lib.orc_zk_dims(code.encode(), C.byref(m), C.byref(n), C.byref(l), C.byref(n_in)))
        m, n, l, n_in = m.value, n.value, l.value, n_in.value
        u = np.zeros((m, n, 4), np.uint64); v = np.zeros_like(u); w = np.zeros_like(u); t = np.zeros((n + 1, 4), np.uint64)
        self._chk(self.lib.orc_zk_qap_dense(code.encode(), _p(u), _p(v), _p(w), _p(t)))
        return dict(u=u, v=v, w=w, t=t, m=m, n=n, input=l, n_in=n_in)

    def zk_weights(self, code, inputs, m):
        a = np.ascontiguousarray(np.asarray(inputs, dtype=np.uint64).reshape(-1, 4))
        out = np.zeros((m, 4), np.uint64)
        self._chk(self.lib.orc_zk_weights(code.encode(), _p(a), C.c_size_t(a.shape[0]), _p(out), C.c_size_t(m)))
        return out


def load(path=None):
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "liboracle.so")
    return Oracle(path)
