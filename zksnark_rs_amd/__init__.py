"""zksnark_rs_amd -- MI355X-native Groth16 prover (hot path of republicprotocol/zksnark-rs).

Python host-side binding over the C ABI (include/zkgpu.h -> libzkgpu.so).  All compute is in the
HIP kernels; this package only marshals numpy buffers / device pointers.  Element conventions are
those of the C ABI: Fr/Fq = 4 little-endian uint64 limbs (canonical integers), G1 affine = 8
limbs, G2 affine = 16 limbs, infinity = all zero.

  ctx  = Context()                                   # one HIP device
  qap  = ctx.qap_sparse(log_n, m, l, u, v, w)        # QAP::from(root_rep) on roots w^j
  crs  = ctx.setup(qap, trapdoor)                    # groth16::setup      (mod.rs:134)
  pf   = ctx.prove(crs, qap, weights, r, s)          # groth16::prove      (mod.rs:213) -> 259 bytes

`zksnark_rs_amd.groth16` mirrors the reference's function names and argument order.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PROOF_BYTES, PARTIAL_BYTES, MAX_IN_FLIGHT, MAX_BATCH

R_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583

__all__ = ["Context", "ZkError", "fr_to_limbs", "limbs_to_int", "ints_to_limbs", "limbs_to_ints",
           "PROOF_BYTES", "PARTIAL_BYTES", "MAX_IN_FLIGHT", "MAX_BATCH", "R_MODULUS", "Q_MODULUS", "SplitMix64", "pairing", "proof_save", "proof_load"]


class ZkError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = _lib.load().zk_strerror(status).decode()
        super().__init__("zkgpu status %d (%s)%s" % (status, msg, ": " + detail if detail else ""))


# ---- limb helpers ---------------------------------------------------------------------------
def fr_to_limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def ints_to_limbs(xs, words=4):
    """list of ints (or nested tuples flattened by the caller) -> (len, words) uint64 array"""
    out = np.empty((len(xs), words), dtype=np.uint64)
    mask = 0xFFFFFFFFFFFFFFFF
    for i, x in enumerate(xs):
        for k in range(words):
            out[i, k] = (x >> (64 * k)) & mask
    return out


def limbs_to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(v) << (64 * i) for i, v in enumerate(a))


def limbs_to_ints(a, words=4):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, words)
    return [sum(int(v) << (64 * i) for i, v in enumerate(row)) for row in a]


class SplitMix64:
    """Deterministic stream shared by tests, bench and the oracle (seeded synthetic inputs)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        while True:
            l = [self.next() for _ in range(4)]
            l[3] &= (1 << 62) - 1
            v = l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)
            if 0 < v < R_MODULUS:
                return v


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(_lib.u64p)


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(_lib.u32p)


def pairing(g1, g2):
    """EllipticEncryptable::pairing (fr.rs:120-122) on the host: (8,) and (16,) limb arrays -> 12 Fq ints."""
    lib = _lib.load()
    a, ap = _u64(np.asarray(g1).reshape(8))
    b, bp = _u64(np.asarray(g2).reshape(16))
    out = np.zeros(48, dtype=np.uint64)
    rc = lib.zk_pairing(ap, bp, out.ctypes.data_as(_lib.u64p))
    if rc != 0:
        raise ZkError(rc)
    return limbs_to_ints(out)


def proof_save(proof, path):
    """zk_proof_save: versioned proof file ("ZKPRFv1" | 259 canonical bytes | checksum)."""
    buf = (C.c_uint8 * PROOF_BYTES).from_buffer_copy(proof)
    rc = _lib.load().zk_proof_save(buf, str(path).encode())
    if rc != 0:
        raise ZkError(rc)


def proof_load(path):
    buf = (C.c_uint8 * PROOF_BYTES)()
    rc = _lib.load().zk_proof_load(str(path).encode(), buf)
    if rc != 0:
        raise ZkError(rc)
    return bytes(buf)


class _Handle:
    def __init__(self, ctx, ptr, free):
        self.ctx, self.ptr, self._free = ctx, ptr, free

    def close(self):
        if self.ptr:
            self._free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Qap(_Handle):
    """Device-resident QAP<CoefficientPoly<FrLocal>> (groth16/mod.rs:60-67)."""
    n = m = input = 0
    dense = False


class Crs(_Handle):
    """Device-resident (SigmaG1, SigmaG2) (groth16/mod.rs:105-121)."""
    n = m = input = 0


class Context:
    def __init__(self, device=0):
        self.lib = _lib.load()
        p = C.c_void_p()
        rc = self.lib.zk_ctx_create(device, C.byref(p))
        if rc != 0:
            raise ZkError(rc)
        self.ptr = p
        self.device = device

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.zk_ctx_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise ZkError(rc, self.lib.zk_last_error(self.ptr).decode())

    def set_option(self, key, value):
        self._check(self.lib.zk_set_option(self.ptr, key.encode(), int(value)))

    def get_option(self, key):
        """Value of an option, -1 for a key this build does not know ("measure_build": 1 in a ZK_MEASURE build of the library)."""
        return int(self.lib.zk_get_option(self.ptr, key.encode()))

    # ---- building blocks ----
    def ntt_fr(self, data, inverse=False, coset=False):
        """field::dft / field::idft (field/mod.rs:508-537) on a (2^k, 4) uint64 array; returns a new array."""
        a = np.array(data, dtype=np.uint64, order="C").reshape(-1, 4)
        n = a.shape[0]
        log_n = n.bit_length() - 1
        if n == 0 or (1 << log_n) != n:
            raise ValueError("length must be a power of two")
        self._check(self.lib.zk_ntt_fr(self.ptr, a.ctypes.data_as(_lib.u64p), log_n, int(inverse), int(coset)))
        return a

    def interpolate_fr(self, roots, values):
        """coefficients of the polynomial of degree < n through (roots[k], values[k]) -- zk_interpolate_fr"""
        r = np.ascontiguousarray(np.asarray(roots, dtype=np.uint64).reshape(-1, 4))
        v = np.ascontiguousarray(np.asarray(values, dtype=np.uint64).reshape(-1, 4))
        if r.shape != v.shape:
            raise ValueError("roots and values differ in length")
        out = np.zeros_like(r)
        self._check(self.lib.zk_interpolate_fr(self.ptr, r.ctypes.data_as(_lib.u64p), v.ctypes.data_as(_lib.u64p), r.shape[0], out.ctypes.data_as(_lib.u64p)))
        return out

    def _msm(self, fn, words, points, scalars, window_bits):
        p, pp = _u64(np.asarray(points).reshape(-1, words))
        s, sp = _u64(np.asarray(scalars).reshape(-1, 4))
        if p.shape[0] != s.shape[0]:
            raise ValueError("points / scalars length mismatch")
        out = np.zeros(words, dtype=np.uint64)
        self._check(fn(self.ptr, pp, sp, p.shape[0], int(window_bits), out.ctypes.data_as(_lib.u64p)))
        return out

    def msm_g1(self, points, scalars, window_bits=0):
        return self._msm(self.lib.zk_msm_g1, 8, points, scalars, window_bits)

    def msm_g2(self, points, scalars, window_bits=0):
        return self._msm(self.lib.zk_msm_g2, 16, points, scalars, window_bits)

    def _batch(self, fn, op, a, b):
        a, ap = _u64(np.asarray(a).reshape(-1, 4))
        if b is None:
            b, bp = a, ap
        else:
            b, bp = _u64(np.asarray(b).reshape(-1, 4))
        out = np.zeros_like(a)
        self._check(fn(self.ptr, op, ap, bp, out.ctypes.data_as(_lib.u64p), a.shape[0]))
        return out

    def fr_batch(self, op, a, b=None):
        return self._batch(self.lib.zk_fr_batch, {"add": 0, "sub": 1, "mul": 2, "inv": 3, "inv_euclid": 4, "inv_divsteps": 5}[op], a, b)

    def fq_batch(self, op, a, b=None):
        return self._batch(self.lib.zk_fq_batch, {"add": 0, "sub": 1, "mul": 2, "inv": 3, "inv_euclid": 4, "inv_divsteps": 5}[op], a, b)

    def _pt2(self, fn, words, a, b, bwords):
        a, ap = _u64(np.asarray(a).reshape(-1, words))
        b, bp = _u64(np.asarray(b).reshape(-1, bwords))
        out = np.zeros_like(a)
        self._check(fn(self.ptr, ap, bp, out.ctypes.data_as(_lib.u64p), a.shape[0]))
        return out

    def g1_mul_batch(self, points, scalars):
        return self._pt2(self.lib.zk_g1_mul_batch, 8, points, scalars, 4)

    def g2_mul_batch(self, points, scalars):
        return self._pt2(self.lib.zk_g2_mul_batch, 16, points, scalars, 4)

    def g1_add_batch(self, a, b):
        return self._pt2(self.lib.zk_g1_add_batch, 8, a, b, 8)

    def g2_add_batch(self, a, b):
        return self._pt2(self.lib.zk_g2_add_batch, 16, a, b, 16)

    # ---- QAP ----
    @staticmethod
    def _rows(rows, keep):
        ptr, gate, val = rows
        ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
        gate = np.ascontiguousarray(gate, dtype=np.uint32)
        val = np.ascontiguousarray(np.asarray(val, dtype=np.uint64).reshape(-1, 4))
        keep += [ptr, gate, val]
        return _lib.SparseRows(ptr.ctypes.data_as(_lib.u64p), gate.ctypes.data_as(_lib.u32p), val.ctypes.data_as(_lib.u64p))

    @staticmethod
    def sparse_desc(log_n, m, input, u, v, w):
        keep = []
        d = _lib.QapSparseDesc(log_n, m, input, Context._rows(u, keep), Context._rows(v, keep), Context._rows(w, keep))
        d._keep = keep
        return d

    def qap_sparse(self, log_n, m, input, u, v, w):
        """u, v, w: (ptr[m+1], gate[nnz], val[nnz,4]) rows by wire (RootRepresentation, circuit/mod.rs:201)."""
        d = self.sparse_desc(log_n, m, input, u, v, w)
        p = C.c_void_p()
        self._check(self.lib.zk_qap_upload_sparse(self.ptr, C.byref(d), C.byref(p)))
        q = Qap(self, p, self.lib.zk_qap_free)
        q.n, q.m, q.input, q.dense, q.log_n = 1 << log_n, m, input, False, log_n
        return q

    def qap_sparse_integers(self, n, m, input, u, v, w):
        """Rows as in qap_sparse, over the roots 1..n that ASTParser emits (circuit/mod.rs:517); any n <= 2^21."""
        d = self.sparse_desc(0, m, input, u, v, w)
        p = C.c_void_p()
        self._check(self.lib.zk_qap_upload_sparse_integers(self.ptr, C.byref(d), n, C.byref(p)))
        q = Qap(self, p, self.lib.zk_qap_free)
        q.n, q.m, q.input, q.dense, q.roots = n, m, input, False, "integers"
        return q

    def qap_sparse_roots(self, roots, m, input, u, v, w):
        """Rows as in qap_sparse over ANY distinct roots (RootRepresentation::roots(), circuit/mod.rs:201-214): gate j = roots[j] ((n, 4) limbs)."""
        r = np.ascontiguousarray(np.asarray(roots, dtype=np.uint64).reshape(-1, 4))
        n = r.shape[0]
        d = self.sparse_desc(0, m, input, u, v, w)
        p = C.c_void_p()
        self._check(self.lib.zk_qap_upload_sparse_roots(self.ptr, C.byref(d), r.ctypes.data_as(_lib.u64p), n, C.byref(p)))
        q = Qap(self, p, self.lib.zk_qap_free)
        q.n, q.m, q.input, q.dense, q.roots = n, m, input, False, "arbitrary"
        return q

    def qap_dense(self, u, v, w, t, input):
        """u, v, w: (m, n, 4) coefficient matrices, t: (n+1, 4)."""
        u, up = _u64(u); v, vp = _u64(v); w, wp = _u64(w); t, tp = _u64(t)
        m, n = u.shape[0], u.shape[1]
        p = C.c_void_p()
        self._check(self.lib.zk_qap_upload_dense(self.ptr, up, vp, wp, tp, m, n, input, C.byref(p)))
        q = Qap(self, p, self.lib.zk_qap_free)
        q.n, q.m, q.input, q.dense = n, m, input, True
        return q

    # ---- CRS ----
    def setup(self, qap, trapdoor):
        """groth16::setup (mod.rs:134-197) with (alpha, beta, gamma, delta, x) injected: (5,4) limbs or 5 ints."""
        if not isinstance(trapdoor, np.ndarray):
            trapdoor = ints_to_limbs(list(trapdoor))
        td, tp = _u64(trapdoor.reshape(5, 4))
        p = C.c_void_p()
        self._check(self.lib.zk_setup(self.ptr, qap.ptr, tp, C.byref(p)))
        c = Crs(self, p, self.lib.zk_crs_free)
        c.n, c.m, c.input = qap.n, qap.m, qap.input
        return c

    @staticmethod
    def crs_arrays(n, m, input):
        return dict(alpha_g1=np.zeros(8, np.uint64), beta_g1=np.zeros(8, np.uint64), delta_g1=np.zeros(8, np.uint64),
                    xi_g1=np.zeros((n, 8), np.uint64), sum_gamma_g1=np.zeros((input + 1, 8), np.uint64),
                    sum_delta_g1=np.zeros((max(m - input - 1, 0), 8), np.uint64), xi_t_g1=np.zeros((max(n - 1, 0), 8), np.uint64),
                    beta_g2=np.zeros(16, np.uint64), gamma_g2=np.zeros(16, np.uint64), delta_g2=np.zeros(16, np.uint64),
                    xi_g2=np.zeros((n, 16), np.uint64))

    def crs_download(self, crs):
        arrs = self.crs_arrays(crs.n, crs.m, crs.input)
        out = _lib.CrsOut(**{k: v.ctypes.data_as(_lib.u64p) for k, v in arrs.items()})
        self._check(self.lib.zk_crs_download(self.ptr, crs.ptr, C.byref(out)))
        return arrs

    @staticmethod
    def crs_desc(n, m, input, arrs):
        keep = {k: np.ascontiguousarray(v, dtype=np.uint64) for k, v in arrs.items()}
        d = _lib.CrsDesc(n, m, input, **{k: v.ctypes.data_as(_lib.u64p) for k, v in keep.items()})
        d._keep = keep
        return d

    def crs_upload(self, n, m, input, arrs):
        d = self.crs_desc(n, m, input, arrs)
        p = C.c_void_p()
        self._check(self.lib.zk_crs_upload(self.ptr, C.byref(d), C.byref(p)))
        c = Crs(self, p, self.lib.zk_crs_free)
        c.n, c.m, c.input = n, m, input
        return c

    def qap_weighted_sum(self, qap, weights, which):
        """sum_i weights[i] * u_i / v_i / w_i (which = 0 / 1 / 2; mod.rs:233-253): coefficients (dense QAP) or values on the domain
        (sparse QAP, u and v only) as an (n, 4) uint64 array."""
        w, wp = _u64(np.asarray(weights).reshape(-1, 4))
        out = np.zeros((qap.n, 4), dtype=np.uint64)
        self._check(self.lib.zk_qap_weighted_sum(self.ptr, qap.ptr, wp, w.shape[0], which, out.ctypes.data_as(_lib.u64p)))
        return out

    def qap_save(self, qap, path):
        """Write the QAP container (zk_qap_save): sparse rows or dense matrices."""
        self._check(self.lib.zk_qap_save(self.ptr, qap.ptr, str(path).encode()))

    def qap_load(self, path):
        p = C.c_void_p()
        self._check(self.lib.zk_qap_load(self.ptr, str(path).encode(), C.byref(p)))
        q = Qap(self, p, self.lib.zk_qap_free)
        n, m, l, dense = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int()
        self._check(self.lib.zk_qap_dims(p, C.byref(n), C.byref(m), C.byref(l), C.byref(dense)))
        q.n, q.m, q.input, q.dense = n.value, m.value, l.value, bool(dense.value)
        kind = self.lib.zk_qap_kind(p)
        q.roots = {0: "unity", 2: "integers", 3: "arbitrary"}.get(kind)
        if kind == 0:
            q.log_n = q.n.bit_length() - 1      # roots of unity: n = 2^log_n
        return q

    def crs_save(self, crs, path):
        """Write the CRS container (zk_crs_save)."""
        self._check(self.lib.zk_crs_save(self.ptr, crs.ptr, str(path).encode()))

    def crs_load(self, path):
        p = C.c_void_p()
        self._check(self.lib.zk_crs_load(self.ptr, str(path).encode(), C.byref(p)))
        c = Crs(self, p, self.lib.zk_crs_free)
        n, m, l = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._check(self.lib.zk_crs_dims(p, C.byref(n), C.byref(m), C.byref(l)))
        c.n, c.m, c.input = n.value, m.value, l.value
        return c

    # ---- prove ----
    def prove(self, crs, qap, weights, r, s):
        """groth16::prove (mod.rs:213-296) with (r, s) injected; returns the 259-byte canonical proof."""
        w, wp = _u64(np.asarray(weights).reshape(-1, 4))
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        out = np.zeros(PROOF_BYTES, dtype=np.uint8)
        self._check(self.lib.zk_prove(self.ptr, crs.ptr, qap.ptr, wp, w.shape[0], rp, sp, out.ctypes.data_as(_lib.u8p)))
        return out.tobytes()

    def prove_dev(self, crs, qap, d_weights_ptr, m, r, s):
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        out = np.zeros(PROOF_BYTES, dtype=np.uint8)
        self._check(self.lib.zk_prove_dev(self.ptr, crs.ptr, qap.ptr, C.c_void_p(d_weights_ptr), m, rp, sp, out.ctypes.data_as(_lib.u8p)))
        return out.tobytes()

    def prove_submit(self, crs, qap, d_weights_ptr, m, r, s):
        """Enqueue one proof without waiting (at most two in flight); returns the ticket for prove_wait."""
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        t = C.c_int(-1)
        self._check(self.lib.zk_prove_submit(self.ptr, crs.ptr, qap.ptr, C.c_void_p(d_weights_ptr), m, rp, sp, C.byref(t)))
        return t.value

    def prove_submit_host(self, crs, qap, weights_host_ptr, m, r, s):
        """zk_prove_submit_host: the witness (m x 4 words) is in host memory at `weights_host_ptr` (page-locked memory from
        host_alloc makes the transfer asynchronous); it must stay valid until prove_wait."""
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        t = C.c_int(-1)
        self._check(self.lib.zk_prove_submit_host(self.ptr, crs.ptr, qap.ptr, C.c_void_p(weights_host_ptr), m, rp, sp, C.byref(t)))
        return t.value

    def host_alloc(self, shape, dtype=np.uint64):
        """Page-locked host array (zk_host_alloc); free with host_free(arr) when no transfer from it is pending."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        if self.lib.zk_host_alloc(nbytes, C.byref(p)) != 0:
            raise ZkError(-3)
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self.lib.zk_host_free(C.c_void_p(p))

    def prove_wait(self, ticket, partial=False):
        if partial:
            self._check(self.lib.zk_prove_wait(self.ptr, ticket, None))
            return None
        out = np.zeros(PROOF_BYTES, dtype=np.uint8)
        self._check(self.lib.zk_prove_wait(self.ptr, ticket, out.ctypes.data_as(_lib.u8p)))
        return out.tobytes()

    def prove_partial_submit(self, crs, qap, d_weights_ptr, m, r, s, rank, world, d_partial_ptr):
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        t = C.c_int(-1)
        self._check(self.lib.zk_prove_partial_submit(self.ptr, crs.ptr, qap.ptr, C.c_void_p(d_weights_ptr), m, rp, sp, rank, world,
                                                     C.c_void_p(d_partial_ptr), C.byref(t)))
        return t.value

    def prove_partial(self, crs, qap, d_weights_ptr, m, r, s, rank, world, d_partial_ptr):
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        self._check(self.lib.zk_prove_partial(self.ptr, crs.ptr, qap.ptr, C.c_void_p(d_weights_ptr), m, rp, sp, rank, world, C.c_void_p(d_partial_ptr)))

    # ---- batches (zkgpu.h: zk_prove_batch_submit / zk_prove_batch_wait) ----
    def prove_batch_submit(self, crs, qap, d_weight_ptrs, ms, rs, ss):
        """Enqueue len(rs) proofs over one CRS / QAP as a batch (own witness pointer, length and (r, s) each)."""
        count = len(rs)
        ptrs = (C.c_void_p * count)(*[C.c_void_p(p) for p in d_weight_ptrs])
        lens = (C.c_size_t * count)(*ms)
        r_ = np.ascontiguousarray(np.stack([fr_to_limbs(x) if isinstance(x, int) else np.asarray(x, dtype=np.uint64) for x in rs]).reshape(-1), dtype=np.uint64)
        s_ = np.ascontiguousarray(np.stack([fr_to_limbs(x) if isinstance(x, int) else np.asarray(x, dtype=np.uint64) for x in ss]).reshape(-1), dtype=np.uint64)
        t = C.c_int(-1)
        self._check(self.lib.zk_prove_batch_submit(self.ptr, crs.ptr, qap.ptr, count, ptrs, lens, r_.ctypes.data_as(_lib.u64p),
                                                   s_.ctypes.data_as(_lib.u64p), C.byref(t)))
        return t.value

    def prove_batch_wait(self, ticket, count):
        out = np.zeros(count * PROOF_BYTES, dtype=np.uint8)
        self._check(self.lib.zk_prove_batch_wait(self.ptr, ticket, count, out.ctypes.data_as(_lib.u8p)))
        raw = out.tobytes()
        return [raw[k * PROOF_BYTES:(k + 1) * PROOF_BYTES] for k in range(count)]

    # ---- multi-GPU scalar exchange (zkgpu.h: zk_prove_scalars_submit / zk_prove_msm_submit) ----
    def prove_exchange_elems(self, qap, world):
        """Element counts (32-byte Fr) of the four exchange arrays L, V, U, H for `world` ranks."""
        out = (C.c_size_t * 4)()
        self._check(self.lib.zk_prove_exchange_elems(qap.ptr, world, out))
        return [int(x) for x in out]

    def prove_scalars_submit(self, crs, qap, d_weights_ptr, m, r, s, world, d_ptrs):
        """SpMV / NTT stage of one proof; the scalars of the four inner products go to d_ptrs = (L, V, U, H)."""
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        t = C.c_int(-1)
        self._check(self.lib.zk_prove_scalars_submit(self.ptr, crs.ptr, qap.ptr, C.c_void_p(d_weights_ptr), m, rp, sp, world,
                                                     *[C.c_void_p(p) for p in d_ptrs], C.byref(t)))
        return t.value

    def prove_msm_submit(self, crs, qap, sets, rank, world, d_ptrs, d_partials_ptr):
        """Inner products of `sets` proofs over this rank's points from the exchanged chunks d_ptrs = (L, V, U, H)."""
        t = C.c_int(-1)
        self._check(self.lib.zk_prove_msm_submit(self.ptr, crs.ptr, qap.ptr, sets, rank, world, *[C.c_void_p(p) for p in d_ptrs],
                                                 C.c_void_p(d_partials_ptr), C.byref(t)))
        return t.value

    def prove_combine(self, crs, d_partials_ptr, world, r, s):
        r_, rp = _u64(fr_to_limbs(r) if isinstance(r, int) else r)
        s_, sp = _u64(fr_to_limbs(s) if isinstance(s, int) else s)
        out = np.zeros(PROOF_BYTES, dtype=np.uint8)
        self._check(self.lib.zk_prove_combine(self.ptr, crs.ptr, C.c_void_p(d_partials_ptr), world, rp, sp, out.ctypes.data_as(_lib.u8p)))
        return out.tobytes()

    # ---- verify ----
    def verify(self, crs, inputs, proof):
        """groth16::verify (mod.rs:299-320): inputs = the `verify` wires (ints or (k,4) limbs), proof = 259 bytes."""
        a = ints_to_limbs(list(inputs)) if not isinstance(inputs, np.ndarray) else np.ascontiguousarray(inputs, dtype=np.uint64).reshape(-1, 4)
        pb = np.frombuffer(bytes(proof), dtype=np.uint8).copy()
        ok = C.c_int(0)
        self._check(self.lib.zk_verify(self.ptr, crs.ptr, a.ctypes.data_as(_lib.u64p), a.shape[0], pb.ctypes.data_as(_lib.u8p), C.byref(ok)))
        return bool(ok.value)

    # ---- profiling ----
    def profile_reset(self):
        self._check(self.lib.zk_profile_reset(self.ptr))

    def profile(self):
        out = {}
        for i in range(self.lib.zk_profile_count(self.ptr)):
            name, ms, cnt, by = C.c_char_p(), C.c_double(), C.c_uint64(), C.c_double()
            self.lib.zk_profile_entry(self.ptr, i, C.byref(name), C.byref(ms), C.byref(cnt), C.byref(by))
            out[name.value.decode()] = dict(total_ms=ms.value, launches=cnt.value, algo_bytes=by.value)
        return out
