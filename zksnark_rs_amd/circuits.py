"""Synthetic "chain" circuit = test_programs/deg_15.zk generalised to n gates.

  gate k (1 <= k < n): t_k = x * (t_{k-1} + a_k)      (t_0 absent)
  gate n             : y   = 1 * (t_{n-1} + a_n)
  verify (x, y)  ->  wires 0:1 1:x 2:y 3:t1 4:a1 5:t2 6:a2 ... (ASTParser order,
  circuit/mod.rs:278-526), m = 2n+2, l = 2, nnz(u, v, w) = n, 2n-1, n.
Gate k sits at domain point w^(k-1).  This is the input side (host, like the reference's
parser/`weights`), not the accelerated path.
"""
import numpy as np

from . import R_MODULUS, ints_to_limbs


def chain_rows(log_n):
    """Returns (m, l, u, v, w) with each of u, v, w = (ptr[m+1], gate[nnz], val[nnz,4]) by wire."""
    n = 1 << log_n
    m = 2 * n + 2
    one = np.array([1, 0, 0, 0], dtype=np.uint64)
    k = np.arange(1, n + 1, dtype=np.int64)           # gate numbers 1..n
    t_wire = 2 * k + 1                                 # wire of t_k (k < n)
    a_wire = np.where(k < n, 2 * k + 2, 2 * n + 1)

    def build(wires, gates):
        order = np.lexsort((gates, wires))
        wires, gates = wires[order], gates[order]
        ptr = np.zeros(m + 1, dtype=np.uint64)
        np.add.at(ptr, wires + 1, 1)
        ptr = np.cumsum(ptr).astype(np.uint64)
        val = np.tile(one, (len(gates), 1))
        return ptr, gates.astype(np.uint32), val

    # u: wire x (1) at gates 1..n-1, wire 0 at gate n
    u = build(np.concatenate([np.full(n - 1, 1, np.int64), [0]]), np.concatenate([k[:-1] - 1, [n - 1]]))
    # v: t_{k-1} at gate k (k>=2), a_k at gate k
    v = build(np.concatenate([t_wire[:-1], a_wire]), np.concatenate([k[1:] - 1, k - 1]))
    # w: t_k at gate k (k<n), y (2) at gate n
    w = build(np.concatenate([t_wire[:-1], [2]]), np.concatenate([k[:-1] - 1, [n - 1]]))
    return m, 2, u, v, w


def chain_weights(log_n, x, avals):
    """Witness for inputs x, a_1..a_n (Python ints mod r): [1, x, y, t1, a1, t2, a2, ...] as (m,4) limbs."""
    n = 1 << log_n
    m = 2 * n + 2
    w = [0] * m
    w[0], w[1] = 1, x % R_MODULUS
    prev = 0
    for k in range(1, n + 1):
        ak = avals[k - 1] % R_MODULUS
        if k < n:
            w[2 * k + 2] = ak
            prev = (x * (prev + ak)) % R_MODULUS
            w[2 * k + 1] = prev
        else:
            w[2 * n + 1] = ak
            w[2] = (prev + ak) % R_MODULUS
    return ints_to_limbs(w)
