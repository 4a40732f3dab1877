"""groth16::{setup, prove, verify} with the reference's names and argument order (groth16/mod.rs).

    qap              = QAP.from_zk(ctx, code)            # QAP::from(ASTParser::try_parse(code))  (fr.rs:140-173)
    sigmag1, sigmag2 = setup(qap)                         # mod.rs:134   (trapdoor from os.urandom unless given)
    proof            = prove(qap, (sigmag1, sigmag2), weights)           # mod.rs:213
    ok               = verify((sigmag1, sigmag2), inputs, proof)         # mod.rs:299

`sigmag1` / `sigmag2` are two views of one device-resident CRS handle (SigmaG1, SigmaG2).
"""
import os

from . import R_MODULUS, ints_to_limbs
from .circuit import Circuit


def random_elem():
    """Random for FrLocal (fr.rs:90-99): uniform, never zero."""
    while True:
        v = int.from_bytes(os.urandom(40), "little") % R_MODULUS
        if v:
            return v


class QAP:
    """QAP<CoefficientPoly<FrLocal>> (mod.rs:60-67), device resident."""

    def __init__(self, ctx, handle, circuit=None):
        self.ctx, self.handle, self.circuit = ctx, handle, circuit

    @classmethod
    def from_zk(cls, ctx, code, sparse=False):
        """sparse: keep the root representation's rows over the roots 1..n instead of interpolating them (any size)"""
        c = Circuit(code)
        return cls(ctx, c.qap_sparse(ctx) if sparse else c.qap(ctx), c)


class _Sigma:
    """One half of the CRS.  The points live on the device; the reference's fields (mod.rs:105-121) are fetched on first
    access as affine coordinates in canonical little-endian 64-bit limbs: (8,) per G1 point, (16,) per G2 point."""
    _fields = ()

    def __init__(self, ctx, crs, shared):
        self.ctx, self.crs, self._shared = ctx, crs, shared

    def __getattr__(self, name):
        if name in type(self)._fields:
            if "arrays" not in self._shared:
                self._shared["arrays"] = self.ctx.crs_download(self.crs)
            return self._shared["arrays"][name + type(self)._suffix]
        raise AttributeError(name)


class SigmaG1(_Sigma):
    _fields = ("alpha", "beta", "delta", "xi", "sum_gamma", "sum_delta", "xi_t")
    _suffix = "_g1"


class SigmaG2(_Sigma):
    _fields = ("beta", "gamma", "delta", "xi")
    _suffix = "_g2"


def setup(qap, trapdoor=None):
    td = trapdoor if trapdoor is not None else [random_elem() for _ in range(5)]
    crs = qap.ctx.setup(qap.handle, ints_to_limbs(list(td)))
    shared = {}
    return SigmaG1(qap.ctx, crs, shared), SigmaG2(qap.ctx, crs, shared)


def prove(qap, sigma, weights, rs=None):
    sigmag1, sigmag2 = sigma
    assert sigmag1.crs is sigmag2.crs, "SigmaG1 / SigmaG2 come from different setups"
    r, s = rs if rs is not None else (random_elem(), random_elem())
    return qap.ctx.prove(sigmag1.crs, qap.handle, weights, r, s)


def verify(sigma, inputs, proof):
    sigmag1, sigmag2 = sigma
    return sigmag1.ctx.verify(sigmag1.crs, inputs, proof)


def weights(code, inputs):
    """circuit::weights (circuit/mod.rs:529-637)."""
    return Circuit(code).weights(inputs)
