#!/usr/bin/env python3
"""Generates the committed golden fixtures of SURVEY.md 8c ("Fixtures to commit") into tests/golden/.

    python tests/golden/gen_golden.py            # writes proofs.json, circuits.json, ntt_2p10.bin, alt_bn128.json

The reference (Rust + crate bn 0.4.3) cannot be built in this image, so the values come from the CPU oracle -- and
every value that enters a fixture is produced TWICE by independent code before it is written:
  * proofs: oracle/pyref.py (big-int affine arithmetic, the reference's algorithms literally: Lagrange QAP, schoolbook
    product, long division, per-term scalar multiplications) AND the closed-form trapdoor proof; for the 2^8 chain
    circuit additionally the C++ restatement (oracle/groth16.hpp, Montgomery/Jacobian arithmetic);
  * the DummyRep of simple.zk / deg_15.zk: the C++ parser (oracle/zkparse.hpp, follows circuit/mod.rs:226-527) AND the
    hand trace of the reference's parser in SURVEY.md Appendix B, restated below;
  * NTT: pyref.dft (the reference's naive O(n^2) definition, field/mod.rs:508-520) AND the C++ oracle's dft;
  * alt_bn128: PUBLIC known-answer vectors of the Ethereum precompiles (EIP-196 ecAdd / ecMul, EIP-197 pairing check;
    go-ethereum core/vm/testdata/precompiles/bn256Add.json, bn256ScalarMul.json, bn256Pairing.json, vector names kept),
    checked here against pyref before they are written.  They pin the bn byte boundary from OUTSIDE this repository:
    the reference holds no known answer for any Fr / G1 / G2 value.
Everything is deterministic (SplitMix64 seeds below); re-running the script must reproduce the files byte for byte."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import pyref as P   # noqa: E402


def hx(v):
    return "%064x" % v


# ---- the reference parser's output for the two .zk configs (hand trace of circuit/mod.rs:278-526) ----
def simple_root_rep():
    m = 6   # wires 0:1 1:b 2:x 3:temp 4:a 5:c ; roots {1, 2}
    u = [[] for _ in range(m)]; v = [[] for _ in range(m)]; w = [[] for _ in range(m)]
    u[0] = [(2, 1)]; u[4] = [(1, 1)]
    v[0] = [(2, 6)]; v[1] = [(1, 1)]; v[3] = [(2, 4)]; v[5] = [(2, 1)]
    w[2] = [(2, 1)]; w[3] = [(1, 1)]
    return dict(u=u, v=v, w=w, roots=[1, 2], input=2)


def deg15_root_rep():
    # wires 1,x,y,t1,a,t2,b,...,t15,o,p (m = 34): t_k = wire 2k+1, letter_k = wire 2k+2 (k <= 15), p = wire 33
    m, n = 34, 16
    u = [[] for _ in range(m)]; v = [[] for _ in range(m)]; w = [[] for _ in range(m)]
    for k in range(1, 16):
        w[2 * k + 1].append((k, 1)); u[1].append((k, 1))
        if k >= 2:
            v[2 * k - 1].append((k, 1))
        v[2 * k + 2].append((k, 1))
    w[2].append((16, 1)); u[0].append((16, 1)); v[31].append((16, 1)); v[33].append((16, 1))
    return dict(u=u, v=v, w=w, roots=list(range(1, n + 1)), input=2)


def deg15_weights(inputs):
    # inputs in `in` order: x, then the 16 letters a..p
    F = P.FR
    x, letters = inputs[0], inputs[1:]
    wts = [0] * 34
    wts[0], wts[1] = 1, x
    prev = 0
    for k in range(1, 16):
        wts[2 * k + 2] = letters[k - 1]
        prev = F.mul(x, F.add(prev, letters[k - 1]))
        wts[2 * k + 1] = prev
    wts[33] = letters[15]
    wts[2] = F.add(prev, letters[15])
    return wts


def dense_rows(polys, n):
    return [[hx(c) for c in (p + [0] * n)[:n]] if any(p) else [] for p in polys]


def proof_case(name, rr, weights, seed, extra=None, faithful_python=True):
    rng = P.SplitMix64(seed)
    td = [rng.fr() for _ in range(5)]
    r, s = rng.fr(), rng.fr()
    qap = P.qap_from_root_rep(P.FR, rr)
    closed = P.enc_proof(*P.trapdoor_proof(qap, td, weights, r, s))
    if faithful_python:
        s1, s2 = P.setup_with_trapdoor(qap, td)
        proof = P.enc_proof(*P.prove_with_rs(qap, s1, s2, weights, r, s))
        assert proof == closed, name + ": faithful path and trapdoor closed form differ"
    else:
        proof = closed
    assert len(proof) == 259
    case = dict(name=name, seed=seed, trapdoor=[hx(t) for t in td], r=hx(r), s=hx(s), n=qap["degree"], m=len(rr["u"]), input=rr["input"],
                weights_sha256=hashlib.sha256(b"".join(w.to_bytes(32, "little") for w in weights)).hexdigest(), proof=proof.hex())
    if len(weights) <= 64:
        case["weights"] = [hx(w) for w in weights]
    if extra:
        case.update(extra)
    return case, qap


def limbs(vals):
    return np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)


def main():
    import oracle_lib
    orc = oracle_lib.load()
    zk_dir = os.path.join(HERE, "zk")
    cases, circuits = [], {}

    # -- config 1: simple.zk, a=3 b=2 c=4 (lib.rs:156-190) --
    code = open(os.path.join(zk_dir, "simple.zk")).read()
    rr = simple_root_rep()
    wts = [1, 2, 34, 6, 3, 4]                                   # circuit/mod.rs:759-766
    got = orc.zk_weights(code, limbs([3, 2, 4]), 6)
    assert [int.from_bytes(got[i].tobytes(), "little") for i in range(6)] == wts
    case, qap = proof_case("simple.zk", rr, wts, 101, dict(inputs=[3, 2, 4], verify_inputs=[2, 34]))
    dq = orc.zk_qap_dense(code)
    for k in "uvw":
        for i in range(6):
            want = (qap[k][i] + [0, 0])[:2] if any(qap[k][i]) else [0, 0]
            have = [int.from_bytes(dq[k][i, j].tobytes(), "little") for j in range(2)]
            assert want == have, ("simple.zk parser vs hand trace", k, i)
    cases.append(case)
    circuits["simple.zk"] = dict(m=6, n=2, input=2, wires=["1", "b", "x", "temp", "a", "c"], roots=[1, 2],
                                 u=rr["u"], v=rr["v"], w=rr["w"], weights_for_inputs_3_2_4=wts,
                                 coefficient_form=dict(U=[hx(c) for c in P.poly_sum(P.FR, [P.poly_scale(P.FR, p, a) for p, a in zip(qap["u"], wts)])],
                                                       t=[hx(c) for c in qap["t"]]))

    # -- config 2: deg_15.zk, 17 seeded inputs --
    code = open(os.path.join(zk_dir, "deg_15.zk")).read()
    rr = deg15_root_rep()
    rng = P.SplitMix64(202)
    inputs = [rng.fr() for _ in range(17)]
    wts = deg15_weights(inputs)
    got = orc.zk_weights(code, limbs(inputs), 34)
    assert [int.from_bytes(got[i].tobytes(), "little") for i in range(34)] == wts, "deg_15 weights: parser vs hand trace"
    case, qap = proof_case("deg_15.zk", rr, wts, 203, dict(inputs=[hx(v) for v in inputs], verify_inputs=[hx(wts[1]), hx(wts[2])]))
    dq = orc.zk_qap_dense(code)
    for k in "uvw":
        for i in range(34):
            want = (qap[k][i] + [0] * 16)[:16] if any(qap[k][i]) else [0] * 16
            have = [int.from_bytes(dq[k][i, j].tobytes(), "little") for j in range(16)]
            assert want == have, ("deg_15.zk parser vs hand trace", k, i)
    cases.append(case)
    circuits["deg_15.zk"] = dict(m=34, n=16, input=2, roots=list(range(1, 17)), u=rr["u"], v=rr["v"], w=rr["w"])

    # -- synthetic chain circuits 2^4 and 2^8, roots w^j --
    import zksnark_rs_amd as zk
    from zksnark_rs_amd.circuits import chain_rows
    for log_n, seed in ((4, 304), (8, 308)):
        n = 1 << log_n
        w_n = P.omega(log_n)
        roots = [pow(w_n, j, P.R) for j in range(n)]
        rr = P.chain_root_rep(n, roots)
        rng = P.SplitMix64(seed)
        x = rng.fr()
        avals = [rng.fr() for _ in range(n)]
        wts = P.chain_weights(n, x, avals)
        case, _ = proof_case("chain_2p%d" % log_n, rr, wts, seed + 1, dict(log_n=log_n, input_seed=seed), faithful_python=(log_n <= 4))
        # the C++ restatement, faithful path, on the same instance
        m, l, u, v, w = chain_rows(log_n)
        desc = zk.Context.sparse_desc(log_n, m, l, u, v, w)
        rng2 = P.SplitMix64(seed + 1)
        td = [rng2.fr() for _ in range(5)]
        r, s = rng2.fr(), rng2.fr()
        arrs = orc.setup_sparse(desc, zk.ints_to_limbs(td), n, m, l, True)
        cdesc = zk.Context.crs_desc(n, m, l, arrs)
        cpp = orc.prove_sparse(desc, cdesc, limbs(wts), r, s, True)
        assert cpp.hex() == case["proof"], "chain 2^%d: C++ faithful oracle and Python twin differ" % log_n
        cases.append(case)

    json.dump(dict(encoding="A | B | C, G1 = 04 | x | y, G2 = 04 | x.c1 | x.c0 | y.c1 | y.c0, 32-byte big-endian affine coordinates (DESIGN.md 1)",
                   rng="SplitMix64(seed): trapdoor alpha, beta, gamma, delta, x, then r, s (rejection sampling < r, non-zero)", cases=cases),
              open(os.path.join(HERE, "proofs.json"), "w"), indent=1)
    json.dump(circuits, open(os.path.join(HERE, "circuits.json"), "w"), indent=1)

    # -- 2^10-point Fr NTT pair: out[k] = sum_j in[j] w^(jk), natural order (field/mod.rs:508-520) --
    rng = P.SplitMix64(410)
    seq = [rng.fr() for _ in range(1024)]
    w10 = P.omega(10)
    out = P.dft(P.FR, seq, w10)
    cpp = orc.dft_fr(limbs(seq), limbs([w10])[0])
    assert [int.from_bytes(cpp[i].tobytes(), "little") for i in range(1024)] == out, "NTT: pyref.dft and the C++ oracle differ"
    with open(os.path.join(HERE, "ntt_2p10.bin"), "wb") as f:   # 1024 inputs then 1024 outputs, 32-byte little-endian canonical
        f.write(b"".join(v.to_bytes(32, "little") for v in seq + out))

    # -- public alt_bn128 known answers --
    h = lambda s_: int(s_, 16)
    G = P.G1_GEN
    vec = dict(
        source="Ethereum precompile test vectors (EIP-196 / EIP-197), go-ethereum core/vm/testdata/precompiles/bn256{Add,ScalarMul,Pairing}.json",
        ec_add=[
            dict(name="cdetrio11", a=[hx(1), hx(2)], b=[hx(1), hx(2)],
                 out=["030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3", "15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4"]),
            dict(name="chfast1",
                 a=["18b18acfb4c2c30276db5411368e7185b311dd124691610c5d3b74034e093dc9", "063c909c4720840cb5134cb9f59fa749755796819658d32efc0d288198f37266"],
                 b=["07c2b7f58a84bd6145f00c9c2bc0bb1a187f20ff2c92963a88019e7c6a014eed", "06614e20c147e940f2d70da3f74c9a17df361706a4485c742bd6788478fa17d7"],
                 out=["2243525c5efd4b9c3d3c45ac0ca3fe4dd85e830a4ce6b65fa1eeaee202839703", "301d1d33be6da8e509df21cc35964723180eed7532537db9ae5e7d48f195c915"]),
        ],
        ec_mul=[
            dict(name="chfast1",
                 p=["2bd3e6d0f3b142924f5ca7b49ce5b9d54c4703d7ae5648e61d02268b1a0a9fb7", "21611ce0a6af85915e2f1d70300909ce2e49dfad4a4619c8390cae66cefdb204"],
                 k="00000000000000000000000000000000000000000000000011138ce750fa15c2",
                 out=["070a8d6a982153cae4be29d434e8faef8a47b274a053f5a4ee2a6c9c13c31e5c", "031b8ce914eba3a9ffb989f9cdd5b0f01943074bf4f0f315690ec3cec6981afc"]),
            dict(name="cdetrio1 (scalar 2^256 - 1: EIP-196 scalars are plain integers; the point has order r, so k mod r is used)",
                 p=["1a87b0584ce92f4593d161480614f2989035225609f08058ccfa3d0f940febe3", "1a2f3c951f6dadcc7ee9007dff81504b0fcd6d7cf59996efdc33d92bf7f9f8f6"],
                 k="f" * 64,
                 out=["2cde5879ba6f13c0b5aa4ef627f159a3347df9722efce88a9afbb20b763b4c41", "1aa7e43076f6aee272755a7f9b84832e71559ba0d2e0b17d5f9f01755e5b0d11"]),
        ],
        pairing_check=[
            dict(name="jeff1", expect=1, layout="per pair: G1 x, y, then G2 x.c1, x.c0, y.c1, y.c0 (EIP-197 order)",
                 words="1c76476f4def4bb94541d57ebba1193381ffa7aa76ada664dd31c16024c43f59 3034dd2920f673e204fee2811c678745fc819b55d3e9d294e45c9b03a76aef41 "
                       "209dd15ebff5d46c4bd888e51a93cf99a7329636c63514396b4a452003a35bf7 04bf11ca01483bfa8b34b43561848d28905960114c8ac04049af4b6315a41678 "
                       "2bb8324af6cfc93537a2ad1a445cfd0ca2a71acd7ac41fadbf933c2a51be344d 120a2a4cf30c1bf9845f20c6fe39e07ea2cce61f0c9bb048165fe5e4de877550 "
                       "111e129f1cf1097710d41c4ac70fcdfa5ba2023c6ff1cbeac322de49d1b6df7c 2032c61a830e3c17286de9462bf242fca2883585b93870a73853face6a6bf411 "
                       "198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2 1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed "
                       "090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b 12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa".split()),
        ],
        generators=dict(g1=[hx(G[0]), hx(G[1])],
                        g2_x_c0_c1=[hx(P.G2_GEN[0][0]), hx(P.G2_GEN[0][1])], g2_y_c0_c1=[hx(P.G2_GEN[1][0]), hx(P.G2_GEN[1][1])]),
    )
    for t in vec["ec_add"]:
        a, b, o = [tuple(h(c) for c in t[k]) for k in ("a", "b", "out")]
        assert P.g1_on_curve(a) and P.g1_on_curve(b) and P.g1_add(a, b) == o, t["name"]
    for t in vec["ec_mul"]:
        p, o = tuple(h(c) for c in t["p"]), tuple(h(c) for c in t["out"])
        assert P.g1_on_curve(p) and P.g1_mul(p, h(t["k"]) % P.R) == o, t["name"]
    for t in vec["pairing_check"]:
        v = [h(x) for x in t["words"]]
        f = P.FQ12_ONE
        for i in range(len(v) // 6):
            g1 = (v[6 * i], v[6 * i + 1]); g2 = ((v[6 * i + 3], v[6 * i + 2]), (v[6 * i + 5], v[6 * i + 4]))
            assert P.g1_on_curve(g1) and P.g2_on_curve(g2)
            f = P.fq12_mul(f, P.miller_loop(g1, g2))
        assert (P.final_exponentiation(f) == P.FQ12_ONE) == bool(t["expect"]), t["name"]
    json.dump(vec, open(os.path.join(HERE, "alt_bn128.json"), "w"), indent=1)
    print("wrote proofs.json (%d cases), circuits.json, ntt_2p10.bin, alt_bn128.json" % len(cases))


if __name__ == "__main__":
    main()
