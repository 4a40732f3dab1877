"""The product's .zk front end (host code behind the C ABI) against the reference's parser KATs and
the oracle.  CPU part: tokenizer / ASTParser::try_parse / weights.  GPU part: QAP::from(root_rep)
(Lagrange interpolation on the GPU) and the whole .zk -> proof flow with no oracle in the loop."""
import os

import numpy as np
import pytest

import zksnark_rs_amd as zk
from zksnark_rs_amd.circuit import Circuit, ParseErr, qap_download_dense

ZK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zk")
PROGS = ["simple.zk", "lispesque_quad.zk", "lispesque_cubic.zk", "deg_15.zk"]

QUAD = """(in x a b c)
                    (out y)
                    (verify x y)

                    (program
                        (= t1
                            (* x a))
                        (= t2
                            (* x (+ t1 b)))
                        (= y
                            (* 1 (+ t2 c))))"""


def rows_as_lists(c, which):
    ptr, gate, val = c.rows(which)
    return [[(int(gate[k]) + 1, zk.limbs_to_int(val[k])) for k in range(int(ptr[i]), int(ptr[i + 1]))] for i in range(c.m)]


def test_try_parse_impl_test():
    """circuit/mod.rs:664-718: the expected DummyRep of the quadratic program, literally."""
    c = Circuit(QUAD)
    assert (c.m, c.n, c.input) == (8, 3, 2)
    assert rows_as_lists(c, 0) == [[(3, 1)], [(1, 1), (2, 1)], [], [], [], [], [], []]
    assert rows_as_lists(c, 1) == [[], [], [], [(2, 1)], [(1, 1)], [(3, 1)], [(2, 1)], [(3, 1)]]
    assert rows_as_lists(c, 2) == [[], [], [(3, 1)], [(1, 1)], [], [(2, 1)], [], []]


def test_weights_test_and_simple_fixture():
    """circuit/mod.rs:745-769 and SURVEY appendix B (hand trace of simple.zk)."""
    c = Circuit(open(os.path.join(ZK_DIR, "simple.zk")).read())
    assert zk.limbs_to_ints(c.weights([3, 2, 4])) == [1, 2, 34, 6, 3, 4]
    assert rows_as_lists(c, 0) == [[(2, 1)], [], [], [], [(1, 1)], []]
    assert rows_as_lists(c, 1) == [[(2, 6)], [(1, 1)], [], [(2, 4)], [], [(2, 1)]]
    assert rows_as_lists(c, 2) == [[], [], [(2, 1)], [(1, 1)], [], []]


@pytest.mark.parametrize("prog", PROGS)
def test_parser_and_weights_match_oracle(orc, prog):
    code = open(os.path.join(ZK_DIR, prog)).read()
    c = Circuit(code)
    q = orc.zk_qap_dense(code)
    assert (c.m, c.n, c.input, c.n_in) == (q["m"], q["n"], q["input"], q["n_in"])
    rng = zk.SplitMix64(17)
    ins = zk.ints_to_limbs([rng.fr() for _ in range(c.n_in)])
    assert np.array_equal(c.weights(ins), orc.zk_weights(code, ins, c.m))


@pytest.mark.parametrize("code,frag", [
    ("(in a) (out b) (verify b)", "Expected exactly one each"),
    ("(in a)\n(out b)\n(verify b)\n(program (= b (* a ( a))))", "SyntaxErr(4, found whitespace after '(')"),
    ("(in a)\n(out b)\n(verify b)\n(program (= b (* a+ a)))", "unexpected operator"),
    ("(out b)\n(in a)\n(verify b)\n(program (= b (* a a)))", "Expected first expression to be 'in'"),
    ("(in a)\n(out b)\n(verify b)\n(program (= b (* a a)) (= b (* a a)))", "cannot be the output of two different gates"),
    ("(in a)\n(out b)\n(verify b)\n(program (= c (* a a)) (= c (* a a)))", "Already declared variable"),
    ("(in a)\n(out b)\n(verify b)\n(program (= b (* (+ (* a 3)) a)))", "LHS of a '*' expression"),
])
def test_parse_errors(code, frag):
    """ParseErr cases of ASTParser::try_parse (circuit/mod.rs:248-515, ast.rs:300-370)."""
    with pytest.raises(ParseErr) as e:
        Circuit(code)
    assert frag in str(e.value)


def test_weights_errors():
    c = Circuit(QUAD)
    with pytest.raises(ParseErr) as e:
        c.weights([1, 2, 3])
    assert "Wrong number of values supplied" in str(e.value)


@pytest.mark.gpu
@pytest.mark.parametrize("prog", PROGS)
def test_qap_from_root_rep_on_gpu(ctx, orc, prog):
    """QAP::from (fr.rs:140-173): the GPU interpolation gives the oracle's dense coefficient tables."""
    code = open(os.path.join(ZK_DIR, prog)).read()
    c = Circuit(code)
    u, v, w, t = qap_download_dense(ctx, c.qap(ctx))
    q = orc.zk_qap_dense(code)
    assert np.array_equal(u, q["u"]) and np.array_equal(v, q["v"]) and np.array_equal(w, q["w"]) and np.array_equal(t, q["t"])


@pytest.mark.gpu
def test_simple_circuit_flow_without_oracle_inputs(ctx, orc):
    """lib.rs:156-190 (simple_circuit_test) with the product's own parser, witness and QAP::from; the
    oracle only checks the result (faithful prove on its own parse of the same file)."""
    code = open(os.path.join(ZK_DIR, "simple.zk")).read()
    c = Circuit(code)
    weights = c.weights([3, 2, 4])
    qap = c.qap(ctx)
    rng = zk.SplitMix64(3)
    td = zk.ints_to_limbs([rng.fr() for _ in range(5)])
    r, s = rng.fr(), rng.fr()
    crs = ctx.setup(qap, td)
    proof = ctx.prove(crs, qap, weights, r, s)
    q = orc.zk_qap_dense(code)
    cdesc = ctx.crs_desc(q["n"], q["m"], q["input"], ctx.crs_download(crs))
    assert proof == orc.prove_dense(q["u"], q["v"], q["w"], q["t"], q["input"], cdesc, weights, r, s)
    assert proof == orc.trapdoor_proof_dense(q["u"], q["v"], q["w"], q["t"], q["input"], td, weights, r, s)
