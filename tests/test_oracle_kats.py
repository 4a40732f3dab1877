"""-m "not gpu": pins the CPU oracle against every known-answer test the reference holds for the
hot path.  Each case is one Rust #[test]/doctest restated in oracle/test_kats.cpp
(which cites the reference file:line); the binary prints PASS/FAIL per case."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = """powers_test dft_test idft_test degree_test evaluate_doctest polynomial_division_test
polynomial_divisionby0_test dummy_add dummy_neg dummy_sub dummy_sum dummy_mul dummy_scalar_mul dummy_div
dummy_lagrange dummy_from_roots dummy_root_poly z251_tests single_mult_honest single_mult_random_proof
quadratic_share_honest quadratic_share_random_proof qap_from_roots qap_from_file qap_from_ast
try_parse_impl_test evaluate_test weights_test simple_zk_fixture deg15_structure bn_constants
exp_encrypted_test single_mult_honest_bn bn_encrypt_quad_cubic_deg15 ntt_matches_dft fast_matches_faithful""".split()


def test_case_list_is_complete(oracle_build):
    out = subprocess.run([os.path.join(oracle_build, "oracle_kats"), "--list"], capture_output=True, text=True, check=True)
    assert out.stdout.split() == CASES


@pytest.mark.parametrize("case", CASES)
def test_reference_kat(oracle_build, case):
    res = subprocess.run([os.path.join(oracle_build, "oracle_kats"), case], capture_output=True, text=True, cwd=ROOT)
    assert res.returncode == 0 and ("PASS " + case) in res.stdout, res.stdout + res.stderr
