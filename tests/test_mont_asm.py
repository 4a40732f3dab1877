"""-m "not gpu": the hand-scheduled Montgomery multipliers (zksnark_rs_amd/csrc/mont_asm.inc) are GENERATED.  The generator executes
every instruction sequence on Python integers with the hardware's wrap-around semantics and compares the limbs with the C++
definition of lazy29.cuh (mul, sqr, a b + c d, the Fq2 product; both moduli; limb extremes) before it writes the file -- so the
register reuse and the column bounds of the inline asm are checked on the CPU.  This test runs that check and refuses a
mont_asm.inc that is not what the generator writes (a hand edit would escape the check)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_generator():
    spec = importlib.util.spec_from_file_location("gen_mont_asm", os.path.join(ROOT, "tools", "gen_mont_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_schedules_hold_on_the_cpu():
    gen = load_generator()
    gen.check()          # asserts inside: limbs == model, value == a b / R mod p, no column leaves the signed 64-bit range
    assert len(gen.gen_mul("mul").ins) == 162 + 43 and len(gen.gen_mul("sqr").ins) == 177
    assert len(gen.gen_mul("sum").ins) == 243 + 43 and len(gen.gen_fp2().ins) == 581


def test_committed_file_is_the_generated_one(tmp_path):
    gen = load_generator()
    out = tmp_path / "mont_asm.inc"
    gen.write(str(out))
    assert out.read_text() == open(os.path.join(ROOT, "zksnark_rs_amd", "csrc", "mont_asm.inc")).read()


def load_body_generator():
    spec = importlib.util.spec_from_file_location("gen_madd_asm", os.path.join(ROOT, "tools", "gen_madd_asm.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_whole_addition_bodies_hold_on_the_cpu():
    """csrc/madd_asm.inc (round 6): the mixed XYZZ addition of k_msm_accumulate<Fq> as one asm body with in-place multipliers.  The
    generator executes both bodies (even / odd) with the hardware's wrap-around semantics against the limb-exact definition of
    lazy29.cuh's madd_xyzz_nz AND against the affine group law on BN254, incl. the same-x masks and the exact-zero comparison."""
    gen = load_body_generator()
    counts = gen.check(trials=60)
    # 6 mul + 2 sqr + 1 sum = 1467 multiply-adds + 9 x 43 around them, P, R, X3 (+ its carry propagation), D, the two filter compares
    assert counts["even"] == counts["odd"] == 1941


def test_committed_body_file_is_the_generated_one(tmp_path):
    gen = load_body_generator()
    out = tmp_path / "madd_asm.inc"
    gen.write(str(out))
    assert out.read_text() == open(os.path.join(ROOT, "zksnark_rs_amd", "csrc", "madd_asm.inc")).read()
