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
    assert len(gen.gen_mul("mul").ins) == 162 + 44 and len(gen.gen_mul("sqr").ins) == 178
    assert len(gen.gen_mul("sum").ins) == 243 + 44 and len(gen.gen_fp2().ins) == 583


def test_committed_file_is_the_generated_one(tmp_path):
    gen = load_generator()
    out = tmp_path / "mont_asm.inc"
    gen.write(str(out))
    assert out.read_text() == open(os.path.join(ROOT, "zksnark_rs_amd", "csrc", "mont_asm.inc")).read()
