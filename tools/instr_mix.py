#!/usr/bin/env python3
"""Static instruction mix of the hot loop of a kernel in an llvm-objdump -d listing.

    llvm-objdump -d <gfx950 code object> > k.dis ; python tools/instr_mix.py k.dis k_msm_accumulate

Finds the kernel's largest backward branch (the main loop), counts the VALU instructions between its target and the
branch, and splits them into the two issue classes tools/ubench_valu.hip measures on gfx950: the 64-bit / integer-multiply
class (4 cycles per wave-instruction and SIMD: v_mad_u64_u32, v_mad_i64_i32, v_mul_lo/hi_u32, v_lshl_add_u64, 64-bit shifts,
carry adds, FP64) and the plain 32-bit class (2 cycles).  Used for bench.py's roofline.alu (profiles/r2_instr_mix.txt)."""
import re
import sys

SLOW = re.compile(r"^v_(mad_u64_u32|mad_i64_i32|mul_lo_u32|mul_hi_u32|mul_hi_i32|lshl_add_u64|ashrrev_i64|lshrrev_b64|lshlrev_b64|addc_co_u32|subb_co_u32|"
                  r"subbrev_co_u32|add_co_u32|sub_co_u32|subrev_co_u32|fma_f64|add_f64|mul_f64|cvt_f64|mad_u32_u24|mul_u32_u24|cmp_.*_[ui]64)")


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    want_sub = sys.argv[3] if len(sys.argv) > 3 else ""
    lines = open(path).read().splitlines()
    start = end = None
    for i, l in enumerate(lines):
        m = re.match(r"^([0-9a-f]+) <(.*)>:$", l)
        if m:
            if start is not None and end is None:
                end = i
            if kernel in m.group(2) and want_sub in m.group(2) and start is None:
                start = i
    end = end or len(lines)
    body = lines[start + 1:end]
    addr = {}
    ins = []
    for l in body:
        m = re.match(r"^\s+(\S+)\s+(.*?)//\s*([0-9A-Fa-f]+):", l)
        if m:
            a = int(m.group(3), 16)
            addr[a] = len(ins)
            ins.append((a, m.group(1), m.group(2)))
    best = None
    for idx, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch"):
            # signed 16-bit dword offset relative to the next instruction
            mm = re.search(r"(\d+)\s*$", args.strip())
            if not mm:
                continue
            off = int(mm.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt < a and tgt in addr:
                span = idx - addr[tgt]
                if best is None or span > best[0]:
                    best = (span, addr[tgt], idx)
    span, lo, hi = best
    loop = ins[lo:hi + 1]
    valu = [op for _, op, _ in loop if op.startswith("v_")]
    slow = [op for op in valu if SLOW.match(op)]
    counts = {}
    for op in valu:
        counts[op] = counts.get(op, 0) + 1
    print("kernel %s: main loop %d instructions, %d VALU, %d in the 4-cycle class (%.1f %%), %d in the 2-cycle class" %
          (kernel, len(loop), len(valu), len(slow), 100.0 * len(slow) / len(valu), len(valu) - len(slow)))
    print("  others: %d SALU/branch, %d VMEM, %d LDS/SMEM-wait" % (sum(op.startswith("s_") for _, op, _ in loop),
          sum(op.startswith(("global_", "buffer_", "scratch_", "flat_")) for _, op, _ in loop), sum(op.startswith("ds_") for _, op, _ in loop)))
    for op, c in sorted(counts.items(), key=lambda kv: -kv[1])[:14]:
        print("    %-22s %6d %s" % (op, c, "(4-cycle)" if SLOW.match(op) else ""))


if __name__ == "__main__":
    main()
