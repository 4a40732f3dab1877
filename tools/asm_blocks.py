#!/usr/bin/env python3
"""Basic blocks of one kernel in a device assembly listing (hipcc --offload-device-only -S): instructions, multiply-adds, scratch
accesses, v_mov and inline-asm statements per block; with a block name, the block's instructions outside its asm statements.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --offload-device-only -S -o g1.s msm_g1.hip
    python tools/asm_blocks.py g1.s k_msm_accumulate [.LBB18_140 ...]

Used to read what the compiler put AROUND the generated bodies of madd_asm.inc / mont_asm.inc (copies at the joins, spills)."""
import re
import sys


def kernel_body(path, kernel):
    lines = open(path).read().splitlines()
    s = next(i for i, l in enumerate(lines) if kernel in l and l.rstrip().split(";")[0].rstrip().endswith(":") and not l.startswith("\t"))
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[s + 1:e]


def blocks_of(body):
    blocks = [{"name": "entry", "ins": [], "asm": []}]
    in_asm = False
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append({"name": m.group(1), "ins": [], "asm": []})
            continue
        st = l.strip()
        if "ASMSTART" in st:
            in_asm = True
            blocks[-1]["asm"].append(0)
            continue
        if "ASMEND" in st:
            in_asm = False
            blocks[-1]["ins"].append("[asm statement, %d instructions]" % blocks[-1]["asm"][-1])
            continue
        if not st or st.startswith(";") or st.startswith("."):
            continue
        ins = st.split(";")[0].strip()
        if in_asm:
            if not ins.endswith(":"):
                blocks[-1]["asm"][-1] += 1
        else:
            blocks[-1]["ins"].append(ins)
    return blocks


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    want = sys.argv[3:]
    blocks = blocks_of(kernel_body(path, kernel))
    for b in blocks:
        outside = [x for x in b["ins"] if not x.startswith("[asm")]
        n_asm = sum(b["asm"])
        if want:
            if b["name"] in want:
                print("====", b["name"], "outside asm:", len(outside), "inside:", n_asm)
                for x in b["ins"]:
                    print("   ", x)
            continue
        scratch = sum(1 for x in outside if x.startswith("scratch_"))
        movs = sum(1 for x in outside if x.startswith("v_mov") or x.startswith("v_pk_mov") or x.startswith("v_accvgpr"))
        valu = sum(1 for x in outside if x.startswith("v_"))
        if len(outside) + n_asm > 40 or scratch:
            print("%-12s outside asm %5d (VALU %4d, v_mov %3d, scratch %3d)   asm statements %s" % (b["name"], len(outside), valu, movs, scratch, b["asm"]))


if __name__ == "__main__":
    main()
