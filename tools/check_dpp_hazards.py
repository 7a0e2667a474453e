#!/usr/bin/env python3
"""Static check of the inline-asm DPP reductions (row16_sum_n in csrc/igemm.hip, first.hip): hipcc pads nothing inside an asm
statement, so a `v_add_f32_dpp` must not read a VGPR that one of the two preceding instructions wrote (2 wait states).
Compiles the two sources to assembly (hipcc -S, gfx950, no GPU needed) and scans every DPP add.

    python tools/check_dpp_hazards.py          # exit code 1 on a hazard
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(op):
    op = op.rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", op)
    return {int(m.group(1))} if m else set()


def scan(path):
    ins = []
    for line in open(path):
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins.append(t)
    n = bad = 0
    for i, t in enumerate(ins):
        if not t.startswith("v_add_f32_dpp"):
            continue
        n += 1
        src = regs(t.split()[2])
        dist = 0
        for k in (1, 2):
            ops = ins[i - k].split()
            if ops[0] == "s_nop":
                dist += int(ops[1]) + 1
            else:
                if ops[0].startswith("v_") and len(ops) > 1 and regs(ops[1]) & src:
                    bad += 1
                    print("HAZARD:", ins[i - 2:i + 1])
                dist += 1
            if dist >= 2:
                break
    return n, bad


def main():
    total_bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("igemm.hip", "first.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                                   "--cuda-device-only", os.path.join(ROOT, "tiaozhanbei_unet_amd", "csrc", src), "-o", out])
            n, bad = scan(out)
            print(f"{src}: {n} DPP adds, {bad} hazards")
            total_bad += bad
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
