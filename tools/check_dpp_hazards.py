#!/usr/bin/env python3
"""Static checks of hand-scheduled inline asm.

(1) The inline-asm DPP reductions (row16_sum_n in csrc/igemm.hip, first.hip): hipcc pads nothing inside an asm
statement, so a `v_add_f32_dpp` must not read a VGPR that one of the two preceding instructions wrote (2 wait states).
(2) The inline-asm `buffer_load_dwordx4 ... offen` y loads of the fused BatchNorm-backward epilogues (conv3_ws16_kernel<.., 2>,
conv3_ws_kernel<.., 2>, convt_dgrad_ws_kernel<64, true>): their only ordering against use is a hand-counted `s_waitcnt vmcnt(n)`
dozens of MFMAs later (hipcc does not count LDS-DMAs, so its own wait would drain them).  The compiler treats the destination as
defined right after the asm statement: no instruction between a load and the next `s_waitcnt vmcnt` may read, write, copy or spill
any of its destination VGPRs (a register-allocator change would otherwise read them before the data lands -- silently
wrong ReLU masks and BatchNorm-backward sums).
(3) Kernels with hand-counted `s_waitcnt vmcnt(n)` (the LDS-DMA convolution / weight-gradient kernels) must not spill: scratch traffic
would join the counted stream (scan_no_scratch).
Compiles the sources to assembly (hipcc -S, gfx950, no GPU needed) and scans.

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


def all_vregs(text):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"(?<![a-z_\[:0-9])v(\d+)(?![\d:\]])", text):
        out.add(int(m.group(1)))
    return out


def scan_inline_loads(path, kernel_patterns):
    """-> (loads checked, violations) over the kernels whose mangled name matches one of the patterns."""
    n = bad = 0
    name, body = None, []
    kernels = {}
    for line in open(path):
        t = line.strip()
        m = re.match(r"^(_Z[\w]+):", t)
        if m:
            name, body = m.group(1), []
            kernels[name] = body
        elif name and t and not t.startswith(";") and not t.startswith("."):
            body.append(t.split(";")[0].strip())
    for kname, ins in kernels.items():
        if not any(pat in kname for pat in kernel_patterns):
            continue
        for i, t in enumerate(ins):
            if not (t.startswith("buffer_load_dwordx4") and " offen" in t and " lds" not in t):
                continue
            dst = regs(t.split()[1])
            if len(dst) != 4:
                continue
            n += 1
            for u in ins[i + 1:]:
                if u.startswith("s_waitcnt") and "vmcnt" in u:
                    break
                if u.endswith(":"):
                    continue
                if all_vregs(u) & dst and not (u.startswith("buffer_load_dwordx4") and not (regs(u.split()[1]) & dst) and not (all_vregs(" ".join(u.split()[2:])) & dst)):
                    bad += 1
                    print(f"EARLY USE in {kname}: `{t}` ... `{u}`")
                    break
            else:
                bad += 1
                print(f"NO WAIT after `{t}` in {kname}")
    return n, bad


def scan_no_scratch(path, kernel_patterns):
    """-> (kernels checked, kernels with scratch traffic).  Kernels whose waits are hand-counted `s_waitcnt vmcnt(n)` over an exact
    per-tile sequence of DMAs, loads and stores must not spill: a scratch_load / scratch_store is one more vector-memory operation
    in that sequence, and a count that is too lax lets a wave read an LDS patch whose DMA has not landed (round 3: a 14-spill
    build of conv3_ws16_kernel<false, 2> passed every small-frame test and produced garbage gradients at 32 tiles per block)."""
    n = bad = 0
    name = None
    for line in open(path):
        t = line.strip()
        m = re.match(r"^(_Z[\w]+):", t)
        if m:
            name = m.group(1) if any(pat in m.group(1) for pat in kernel_patterns) else None
            n += name is not None
            seen = False
        elif name and t.startswith("scratch_") and not seen:
            seen = True
            bad += 1
            print(f"SPILL in {name}: `{t.split(';')[0].strip()}`")
    return n, bad


NO_SPILL_KERNELS = ("conv3_ws16_kernel", "conv3_pdma", "conv3_pp", "wgrad16_kernelILi2ELi1E")

Y_LOAD_KERNELS = ("conv3_ws16_kernelILb0ELi2E", "conv3_ws_kernelILb0ELi2E", "convt_dgrad_ws_kernelILi64ELb1E")


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
            if src == "igemm.hip":
                n2, bad2 = scan_inline_loads(out, Y_LOAD_KERNELS)
                print(f"{src}: {n2} inline-asm y loads, {bad2} used before their hand-counted wait")
                total_bad += bad2
                if n2 < 12:
                    print("expected at least 12 inline-asm y loads (3 kernel families x 4)")
                    total_bad += 1
                n3, bad3 = scan_no_scratch(out, NO_SPILL_KERNELS)
                print(f"{src}: {n3} kernels with hand-counted vmcnt waits, {bad3} with scratch traffic")
                total_bad += bad3 + (n3 < 12)
        out = os.path.join(tmp, "wgrad.hip.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                               "--cuda-device-only", os.path.join(ROOT, "tiaozhanbei_unet_amd", "csrc", "wgrad.hip"), "-o", out])
        n3, bad3 = scan_no_scratch(out, NO_SPILL_KERNELS)
        print(f"wgrad.hip: {n3} kernels with hand-counted vmcnt waits, {bad3} with scratch traffic")
        total_bad += bad3 + (n3 < 2)
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
