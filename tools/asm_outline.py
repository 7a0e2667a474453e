#!/usr/bin/env python3
"""Outline of a kernel in hipcc -S output: positions of MFMAs, barriers, LDS-DMAs, stores, scratch (spill) traffic.
    python tools/asm_outline.py /tmp/igemm.s conv3_ws16_kernelILb0ELi2E"""
import sys

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    i = s.index("\n_ZN", s.index(name) - 200 if s.index(name) > 200 else 0)
    i = s.index(name)
    i = s.rindex("\n", 0, i) + 1
    j = s.index(".Lfunc_end", i)
    body = [l for l in s[i:j].split("\n") if l.strip() and not l.strip().startswith(";")]
    pos = lambda pred: [k for k, l in enumerate(body) if pred(l)]
    mf = pos(lambda l: "v_mfma" in l)
    print(name, "instructions", len(body), "mfma", len(mf), "first/last", mf[0], mf[-1])
    print("   barriers", pos(lambda l: "s_barrier" in l))
    print("   scratch ", [(k, body[k].split()[0]) for k in pos(lambda l: "scratch_" in l)])
    print("   lds-dma ", pos(lambda l: "buffer_load" in l and " lds" in l))
    print("   stores  ", pos(lambda l: "buffer_store" in l))
    print("   loads   ", pos(lambda l: "buffer_load" in l and " lds" not in l))
    print("   vmcnt   ", [(k, body[k].split("vmcnt")[1][:4]) for k in pos(lambda l: "s_waitcnt" in l and "vmcnt" in l)])
