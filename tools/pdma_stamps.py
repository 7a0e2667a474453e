#!/usr/bin/env python3
"""Diagnostic: where do the cycles of conv3_pdma*_kernel go?  Needs the stamps build of the library:

    make -C tiaozhanbei_unet_amd/csrc stamps      (-> libunet_hip_stamps.so, s_memtime stamps around every tap phase)
    UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_stamps.so python tools/pdma_stamps.py 32 512 512 32 32

Prints, per wave class, the mean cycles per tap spent in: the counted vmcnt wait, the barrier, DMA issue, fragment
reads + MFMAs; and the epilogue cycles per work item.  (Stamps cost ~10 % themselves: read ratios, not absolutes.)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiaozhanbei_unet_amd import _lib as L, ops  # noqa: E402


def main():
    n, ci, co, h, w = map(int, sys.argv[1:6])
    dev = torch.device("cuda:0")
    lib = L.lib()
    handle = C.CDLL(L.LIB_PATH)
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    handle.unet_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
    dt = torch.bfloat16
    x = torch.randn(n, ci, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    y = ops._nhwc_empty(n, co, h, w, dt, dev)
    wp = ops.pack_weight(wt, L.PACK_CONV_FWD, co, ci, dt)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    V = ops._views
    for _ in range(5):
        L.check(lib.unet_conv3x3(L.UNET_BF16, n, h, w, V([(x, 0, 0), None]), C.c_void_p(wp.data_ptr()), co,
                                 V([(y, 0, 0), None]), co, 0, 0, st), "fwd")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.check(lib.unet_conv3x3(L.UNET_BF16, n, h, w, V([(x, 0, 0), None]), C.c_void_p(wp.data_ptr()), co,
                                 V([(y, 0, 0), None]), co, 0, 0, st), "fwd")
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    d = dbg.view(256, 8, 8).cpu().double()
    taps = d[:, :, 4].clamp(min=1)
    pp = os.environ.get("UNET_PDMA_PP", "0") in "12"
    names = ["reads+dma issue", "vmcnt+lgkm wait", "barrier(L)", "mfma", "barrier(C)"] if pp else \
        ["vmcnt wait", "barrier", "dma issue", "reads+mfma"]
    cols = [0, 1, 2, 3, 6] if pp else [0, 1, 2, 3]
    clock = float(d[:, :, 7].median()) / 2 ** 20 * 0.1
    print(f"conv fwd n={n} {ci}->{co} {h}x{w} {'ping-pong' if pp else 'lock-step'}: {us:.1f} us/launch (stamped build), "
          f"taps/wave {float(taps.mean()):.0f}, in-kernel clock {clock:.2f} GHz")
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        per = [float((d[:, sl, i] / taps[:, sl]).mean()) for i in cols]
        tot = sum(per)
        print(f"  {grp}: " + "  ".join(f"{nm} {v:7.0f} ({100 * v / tot:4.1f}%)" for nm, v in zip(names, per)) +
              f"   total/tap {tot:.0f} cyc;  " + (f"of which fragment-read issue {float((d[:, sl, 5] / taps[:, sl]).mean()):.0f}/tap" if pp
                                                 else f"epilogue/launch {float(d[:, sl, 5].mean()):.0f} cyc"))


if __name__ == "__main__":
    main()
