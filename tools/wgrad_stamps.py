#!/usr/bin/env python3
"""Diagnostic (stamps build): cycles per pixel tile of wgrad_dma_kernel's phases.
    UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_stamps.so python tools/wgrad_stamps.py 32 256 256 64 64"""
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
    dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
    handle.unet_debug_set_buffer_wgrad(C.c_void_p(dbg.data_ptr()))
    dt = torch.bfloat16
    x = torch.randn(n, ci, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, co, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    need = lib.unet_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    run = lambda: L.check(lib.unet_conv3x3_wgrad(L.UNET_BF16, n, h, w, ops._views([(x, 0, 0), None]), p(gy), co, p(dw), ci,
                                                 p(ws), need, st), "wgrad")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    d = dbg.view(512, 4, 8).cpu().double()
    d = d[d[:, 0, 4] > 0]
    tiles = d[:, :, 4].clamp(min=1)
    clock = float(d[:, :, 5].median()) / 2 ** 20 * 0.1
    names = ["vmcnt(0) wait", "barrier", "dma issue (10/wave)", "fragment reads + 72 mfma"]
    per = [float((d[:, :, i] / tiles).mean()) for i in range(4)]
    print(f"wgrad n={n} {ci}->{co} {h}x{w}: {us:.1f} us/launch incl. reduce (stamped), blocks {d.shape[0]}, tiles/block "
          f"{float(tiles.mean()):.1f}, in-kernel clock {clock:.2f} GHz")
    print("  per tile: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, per)) + f"   total {sum(per):.0f} cyc (ideal MFMA 2304)")


if __name__ == "__main__":
    main()
