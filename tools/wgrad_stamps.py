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
    new = os.environ.get("UNET_WGRAD_IMPL", "") in ("", "3") and (co % 128 == 0 or (co == 64 and ci % 128 == 0 and w > 16))
    d = (dbg.view(256, 8, 8) if new else dbg.view(512, 4, 8)).cpu().double()
    d = d[d[:, 0, 4] > 0]
    tiles = d[:, :, 4].clamp(min=1)
    clock = float(d[:, :, 5].median()) / 2 ** 20 * 0.1
    names = ["vmcnt(0) wait", "barrier", "dma issue", "fragment reads + mfma"]
    print(f"wgrad n={n} {ci}->{co} {h}x{w} ({'wgrad16_kernel' if new else 'wgrad_dma_kernel'}): {us:.1f} us/launch incl. reduce "
          f"(stamped), blocks {d.shape[0]}, tiles/block {float(tiles.mean()):.1f}, in-kernel clock {clock:.2f} GHz")
    groups = [("all waves", slice(None))] if not new else [("dY waves (DMA mid-tile)", slice(0, 4)), ("X waves (DMA at tile start)", slice(4, 8))]
    if new and co == 64:
        groups = [("dY waves", slice(0, 2)), ("X waves", slice(2, 8))]
    for label, sl in groups:
        per = [float((d[:, sl, i] / tiles[:, sl]).mean()) for i in range(4)]
        ideal = 2304
        print(f"  {label}: per tile: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, per)) +
              f"   total {sum(per):.0f} cyc (ideal MFMA {ideal} per wave, {2 * ideal} per SIMD)")
    if new:
        print(f"  partial-slab stores (drained): {float(d[:, :, 6].mean()):.0f} cyc; kernel body {float(d[:, :, 7].mean()):.0f} cyc "
              f"= {float(d[:, :, 7].mean()) / clock / 1e3:.1f} us at the in-kernel clock")


if __name__ == "__main__":
    main()
