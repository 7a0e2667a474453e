#!/usr/bin/env python3
"""Diagnostic (stamps build): cycles per tile of conv3_ws16_kernel's phases, per wave class.
    UNET_HIP_LIB=tiaozhanbei_unet_amd/libunet_hip_stamps.so python tools/ws16_stamps.py 32 64 64 256 256 [stats]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiaozhanbei_unet_amd import _lib as L, ops  # noqa: E402


def main():
    n, ci, co, h, w = map(int, sys.argv[1:6])
    stats = len(sys.argv) > 6
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
    cap = lib.unet_conv3x3_stats_max_parts(n, h, w)
    part = torch.empty(cap * 2 * co, device=dev)
    nparts = C.c_int32(0)
    p = lambda t: C.c_void_p(t.data_ptr())
    if stats:
        run = lambda: L.check(lib.unet_conv3x3_stats(L.UNET_BF16, n, h, w, V([(x, 0, 0), None]), p(wp), co, p(y), p(part),
                                                     C.byref(nparts), st), "fwd+stats")
    else:
        run = lambda: L.check(lib.unet_conv3x3(L.UNET_BF16, n, h, w, V([(x, 0, 0), None]), p(wp), co, V([(y, 0, 0), None]),
                                               co, 0, 0, st), "fwd")
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
    d = dbg.view(256, 8, 8).cpu().double()
    tiles = d[:, :, 7].clamp(min=1)
    clock = float(d[:, :, 6].median()) / 2 ** 20 * 0.1
    names = ["vmcnt wait", "barrier", "top: dma / deferred stores / geometry", "mfma loop", "late dma", "epilogue + stores"]
    print(f"conv {'fwd+stats' if stats else 'fwd'} n={n} {ci}->{co} {h}x{w} ws16 UNET_WS_STG={os.environ.get('UNET_WS_STG', 'default')}: "
          f"{us:.1f} us/launch (stamped), tiles/block {float(tiles.mean()):.0f}, in-kernel clock {clock:.2f} GHz")
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        per = [float((d[:, sl, i] / tiles[:, sl]).mean()) for i in range(6)]
        print(f"  {grp}: " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, per)) + f"   total/tile {sum(per):.0f} cyc")


if __name__ == "__main__":
    main()
