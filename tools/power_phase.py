#!/usr/bin/env python3
"""One PHASE of the training step in a loop for a few seconds, so that `rocm-smi --showpower --showclocks` sampled beside it
reads that phase's steady state (tools/power_phases.sh): what do the MFMA kernels and the BatchNorm passes draw on their
own?  ("power-limited" as a number.)

    python tools/power_phase.py conv512 | wgrad512 | conv64 | bn_apply | bn_bwd | idle  [--seconds 6]
"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiaozhanbei_unet_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("phase")
    ap.add_argument("--seconds", type=float, default=6.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    dt, DT = torch.bfloat16, L.UNET_BF16
    V = ops._views
    if a.phase in ("conv512", "wgrad512", "conv64", "conv128", "conv128_64", "wgrad64", "wgrad128_64", "conv1024_512"):
        n, ci, co, h = {"conv512": (32, 512, 512, 32), "wgrad512": (32, 512, 512, 32), "conv64": (32, 64, 64, 256),
                        "conv128": (32, 128, 128, 128), "conv128_64": (32, 128, 64, 256), "wgrad64": (32, 64, 64, 256),
                        "wgrad128_64": (32, 128, 64, 256), "conv1024_512": (32, 1024, 512, 32)}[a.phase]
        # post-ReLU-like operands (half zeros), as in the step
        x = torch.randn(n, ci, h, h, device=dev).clamp_min(0).to(dt).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(n, co, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        y = ops._nhwc_empty(n, co, h, h, dt, dev)
        wp = ops.pack_weight(wt, L.PACK_CONV_FWD, co, ci, dt)
        dw = torch.empty_like(wt)
        need = lib.unet_conv3x3_wgrad_workspace(n, h, h, ci, co)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        flops = 2.0 * n * h * h * co * ci * 9
        if a.phase.startswith("wgrad"):
            run = lambda: L.check(lib.unet_conv3x3_wgrad(DT, n, h, h, V([(x, 0, 0), None]), p(gy), co, p(dw), ci, p(ws), need, st), "wgrad")
        else:
            run = lambda: L.check(lib.unet_conv3x3(DT, n, h, h, V([(x, 0, 0), None]), p(wp), co, V([(y, 0, 0), None]), co, 0, 0, st), "fwd")
    elif a.phase in ("bn_apply", "bn_bwd"):
        n, c, h = 32, 64, 256
        y = torch.randn(n, c, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        g = torch.randn(n, c, h, h, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        out = torch.empty_like(y)
        coef = torch.rand(4, c, device=dev) + 0.5
        gamma = torch.rand(c, device=dev) + 0.5
        dgb = torch.empty(2, c, device=dev)
        part = torch.randn(256, 2, c, device=dev) * 1e-3
        ws = torch.empty(3 * c * 4, dtype=torch.uint8, device=dev)
        pixels = n * h * h
        flops = 0.0
        if a.phase == "bn_apply":
            run = lambda: L.check(lib.unet_bn_relu_apply(DT, p(y), pixels, c, p(coef[2]), p(coef[3]), p(out), st), "apply")
        else:
            run = lambda: L.check(lib.unet_bn_bwd_premasked(DT, p(g), p(y), pixels, c, p(gamma), p(coef[0]), p(coef[1]), p(part), 256,
                                                            p(dgb[0]), p(dgb[1]), p(out), p(ws), ws.numel(), st), "premasked")
    elif a.phase == "idle":
        time.sleep(a.seconds)
        print("idle")
        return
    else:
        raise SystemExit("unknown phase")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.time()
    iters = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < a.seconds:
        for _ in range(200):
            run()
        iters += 200
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{a.phase}: {ms * 1e3:.1f} us per launch" + (f", {flops / ms / 1e9:.0f} TFLOP/s" if flops else ""), flush=True)


if __name__ == "__main__":
    main()
