#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound streaming passes of the step through the C-ABI, next to torch's own elementwise kernels on
the same tensors (what this card gives a plain copy / add): achieved TB/s = algorithmic bytes / time.

    python tools/bench_stream.py [--n 32] [--iters 30] [--ab 0,1 --abvar UNET_EW_VAR]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiaozhanbei_unet_amd import _lib as L, ops  # noqa: E402

LEVELS = [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--ab", default="")
    ap.add_argument("--abvar", default="UNET_EW_VAR")
    ap.add_argument("--levels", default="0,1,2,3,4")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    dt, DT = torch.bfloat16, L.UNET_BF16
    variants = a.ab.split(",") if a.ab else [None]
    for li in [int(v) for v in a.levels.split(",")]:
        c, s = LEVELS[li]
        n, h, w = a.n, s, s
        pixels = n * h * w
        y = torch.randn(n, c, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        g = torch.randn(n, c, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        out = torch.empty_like(y)
        coef = torch.rand(4, c, device=dev) + 0.5
        gamma = torch.rand(c, device=dev) + 0.5
        dgb = torch.empty(2, c, device=dev)
        nparts = 256
        part = torch.randn(nparts, 2, c, device=dev)
        ws = torch.empty(3 * c * 4, dtype=torch.uint8, device=dev)
        pooled = ops._nhwc_empty(n, c, h // 2, w // 2, dt, dev)
        gp = torch.randn(n, c, h // 2, w // 2, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        ppart = torch.empty(lib.unet_bn_relu_pool_max_parts() * 2 * c, device=dev)
        npp = C.c_int32(0)
        el = pixels * c
        cases = {
            "torch copy (1R 1W)": (lambda: out.copy_(y), 4 * el),
            "torch add (2R 1W)": (lambda: torch.add(y, g, out=out), 6 * el),
            "torch add in place (2R 1W)": (lambda: g.add_(y), 6 * el),
            "bn_relu_apply (1R 1W)": (lambda: L.check(lib.unet_bn_relu_apply(DT, p(y), pixels, c, p(coef[2]), p(coef[3]), p(out), st), "apply"), 4 * el),
            "bn_bwd_premasked (2R 1W, in place)": (lambda: L.check(lib.unet_bn_bwd_premasked(
                DT, p(g), p(y), pixels, c, p(gamma), p(coef[0]), p(coef[1]), p(part), nparts, p(dgb[0]), p(dgb[1]), p(g), p(ws),
                ws.numel(), st), "premasked"), 6 * el),
            "bn_relu_pool_fwd (1R 1.25W)": (lambda: L.check(lib.unet_bn_relu_pool_fwd(DT, p(y), n, h, w, c, p(coef[2]), p(coef[3]), p(out), p(pooled), st), "poolf"), int(4.5 * el)),
            "bn_relu_pool_bwd (2.25R 1W)": (lambda: L.check(lib.unet_bn_relu_pool_bwd(DT, p(y), p(gp), p(g), n, h, w, c, p(coef[2]), p(coef[3]), p(coef[0]), p(out), p(ppart),
                                                                                    C.byref(npp), st), "poolb"), int(6.5 * el)),
        }
        print(f"--- N={n} C={c} {h}x{w}: {2 * el / 2**20:.0f} MiB per bf16 tensor", flush=True)
        for name, (fn, nbytes) in cases.items():
            best = {v: 1e9 for v in variants}
            for rnd in range(3 if a.ab else 1):
                for v in variants:
                    if v is not None:
                        os.environ[a.abvar] = v
                        lib.unet_tuning_reload()
                    best[v] = min(best[v], timeit(fn, a.iters))
            txt = "  ".join((f"[{v}] " if v is not None else "") + f"{ms * 1e3:7.1f} us {nbytes / ms / 1e9:5.2f} TB/s" for v, ms in best.items())
            print(f"  {name:38s} {txt}", flush=True)


if __name__ == "__main__":
    main()
