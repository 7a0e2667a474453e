#!/usr/bin/env python3
"""Micro-benchmark of single hot-path operators through the C-ABI (for kernel tuning / rocprofv3 --pmc).

    python tools/bench_layer.py conv 32 512 512 32 32 [--iters 20] [--dtype bf16] [--op fwd|dgrad|wgrad]
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
    ap.add_argument("kind", choices=["conv", "convt", "first"])
    ap.add_argument("n", type=int); ap.add_argument("cin", type=int); ap.add_argument("cout", type=int)
    ap.add_argument("h", type=int); ap.add_argument("w", type=int)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--op", default="fwd", choices=["fwd", "fwdstats", "dgrad", "wgrad"])
    ap.add_argument("--ab", default="", help="comma list of values of --abvar to A/B interleaved in one process")
    ap.add_argument("--abvar", default="UNET_CONV_IMPL")
    ap.add_argument("--acc", type=int, default=0, help="accumulate bit mask of the conv dgrad (dst += result)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda:0")
    n, ci, co, h, w = a.n, a.cin, a.cout, a.h, a.w
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    x = torch.randn(n, ci, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    if a.kind == "first":
        xf = torch.randn(n, ci, h, w, device=dev)
        wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        gy = torch.randn(n, co, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        y = ops._nhwc_empty(n, co, h, w, dt, dev)
        dw = torch.empty_like(wt)
        cap = lib.unet_conv3x3_stats_max_parts(n, h, w)
        part = torch.empty(cap * 2 * co, device=dev)
        nparts = C.c_int32(0)
        need = lib.unet_conv3x3_first_wgrad_workspace(n, h, w)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        flops = 2.0 * n * h * w * co * ci * 9
        if a.op == "fwd":
            run = lambda: L.check(lib.unet_conv3x3_first_stats(n, h, w, p(xf), ci, p(wt), p(y), p(part), C.byref(nparts), st), "first fwd")
        else:
            run = lambda: L.check(lib.unet_conv3x3_first_wgrad(n, h, w, p(xf), ci, p(gy), p(dw), p(ws), need, st), "first wgrad")
    elif a.kind == "conv":
        wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
        gy = torch.randn(n, co, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        y = ops._nhwc_empty(n, co, h, w, dt, dev)
        dx = ops._nhwc_empty(n, ci, h, w, dt, dev)
        dw = torch.empty_like(wt)
        wp = ops.pack_weight(wt, L.PACK_CONV_FWD, co, ci, dt)
        wpd = ops.pack_weight(wt, L.PACK_CONV_DGRAD, ci, co, dt)
        need = lib.unet_conv3x3_wgrad_workspace(n, h, w, ci, co)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        flops = 2.0 * n * h * w * co * ci * 9
        V = lambda items: ops._views(items)
        if a.op == "fwd":
            run = lambda: L.check(lib.unet_conv3x3(ops._DT[dt], n, h, w, V([(x, 0, 0), None]), p(wp), co,
                                                   V([(y, 0, 0), None]), co, 0, 0, st), "fwd")
        elif a.op == "fwdstats":                      # forward + BatchNorm partial sums in the epilogue (training)
            cap = lib.unet_conv3x3_stats_max_parts(n, h, w)
            part = torch.empty(cap * 2 * co, device=dev)
            nparts = C.c_int32(0)
            run = lambda: L.check(lib.unet_conv3x3_stats(ops._DT[dt], n, h, w, V([(x, 0, 0), None]), p(wp), co, p(y),
                                                         p(part), C.byref(nparts), st), "fwd+stats")
        elif a.op == "dgrad":
            run = lambda: L.check(lib.unet_conv3x3(ops._DT[dt], n, h, w, V([(gy, 0, 0), None]), p(wpd), ci,
                                                   V([(dx, 0, 0), None]), ci, a.acc, 1, st), "dgrad")
        else:
            run = lambda: L.check(lib.unet_conv3x3_wgrad(ops._DT[dt], n, h, w, V([(x, 0, 0), None]), p(gy), co, p(dw),
                                                         ci, p(ws), need, st), "wgrad")
    else:
        wt = torch.randn(ci, co, 2, 2, device=dev) * 0.05
        b = torch.zeros(co, device=dev)
        y = ops._nhwc_empty(n, co, 2 * h, 2 * w, dt, dev)
        wp = ops.pack_weight(wt, L.PACK_CONVT_FWD, co, ci, dt)
        flops = 2.0 * n * h * w * co * ci * 4
        wpd = ops.pack_weight(wt, L.PACK_CONVT_DGRAD, ci, co, dt)
        gy = torch.randn(n, co, 2 * h, 2 * w, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        dx = ops._nhwc_empty(n, ci, h, w, dt, dev)
        dw = torch.empty_like(wt)
        db = torch.empty(co, device=dev)
        need = lib.unet_convt2x2_wgrad_workspace(n, h, w, ci, co)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        if a.op == "fwd":
            run = lambda: L.check(lib.unet_convt2x2_fwd(ops._DT[dt], n, h, w, p(x), ci, p(wp), p(b), p(y), co, st), "convt")
        elif a.op == "dgrad":
            run = lambda: L.check(lib.unet_convt2x2_dgrad(ops._DT[dt], n, h, w, p(gy), co, p(wpd), p(dx), ci, st), "convt dgrad")
        else:
            run = lambda: L.check(lib.unet_convt2x2_wgrad(ops._DT[dt], n, h, w, p(x), ci, p(gy), co, p(dw), p(db),
                                                          p(ws), need, st), "convt wgrad")
    variants = a.ab.split(",") if a.ab else [None]
    best = {v: 1e9 for v in variants}
    for rnd in range(4 if a.ab else 1):
        for v in variants:
            if v is not None:
                os.environ[a.abvar] = v
                lib.unet_tuning_reload()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / a.iters)
    if a.ab and a.kind != "first":
        outs = {}
        for v in variants:
            os.environ[a.abvar] = v
            lib.unet_tuning_reload()
            tgt = {"fwd": y, "fwdstats": y, "dgrad": dx, "wgrad": dw}[a.op]
            tgt.zero_()
            run()
            torch.cuda.synchronize()
            outs[v] = tgt.float().clone()
        ref = outs[variants[0]]
        for v in variants[1:]:
            print(f"   check {a.abvar}={v} vs {variants[0]}: max|diff| {float((outs[v] - ref).abs().max()):.3e} "
                  f"(|ref|max {float(ref.abs().max()):.3e})", flush=True)
    for v, ms in best.items():
        tag = "" if v is None else f" impl={v}"
        print(f"{a.kind} {a.op} n={n} {ci}->{co} {h}x{w} {a.dtype}{tag}: {ms * 1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
