#!/usr/bin/env python3
"""One-GPU probe of what CU sharing with a resident collective costs the persistent kernels, and what
unet_set_reserved_cus() buys back (no multi-GPU node is available to the builder).

While a stand-in for RCCL's all-reduce kernels -- k workgroups that each hold 48 KiB of LDS and spin on a side stream
(unet_debug_spin) -- is resident, a training step of the benchmark configuration is timed with the launchers sized for
all 256 CUs (reserved 0) and for 256 - r CUs.  A CU that hosts a spinner has no room for a conv block's 125-160 KiB of
LDS; with a static one-block-per-CU partition the displaced blocks run as a second round.

    python tools/cu_share_probe.py [--steps 10]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiaozhanbei_unet_amd as P  # noqa: E402
from tiaozhanbei_unet_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--fine", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.lib()
    torch.manual_seed(0)
    model = P.AnomalyUNet(3, precision="bf16").to(dev).train()
    crit = P.CombinedLoss()
    opt = P.get_optimizer(model, "adam", 1e-3, 1e-4)
    g = torch.Generator(device=dev).manual_seed(42)
    x = torch.randn(a.batch, 3, 256, 256, device=dev, generator=g)
    m = (torch.rand(a.batch, 1, 256, 256, device=dev, generator=g) < 0.02).float()
    side = torch.cuda.Stream(device=dev)

    def step():
        r, am = model(x)
        loss = crit(r, am, x, m)["total_loss"]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    def timed(spinners, lds, reserved):
        L.check(lib.unet_set_reserved_cus(reserved), "reserve")
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        if spinners:           # resident for the whole timed region (2 s is far longer than it)
            L.check(lib.unet_debug_spin(spinners, lds, 2_000_000, C.c_void_p(side.cuda_stream)), "spin")
            time.sleep(0.01)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.current_stream(dev).synchronize()
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        torch.cuda.synchronize()                                  # (waits for the spinners to expire)
        return round(dt, 3)

    rows = []
    # (spinning blocks, LDS each): 0-LDS spinners share a CU with a conv block -- they isolate the cost of a second
    # active queue from the cost of displaced blocks
    cases = ((0, 0), (1, 0), (8, 0), (8, 48 * 1024), (16, 48 * 1024), (32, 48 * 1024))
    reserves = (0, 8, 16, 32)
    if a.fine:          # where is the threshold?  (8 spinners; 16 / 48 KiB each)
        cases, reserves = ((8, 16 * 1024), (8, 48 * 1024)), (0, 8, 16, 24, 32, 40, 64)
    for k, lds in cases:
        row = {"spinning_blocks": k, "lds_each": lds, "ms_per_step": {}}
        for r in reserves:
            row["ms_per_step"][f"reserved_{r}"] = timed(k, lds, r)
        rows.append(row)
        print(json.dumps(row), flush=True)
    L.check(lib.unet_set_reserved_cus(0), "reserve")
    print(json.dumps({"probe": "cu_share", "batch": a.batch, "steps": a.steps, "cu_budget_reserved_0": int(lib.unet_get_cu_budget()),
                      "rows": rows}))


if __name__ == "__main__":
    main()
