#!/usr/bin/env python3
"""Headline benchmark: training images/sec of AnomalyUNet 256x256 bs=32 (per GPU) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + CombinedLoss (MSE + focal) + backward + (RCCL gradient all-reduce when N>1) +
Adam, on synthetic 3x256x256 batches already resident in HBM (BASELINE.json configs[2], synthetic data).
Prints ONE JSON line (rank 0) with the driver's contract plus:
  roofline     -- the dominant kernel class, timed live with hipEvents on its own stream
                  (libunet_hip's per-class event brackets) against the dense MFMA peak of the dtype;
  cpu_baseline -- the CPU oracle (a port of the reference's train step) timed on this box's host cores
                  on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_IMG_256 = 158.637      # AnomalyUNet forward, 256x256
PEAK = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (weak scaling)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mask", default="bernoulli", choices=["bernoulli", "zeros"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2)
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: affinity mask, cgroup quota, and the GPU box's
    per-GPU CPU share (16) when the container still sees every core of the host."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n if n <= 32 else 16


def cpu_baseline(size, batch):
    """The oracle's train step (fwd + MSE/focal + bwd + Adam) on the host cores, bounded sample."""
    from oracle import unet_oracle as O, weights as W
    cores = host_cores()
    torch.set_num_threads(cores)
    state = dict(W.make_state(W.state_spec("anomaly_unet", 3, 1, False), 0))
    g = torch.Generator().manual_seed(42)
    image = torch.randn(batch, 3, size, size, generator=g)
    mask = (torch.rand(batch, 1, size, size, generator=g) < 0.02).float()
    opt = {}
    state, _ = O.train_step(state, opt, image, mask)          # warm-up (allocations, oneDNN primitives)
    t0 = time.perf_counter()
    steps = 0
    while steps < 8 and (steps < 2 or time.perf_counter() - t0 < 10.0):     # ~10 s of host work, at least 2 steps
        state, _ = O.train_step(state, opt, image, mask)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps ({dt:.1f} s, after 1 warm-up) of the oracle train step, AnomalyUNet "
                      f"{size}x{size} bs={batch} fp32, torch {torch.__version__} CPU"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()            # (rehearsals may stack ranks on one card)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("UNET_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd import ops
    from tiaozhanbei_unet_amd.ddp import DataParallel

    torch.manual_seed(0)                                    # identical init on every rank
    model = P.AnomalyUNet(3, precision=args.precision).to(dev)
    model.train()
    net = DataParallel(model) if world > 1 else model
    criterion = P.CombinedLoss()
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)

    g = torch.Generator(device=dev).manual_seed(42 + rank)
    n, s = args.batch, args.size
    images = torch.randn(n, 3, s, s, device=dev, generator=g)
    if args.mask == "bernoulli":
        masks = (torch.rand(n, 1, s, s, device=dev, generator=g) < 0.02).float()
    else:
        masks = torch.zeros(n, 1, s, s, device=dev)

    def step():
        recon, amap = net(images)
        losses = criterion(recon, amap, images, masks)
        optimizer.zero_grad(set_to_none=True)
        losses["total_loss"].backward()
        if world > 1:
            net.finish_gradients()
        optimizer.step()
        return losses

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(losses["total_loss"])

    # ---- roofline of the dominant kernel class: hipEvent brackets inside libunet_hip, 2 extra steps
    roof = None
    # (every rank runs the two extra steps -- they contain the gradient collectives; only rank 0 brackets them)
    # The two decoder branches normally run on two HIP streams; for these two steps they run on ONE stream so
    # that an event bracket times its kernel alone (concurrent kernels would inflate each other's brackets).
    model.two_streams = False
    if rank == 0:
        ops.prof_enable(True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        prof = ops.prof_collect()
        ops.prof_enable(False)
        mfma = {k: v for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0}
        if mfma:
            name = max(mfma, key=lambda k: mfma[k]["ms"])
            d = mfma[name]
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            peak = PEAK[args.precision]
            roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None,
                    "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // 2,
                    "measured": "hipEvent brackets on the launch stream, 2 single-stream steps after the timed region",
                    "per_class_ms_per_step": {k: round(v["ms"] / 2, 3) for k, v in prof.items() if v["launches"]},
                    "per_class_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
                                         for k, v in mfma.items()}}
    if world > 1:
        import torch.distributed as dist
        dist.barrier()

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / elapsed
        scale = (s * s) / 65536.0
        train_tflops = 3 * FWD_GFLOP_PER_IMG_256 * scale * value / 1e3
        out = {
            "metric": "training images/sec, AnomalyUNet 256x256 bs=32 per GPU", "value": round(value, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"AnomalyUNet 3x{s}x{s}, bs={args.batch}/GPU, MSE+focal loss, Adam "
                                   f"(BASELINE.json configs[2], synthetic randn images, {args.mask} masks)",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "accumulate": "fp32", "master_params": "fp32"},
            "model_tflops": round(train_tflops, 1),
            "mfma_frac_of_peak_whole_step": round(train_tflops / PEAK[args.precision] / world, 4),
            "final_loss": round(loss_val, 5),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(s, args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
