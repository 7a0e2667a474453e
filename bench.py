#!/usr/bin/env python3
"""Headline benchmark: training images/sec of AnomalyUNet 256x256 bs=32 (per GPU) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + CombinedLoss (MSE + focal) + backward + (RCCL gradient all-reduce when N>1) +
Adam, on synthetic 3x256x256 batches already resident in HBM (BASELINE.json configs[2], synthetic data).
Other single-GPU legs: `--ssim` (configs[2] as written: SSIM reconstruction head + focal) and
`--model unet --precision fp32 --batch 16` (configs[1]: UNet(3,1) seg-only, focal on sigmoid(logits)).

Timing: W warm-up steps, then `--blocks` (default 5) timed blocks of EXACTLY K steps, each bracketed by a barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks.  `value` / `ms_per_step` are those of the
MEDIAN block (boxes of the pool differ by up to 10 %, blocks on one box by ~1 %); `ms_per_step_p10/p90/blocks` say so.

Prints ONE JSON line (rank 0) with the driver's contract plus:
  roofline     -- the dominant KERNEL (most event time among the MFMA kernels), timed live with hipEvents on its own
                  stream (libunet_hip's per-launch brackets) against the dense MFMA peak of the dtype; `traffic` =
                  HBM bytes per launch of that kernel from the committed PMC passes of this same command
                  (profiles/r04_pmc_traffic.json, tools/pmc_traffic.sh; quoted only when that summary was taken from
                  THIS build -- same sha256 of csrc/ -- else null), beside the algorithmic bytes of the same launches;
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

FWD_GFLOP_PER_IMG_256 = {"anomaly_unet": 158.637, "unet": 96.335}      # SURVEY 8(d), forward, 256x256
PEAK = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA TFLOP/s, MI355X_MICROARCH.md chip table
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")


def csrc_sha16():
    """sha256 (first 16 hex digits) of the kernel sources: ties a committed PMC summary to the build it was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tiaozhanbei_unet_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps (median reported)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (weak scaling); default 32 (unet: 16)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--height", type=int, default=None, help="non-square frames (BASELINE configs[4]: --height 1408 --width 512)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--model", default="anomaly_unet", choices=["anomaly_unet", "unet"])
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32"], help="default bf16 (unet: fp32, configs[1])")
    ap.add_argument("--ssim", action="store_true", help="SSIM reconstruction head instead of MSE (configs[2] --use_ssim)")
    ap.add_argument("--mask", default="bernoulli", choices=["bernoulli", "zeros"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the two event-bracketed extra steps (PMC passes)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 32 if args.model == "anomaly_unet" else 16
    if args.precision is None:
        args.precision = "bf16" if args.model == "anomaly_unet" else "fp32"
    return args


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


def cpu_baseline(size, batch, model):
    """The oracle's train step (fwd + loss + bwd + Adam) on the host cores, bounded sample (~10-20 s)."""
    from oracle import unet_oracle as O, weights as W
    cores = host_cores()
    torch.set_num_threads(cores)
    kind = "anomaly_unet" if model == "anomaly_unet" else "unet"
    state = dict(W.make_state(W.state_spec(kind, 3, 1, False), 0))
    g = torch.Generator().manual_seed(42)
    image = torch.randn(batch, 3, size, size, generator=g)
    mask = (torch.rand(batch, 1, size, size, generator=g) < 0.02).float()
    opt = {}
    kw = dict(model=kind) if kind == "anomaly_unet" else dict(model="unet", recon_weight=0.0)
    state, _ = O.train_step(state, opt, image, mask, **kw)          # warm-up (allocations, oneDNN primitives)
    t0 = time.perf_counter()
    steps = 0
    while steps < 6 and (steps < 2 or time.perf_counter() - t0 < 12.0):     # ~12 s of host work, at least 2 steps
        state, _ = O.train_step(state, opt, image, mask, **kw)
        steps += 1
    dt = time.perf_counter() - t0
    name = "AnomalyUNet" if kind == "anomaly_unet" else "UNet(3,1) seg-only"
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} timed steps ({dt:.1f} s, after 1 warm-up) of the oracle train step, {name} "
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
    backend, reserved = None, 0
    if world > 1:
        import torch.distributed as dist
        from tiaozhanbei_unet_amd.ddp import configure_overlap
        backend = os.environ.get("UNET_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for rehearsals
        reserved = configure_overlap()                          # RCCL channel cap + CU budget of the persistent kernels
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd import ops
    from tiaozhanbei_unet_amd.ddp import DataParallel

    torch.manual_seed(0)                                    # identical init on every rank
    if args.model == "anomaly_unet":
        core = P.AnomalyUNet(3, precision=args.precision).to(dev)
        model = core
        criterion = P.CombinedLoss(recon_criterion=P.SSIMLoss() if args.ssim else None)
    else:
        from tiaozhanbei_unet_amd.train import _SegOnly
        core = P.UNet(3, 1, precision=args.precision).to(dev)
        model = _SegOnly(core)                             # (input as dummy reconstruction, sigmoid(logits))
        criterion = P.CombinedLoss(recon_weight=0.0, seg_weight=1.0)
    model.train()
    net = DataParallel(model) if world > 1 else model
    optimizer = P.get_optimizer(core, "adam", 1e-3, 1e-4)

    g = torch.Generator(device=dev).manual_seed(42 + rank)
    n, s = args.batch, args.size
    fh, fw = args.height or s, args.width or s
    images = torch.randn(n, 3, fh, fw, device=dev, generator=g)
    if args.mask == "bernoulli":
        masks = (torch.rand(n, 1, fh, fw, device=dev, generator=g) < 0.02).float()
    else:
        masks = torch.zeros(n, 1, fh, fw, device=dev)

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
    block_s = []
    for _ in range(max(1, args.blocks)):
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
        block_s.append(elapsed)
    loss_val = float(losses["total_loss"].detach())
    ordered = sorted(block_s)
    elapsed = ordered[len(ordered) // 2] if len(ordered) % 2 else 0.5 * (ordered[len(ordered) // 2 - 1] + ordered[len(ordered) // 2])

    def pct(q):
        return ordered[min(len(ordered) - 1, max(0, int(round(q * (len(ordered) - 1)))))]

    # ---- roofline of the dominant kernel: hipEvent brackets inside libunet_hip, 2 extra steps
    roof = None
    default_cfg = (args.model == "anomaly_unet" and args.precision == "bf16" and args.batch == 32 and args.size == 256 and not args.height and not args.width
                   and not args.ssim and args.mask == "bernoulli")
    if not args.no_roofline:
        # (every rank runs the two extra steps -- they contain the gradient collectives; only rank 0 brackets them)
        # The two decoder branches normally run on two HIP streams; for these two steps they run on ONE stream so
        # that an event bracket times its kernel alone (concurrent kernels would inflate each other's brackets).
        if hasattr(core, "two_streams"):
            core.two_streams = False
        if rank == 0:
            ops.prof_enable(True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        if rank == 0:
            prof = ops.prof_collect()
            kern = ops.prof_kernels()
            ops.prof_enable(False)
            mfma = {k: v for k, v in kern.items() if v["flops"] > 0 and v["ms"] > 0}
            if mfma:
                name = max(mfma, key=lambda k: mfma[k]["ms"])
                d = mfma[name]
                achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
                peak = PEAK[args.precision]
                traffic, traffic_note, traffic_src = None, "no PMC summary for this kernel in " + os.path.basename(PMC_FILE), None
                try:
                    pmc_all = json.load(open(PMC_FILE))
                    pmc = pmc_all["kernels"]
                    traffic_src = pmc_all.get("csrc_sha16")
                    key = name.split(" ")[0]
                    # a bracket name covers the lock-step (pdma) and ping-pong (pp) instantiations of one kernel body;
                    # "(+ reduce)" brackets also cover the split-K reduction kernels that follow the main kernel
                    fam = [k for k in pmc if k.split("<")[0] in (key, key.replace("_pdma", "_pp"))]
                    main = list(fam)
                    if "(+ reduce)" in name:
                        fam += [k for k in pmc if k.split("<")[0].startswith(key.split("_")[0] + "_reduce")]
                    if main and default_cfg and traffic_src == csrc_sha16():
                        nl = sum(pmc[k]["launches"] for k in main)
                        traffic = int(sum(pmc[k]["traffic_bytes_per_launch"] * pmc[k]["launches"] for k in fam) / nl)
                        traffic_note = ("bytes per launch, 2*FETCH_SIZE + WRITE_SIZE (KiB units, gfx950 x2 on FETCH) from the "
                                        "rocprofv3 --pmc passes of this command on this build (" + os.path.basename(PMC_FILE) +
                                        ", tools/pmc_traffic.sh; launch-weighted over " + ", ".join(sorted(fam)) + ")")
                    elif main:
                        traffic_note = ("PMC summary " + os.path.basename(PMC_FILE) + " was taken from another build or "
                                        "configuration (csrc sha " + str(traffic_src) + " vs " + csrc_sha16() + "): not quoted")
                except (OSError, ValueError, KeyError):
                    pass
                roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                        "traffic_note": traffic_note,
                        "traffic_csrc_sha16": traffic_src,
                        # input + weights + output of exactly the launches this bracket covered (stated by the launcher)
                        "algorithmic_bytes_per_launch": int(d["bytes"] / d["launches"]) if d.get("bytes") else None,
                        "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // 2,
                        "measured": "hipEvent brackets on the launch stream, 2 single-stream steps after the timed region",
                        "per_kernel_ms_per_step": {k: round(v["ms"] / 2, 3) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])},
                        "per_kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) for k, v in mfma.items()},
                        # algorithmic MB per launch of every bracket whose launcher states them (the figure PMC traffic is held against)
                        "per_kernel_algorithmic_mb_per_launch": {k: round(v["bytes"] / v["launches"] / 1e6, 1)
                                                                 for k, v in kern.items() if v.get("bytes")},
                        "per_class_ms_per_step": {k: round(v["ms"] / 2, 3) for k, v in prof.items() if v["launches"]},
                        "per_class_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)
                                             for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0}}
                # BASELINE.json's target reads "MFMA utilisation on 3x3 DoubleConv fwd+bwd": the three conv3x3 classes'
                # FLOPs over THEIR bracketed kernel time (statistics epilogues and split-K reductions included, the
                # BatchNorm-apply / pooling passes not), next to the whole-step figure below
                c3 = [v for k, v in prof.items() if k.startswith("conv3x3_") and v["flops"] > 0 and v["ms"] > 0]
                if c3:
                    roof["conv3x3_fwd_bwd_mfma_frac_of_peak_kernel_time"] = round(
                        sum(v["flops"] for v in c3) / (sum(v["ms"] for v in c3) * 1e-3) / 1e12 / peak, 4)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()

    dist_seen = (None, 1)
    if world > 1:
        import torch.distributed as dist
        dist_seen = (dist.get_backend(), dist.get_world_size())
    from tiaozhanbei_unet_amd import _lib as _L
    cu_budget = int(_L.lib().unet_get_cu_budget())
    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / elapsed
        scale = (fh * fw) / 65536.0
        sz = f"{fh}x{fw}"
        train_tflops = 3 * FWD_GFLOP_PER_IMG_256[args.model] * scale * value / 1e3
        if args.model == "anomaly_unet":
            metric = f"training images/sec, AnomalyUNet {sz} bs={args.batch} per GPU"
            loss_name = "SSIM+focal loss (--use_ssim)" if args.ssim else "MSE+focal loss"
            workload = (f"AnomalyUNet 3x{sz}, bs={args.batch}/GPU, {loss_name}, Adam (BASELINE.json configs[2], "
                        f"synthetic randn images, {args.mask} masks)")
        else:
            metric = f"training images/sec, UNet(3,1) {sz} bs={args.batch} per GPU (seg-only)"
            workload = (f"UNet(3,1) 3x{sz}, bs={args.batch}/GPU, focal loss on sigmoid(logits), Adam (BASELINE.json "
                        f"configs[1], synthetic randn images, {args.mask} masks)")
        out = {
            "metric": metric, "value": round(value, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       "accumulate": "fp32", "master_params": "fp32",
                       # what torch.distributed actually saw (a SCALE record shows RCCL had N ranks) and the CU split
                       "dist_backend": dist_seen[0], "dist_world": dist_seen[1],
                       "reserved_cus": reserved, "cu_budget": cu_budget},
            "blocks": len(block_s),
            "ms_per_step_blocks": [round(1e3 * b / args.steps, 3) for b in block_s],
            "ms_per_step_p10": round(1e3 * pct(0.1) / args.steps, 3),
            "ms_per_step_p90": round(1e3 * pct(0.9) / args.steps, 3),
            "model_tflops": round(train_tflops, 1),
            "mfma_frac_of_peak_whole_step": round(train_tflops / PEAK[args.precision] / world, 4),
            "final_loss": round(loss_val, 5),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(s, args.cpu_batch, args.model)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
