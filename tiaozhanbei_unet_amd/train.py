#!/usr/bin/env python3
"""Training CLI with the reference's contract (/root/reference/src/train.py): same flag names and defaults
(:38-97), output tree `{save_dir}/{category}_{model}_{ts}/{checkpoints,results,visualizations,logs}` (:125-128),
`args.json`, checkpoint dicts and `training_results.json` keys (:280-287) -- running on the HIP path.

    python -m tiaozhanbei_unet_amd.train --category bottle --epochs 2 [--precision bf16] [--synthetic]
    python -m torch.distributed.run --nproc-per-node 8 -m tiaozhanbei_unet_amd.train ...   (data parallel)

Build-only additions: --precision {fp32,bf16}, --synthetic (generate an MVTec-layout toy dataset).
`--model unet` trains the seg-only path (focal on sigmoid(logits)); in the reference that combination crashes
(train_epoch unpacks two outputs, src/train_utils.py:122).  `--use_ssim` selects the SSIM reconstruction head
(a dead flag in the reference, src/train.py:191-194).
"""
import argparse
import json
import os
import tempfile
import time
from datetime import datetime

import torch

FLAGS = [  # name, kwargs  -- reference src/train.py:38-97
    ("--data_root", dict(type=str, default="../datasets/mvtec_anomaly_detection")),
    ("--category", dict(type=str, default="bottle")),
    ("--image_size", dict(type=int, default=256)),
    ("--model", dict(type=str, default="anomaly_unet", choices=["unet", "anomaly_unet"])),
    ("--bilinear", dict(action="store_true")),
    ("--epochs", dict(type=int, default=100)),
    ("--batch_size", dict(type=int, default=16)),
    ("--learning_rate", dict(type=float, default=1e-3)),
    ("--weight_decay", dict(type=float, default=1e-4)),
    ("--optimizer", dict(type=str, default="adam", choices=["adam", "adamw", "sgd"])),
    ("--scheduler", dict(type=str, default="cosine", choices=["cosine", "step", "plateau", "none"])),
    ("--recon_weight", dict(type=float, default=1.0)),
    ("--seg_weight", dict(type=float, default=1.0)),
    ("--use_ssim", dict(action="store_true")),
    ("--num_workers", dict(type=int, default=4)),
    ("--device", dict(type=str, default="auto")),
    ("--seed", dict(type=int, default=42)),
    ("--save_dir", dict(type=str, default="../outputs")),
    ("--save_freq", dict(type=int, default=10)),
    ("--resume", dict(type=str, default=None)),
    ("--val_freq", dict(type=int, default=5)),
    ("--debug", dict(action="store_true")),
    ("--debug_samples", dict(type=int, default=20)),
    # build-only
    ("--precision", dict(type=str, default="fp32", choices=["fp32", "bf16"])),
    ("--synthetic", dict(action="store_true")),
]


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Train UNet for MVTec anomaly detection (MI355X HIP path)")
    for name, kw in FLAGS:
        ap.add_argument(name, **kw)
    return ap.parse_args(argv)


class _SegOnly(torch.nn.Module):
    """UNet under train_epoch's two-output contract: (dummy reconstruction = input, sigmoid(logits))."""

    def __init__(self, unet):
        super().__init__()
        self.unet = unet

    def forward(self, x):
        return x, self.unet(x, sigmoid=True)          # sigmoid inside the head kernel (ops.Head / fused head)


def main(argv=None):
    from . import AnomalyUNet, CombinedLoss, SSIMLoss, UNet, get_optimizer, get_scheduler, train_epoch, validate_epoch
    from .dataset import get_available_categories, get_dataloaders, write_synthetic_mvtec
    from .ddp import DataParallel
    from .utils import create_output_dirs, load_checkpoint, plot_training_curves, print_metrics, save_checkpoint

    args = parse_args(argv)
    torch.manual_seed(args.seed)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if args.device == "cpu" or not torch.cuda.is_available():
        raise SystemExit("this build computes only on an AMD GPU (libunet_hip.so); there is no CPU path")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        from .ddp import configure_overlap
        configure_overlap()                 # RCCL channel cap + CU budget of the persistent kernels, before the communicator exists
        torch.distributed.init_process_group("nccl", device_id=device)
    say = print if rank == 0 else (lambda *a, **k: None)
    say(f"Using device: {device}\nTraining category: {args.category}")

    if args.synthetic:
        args.data_root = write_synthetic_mvtec(tempfile.mkdtemp(prefix="mvtec_syn_"), args.category,
                                               size=max(args.image_size, 32))
    if args.category not in get_available_categories(args.data_root):
        say(f"Category '{args.category}' not found!\nAvailable categories: {get_available_categories(args.data_root)}")
        return

    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    if world > 1:                                   # one run directory: rank 0's time stamp
        box = [stamp]
        torch.distributed.broadcast_object_list(box, src=0)
        stamp = box[0]
    exp_dir = os.path.join(args.save_dir, f"{args.category}_{args.model}_{stamp}")
    dirs = create_output_dirs(exp_dir)
    if rank == 0:
        with open(os.path.join(exp_dir, "args.json"), "w") as f:
            json.dump(vars(args), f, indent=2)

    # data parallel: every rank trains on its own shard (dataset.ShardSampler: same permutation on every rank, strided
    # shards padded to equal length), validation runs on rank 0 over the whole test split
    train_loader, val_loader = get_dataloaders(args.data_root, args.category, args.batch_size, args.image_size,
                                               args.num_workers, rank=rank, world=world, seed=args.seed,
                                               device_preprocess=True)      # workers decode, the GPU transforms
    if args.debug:
        import random
        from torch.utils.data import DataLoader, Subset
        from .dataset import ShardSampler
        picker = random.Random(args.seed)           # the same subset on every rank
        def limit(loader, train):
            idx = picker.sample(range(len(loader.dataset)), min(args.debug_samples, len(loader.dataset)))
            sub = Subset(loader.dataset, idx)
            kw = dict(batch_size=args.batch_size, num_workers=args.num_workers, pin_memory=True,
                      collate_fn=loader.collate_fn)
            if train and world > 1:
                return DataLoader(sub, sampler=ShardSampler(len(sub), rank, world, True, args.seed), **kw)
            return DataLoader(sub, shuffle=train, **kw)
        train_loader, val_loader = limit(train_loader, True), limit(val_loader, False)
    say(f"Train samples: {len(train_loader.dataset)}\nValidation samples: {len(val_loader.dataset)}")

    if args.model == "anomaly_unet":
        core = AnomalyUNet(n_channels=3, bilinear=args.bilinear, precision=args.precision)
        model = core
    else:
        core = UNet(n_channels=3, n_classes=1, bilinear=args.bilinear, precision=args.precision)
        model = _SegOnly(core)
    model = model.to(device)
    total_params = sum(p.numel() for p in core.parameters())
    say(f"Total parameters: {total_params:,}")

    criterion = CombinedLoss(args.recon_weight if args.model == "anomaly_unet" else 0.0, args.seg_weight,
                             recon_criterion=SSIMLoss() if args.use_ssim else None)
    optimizer = get_optimizer(core, args.optimizer, args.learning_rate, args.weight_decay)
    scheduler = get_scheduler(optimizer, args.scheduler, args.epochs)
    start_epoch = 0
    if args.resume:
        start_epoch = load_checkpoint(core, optimizer, args.resume, device)[0] + 1
    net = DataParallel(model) if world > 1 else model
    hook = net.finish_gradients if world > 1 else None

    train_losses, val_losses, best = [], [], float("inf")
    for epoch in range(start_epoch, args.epochs):
        t0 = time.time()
        if hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        tm = train_epoch(net, train_loader, criterion, optimizer, device, epoch, step_hook=hook)
        train_losses.append(tm["total_loss"])
        if scheduler and args.scheduler != "plateau":
            scheduler.step()
        validating = epoch % args.val_freq == 0 or epoch == args.epochs - 1
        vm = validate_epoch(model, val_loader, criterion, device) if (validating and rank == 0) else None
        if validating and scheduler and args.scheduler == "plateau":
            # every replica must take the same learning-rate decision: rank 0's validation loss goes to all
            vl = torch.tensor([vm["total_loss"] if vm is not None else 0.0], dtype=torch.float64, device=device)
            if world > 1:
                torch.distributed.broadcast(vl, src=0)
            scheduler.step(float(vl))
        if vm is not None:
            val_losses.append(vm["total_loss"])
            say(f"\nEpoch {epoch}/{args.epochs - 1}\nTrain Loss: {tm['total_loss']:.4f} (Recon: {tm['recon_loss']:.4f}, "
                f"Seg: {tm['seg_loss']:.4f})\nVal Loss: {vm['total_loss']:.4f} (Recon: {vm['recon_loss']:.4f}, "
                f"Seg: {vm['seg_loss']:.4f})")
            print_metrics(vm["image_metrics"], "Image-level")
            if vm["pixel_metrics"]:
                print_metrics(vm["pixel_metrics"], "Pixel-level")
            if vm["total_loss"] < best:
                best = vm["total_loss"]
                save_checkpoint(core, optimizer, epoch, best, os.path.join(dirs["checkpoints"], "best_model.pth"))
        if rank == 0 and (epoch % args.save_freq == 0 or epoch == args.epochs - 1):
            save_checkpoint(core, optimizer, epoch, tm["total_loss"],
                            os.path.join(dirs["checkpoints"], f"checkpoint_epoch_{epoch}.pth"))
        say(f"Epoch time: {time.time() - t0:.2f}s")

    if rank == 0:
        plot_training_curves(train_losses, val_losses, os.path.join(dirs["results"], "training_curves.png"))
        with open(os.path.join(dirs["results"], "training_results.json"), "w") as f:
            json.dump({"train_losses": train_losses, "val_losses": val_losses, "best_val_loss": best,
                       "total_epochs": args.epochs, "total_params": total_params, "args": vars(args)}, f, indent=2)
        say(f"\nTraining completed!\nBest validation loss: {best:.4f}\nResults saved to: {exp_dir}")
    if world > 1:
        torch.distributed.destroy_process_group()
    return exp_dir


if __name__ == "__main__":
    main()
