#!/usr/bin/env python3
"""Evaluation CLI with the reference's contract (/root/reference/src/test.py: flags :26-61, test_model :66-133,
`test_metrics.json` / `detailed_results.json` :187-232) on the HIP path.  The forward runs on the GPU; thresholding
and metrics are host numpy, as in the reference."""
import argparse
import json
import os

import numpy as np
import torch

FLAGS = [  # reference src/test.py:26-61
    ("--data_root", dict(type=str, default="../datasets/mvtec_anomaly_detection")),
    ("--category", dict(type=str, default="bottle")),
    ("--image_size", dict(type=int, default=256)),
    ("--model", dict(type=str, default="anomaly_unet", choices=["unet", "anomaly_unet"])),
    ("--bilinear", dict(action="store_true")),
    ("--checkpoint", dict(type=str, required=True)),
    ("--batch_size", dict(type=int, default=16)),
    ("--num_workers", dict(type=int, default=4)),
    ("--device", dict(type=str, default="auto")),
    ("--threshold", dict(type=float, default=None)),
    ("--pixel_thresholds", dict(type=float, nargs="+", default=[0.3, 0.5, 0.7])),
    ("--output_dir", dict(type=str, default="../test_results")),
    ("--save_visualizations", dict(action="store_true")),
    ("--max_vis_samples", dict(type=int, default=20)),
    ("--precision", dict(type=str, default="fp32", choices=["fp32", "bf16"])),   # build-only
]


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Test UNet for MVTec anomaly detection (MI355X HIP path)")
    for name, kw in FLAGS:
        ap.add_argument(name, **kw)
    return ap.parse_args(argv)


def test_model(model, test_loader, device, threshold=None, pixel_thresholds=None):
    from . import AnomalyUNet, ops
    from .utils import compute_anomaly_score, get_optimal_threshold
    model.eval()
    out = {k: [] for k in ("images", "reconstructions", "anomaly_maps", "masks_true", "labels", "anomaly_types",
                           "image_paths", "anomaly_scores")}
    pix = None
    with torch.no_grad():
        from .train_utils import _batches
        for batch, images, _ in _batches(test_loader, device):     # (raw uint8 batches are transformed on the device)
            if isinstance(model, AnomalyUNet):
                recon, amap = model(images)
            else:
                amap, recon = model(images, sigmoid=True), images      # sigmoid inside the head kernel
            if pixel_thresholds:          # pixel-level confusion counts of the anomalous images, on the device (:86-101)
                pix = ops.threshold_confusion(amap, batch["mask"], pixel_thresholds,
                                              select=torch.as_tensor(np.asarray(batch["label"]) == 1), counts=pix)
            out["anomaly_scores"].extend(compute_anomaly_score(recon, images).cpu().numpy())
            out["images"].extend(images.cpu()); out["reconstructions"].extend(recon.cpu())
            out["anomaly_maps"].extend(amap.cpu().numpy()); out["masks_true"].extend(batch["mask"].cpu().numpy())
            out["labels"].extend(np.asarray(batch["label"])); out["anomaly_types"].extend(batch["anomaly_type"])
            out["image_paths"].extend(batch["image_path"])
    for k in ("labels", "anomaly_scores", "masks_true", "anomaly_maps"):
        out[k] = np.array(out[k])
    if threshold is None:
        # the reference feeds per-pixel score maps here; image-level score = their mean
        img_scores = out["anomaly_scores"].reshape(len(out["labels"]), -1).mean(1)
        threshold = float(get_optimal_threshold(out["labels"], img_scores)[0]) if len(np.unique(out["labels"])) > 1 else 0.5
        print(f"Optimal threshold: {threshold:.4f}")
    out["image_scores"] = out["anomaly_scores"].reshape(len(out["labels"]), -1).mean(1)
    out["predictions"] = (out["image_scores"] > threshold).astype(int)
    out["threshold"] = threshold
    if pix is not None:
        out["pixel_counts"] = {float(t): c for t, c in zip(pixel_thresholds, pix.cpu().tolist())}
    return out


def evaluate_results(results, pixel_thresholds):
    from .utils import calculate_metrics
    ev = {"image_metrics": calculate_metrics(results["labels"], results["predictions"], results["image_scores"]),
          "pixel_metrics": {}, "type_metrics": {}}
    bad = results["labels"] == 1
    if bad.sum() > 0 and "pixel_counts" in results:         # counted on the device by test_model
        from .utils import metrics_from_counts
        for t in pixel_thresholds:
            tp, fp, fn, tn = results["pixel_counts"][float(t)]
            if tp + fn > 0 and fp + tn > 0:
                ev["pixel_metrics"][f"threshold_{t}"] = metrics_from_counts(tp, fp, fn, tn)
    elif bad.sum() > 0:
        truth = (results["masks_true"][bad] > 0.5).astype(np.uint8).ravel()
        if len(np.unique(truth)) > 1:
            for t in pixel_thresholds:
                pred = (results["anomaly_maps"][bad] > t).astype(np.uint8).ravel()
                ev["pixel_metrics"][f"threshold_{t}"] = calculate_metrics(truth, pred)
    for kind in sorted(set(results["anomaly_types"])):
        sel = np.array([k == kind for k in results["anomaly_types"]])
        ev["type_metrics"][kind] = {"count": int(sel.sum()),
                                    "detected": int(results["predictions"][sel].sum())}
    return ev


def main(argv=None):
    from . import AnomalyUNet, UNet
    from .dataset import get_available_categories, get_dataloaders
    from .utils import load_checkpoint, print_metrics
    args = parse_args(argv)
    if args.device == "cpu" or not torch.cuda.is_available():
        raise SystemExit("this build computes only on an AMD GPU (libunet_hip.so); there is no CPU path")
    device = torch.device("cuda")
    if args.category not in get_available_categories(args.data_root):
        print(f"Category '{args.category}' not found!")
        return
    out_dir = os.path.join(args.output_dir, f"{args.category}_test_results")
    os.makedirs(out_dir, exist_ok=True)
    _, loader = get_dataloaders(args.data_root, args.category, args.batch_size, args.image_size, args.num_workers,
                                device_preprocess=True)                     # workers decode, the GPU transforms
    model = (AnomalyUNet(3, args.bilinear, precision=args.precision) if args.model == "anomaly_unet"
             else UNet(3, 1, args.bilinear, precision=args.precision)).to(device)
    load_checkpoint(model, None, args.checkpoint, device)
    results = test_model(model, loader, device, args.threshold, pixel_thresholds=args.pixel_thresholds)
    ev = evaluate_results(results, args.pixel_thresholds)
    print_metrics(ev["image_metrics"], "Image-level")

    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        if isinstance(o, np.generic):
            return o.item()
        return o.tolist() if isinstance(o, np.ndarray) else o

    with open(os.path.join(out_dir, "test_metrics.json"), "w") as f:
        json.dump({**plain(ev), "threshold": float(results["threshold"]), "args": vars(args)}, f, indent=2)
    with open(os.path.join(out_dir, "detailed_results.json"), "w") as f:
        json.dump({"labels": results["labels"].tolist(), "predictions": results["predictions"].tolist(),
                   "anomaly_scores": results["image_scores"].tolist(), "anomaly_types": list(results["anomaly_types"]),
                   "image_paths": list(results["image_paths"]), "threshold": float(results["threshold"])}, f, indent=2)
    print(f"\nTesting completed!\nResults saved to: {out_dir}")
    return out_dir


if __name__ == "__main__":
    main()
