"""Host-side bookkeeping with the reference's contracts (/root/reference/src/utils.py):
checkpoint dict keys (:37-58), metric-dict keys (:61-94), anomaly score (:205-215), output
tree (:272-282), ``AverageMeter`` (:285-300).  No device arithmetic of the hot path lives here.
"""
from __future__ import annotations

import os

import numpy as np
import torch


class AverageMeter:
    """Running (weighted) average."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def create_output_dirs(base_dir):
    out = {}
    for name in ("checkpoints", "results", "visualizations", "logs"):
        out[name] = os.path.join(base_dir, name)
        os.makedirs(out[name], exist_ok=True)
    return out


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model


def save_checkpoint(model, optimizer, epoch, loss, filepath):
    """Same dict as the reference (un-prefixed keys even under the data-parallel wrapper)."""
    torch.save({"epoch": epoch, "model_state_dict": _unwrap(model).state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "loss": loss}, filepath)
    print(f"Checkpoint saved to {filepath}")


def load_checkpoint(model, optimizer, filepath, device):
    ckpt = torch.load(filepath, map_location=device, weights_only=True)
    _unwrap(model).load_state_dict(ckpt["model_state_dict"])
    if optimizer:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    print(f"Checkpoint loaded from {filepath}, epoch {ckpt['epoch']}, loss {ckpt['loss']:.4f}")
    return ckpt["epoch"], ckpt["loss"]


def compute_anomaly_score(reconstruction, original, method="mse"):
    """Per-pixel channel-mean error map (reference src/utils.py:205-215; 'ssim' is the reference's MSE placeholder).
    GPU tensors run in libunet_hip.so (unet_anomaly_score); host tensors (plots, reports) stay host arithmetic."""
    if method not in ("mse", "ssim", "l1"):
        raise ValueError(f"Unknown method: {method}")
    if reconstruction.is_cuda and original.is_cuda and reconstruction.dim() == 4:
        from . import ops
        return ops.anomaly_score(reconstruction, original, l1=(method == "l1"))[0]
    diff = reconstruction - original
    return diff.abs().mean(dim=1) if method == "l1" else (diff * diff).mean(dim=1)


def metrics_from_counts(tp, fp, fn, tn):
    """calculate_metrics' threshold metrics from confusion counts (the device epilogue, ops.threshold_confusion)."""
    m = {"accuracy": (tp + tn) / max(tp + tn + fp + fn, 1),
         "precision": tp / (tp + fp) if tp + fp else 0,
         "recall": tp / (tp + fn) if tp + fn else 0,
         "specificity": tn / (tn + fp) if tn + fp else 0}
    pr = m["precision"] + m["recall"]
    m["f1_score"] = 2 * m["precision"] * m["recall"] / pr if pr > 0 else 0
    return m


def calculate_metrics(y_true, y_pred, y_scores=None):
    y_true = np.asarray(y_true, dtype=int).ravel()
    y_pred = np.asarray(y_pred, dtype=int).ravel()
    tp = int(np.sum((y_true == 1) & (y_pred == 1)))
    tn = int(np.sum((y_true == 0) & (y_pred == 0)))
    fp = int(np.sum((y_true == 0) & (y_pred == 1)))
    fn = int(np.sum((y_true == 1) & (y_pred == 0)))
    m = {"accuracy": (tp + tn) / max(tp + tn + fp + fn, 1),
         "precision": tp / (tp + fp) if tp + fp else 0,
         "recall": tp / (tp + fn) if tp + fn else 0,
         "specificity": tn / (tn + fp) if tn + fp else 0}
    pr = m["precision"] + m["recall"]
    m["f1_score"] = 2 * m["precision"] * m["recall"] / pr if pr > 0 else 0
    if y_scores is not None:
        try:
            from sklearn.metrics import auc, precision_recall_curve, roc_auc_score
            m["auroc"] = roc_auc_score(y_true, y_scores)
            precision, recall, _ = precision_recall_curve(y_true, y_scores)
            m["auprc"] = auc(recall, precision)
        except ValueError:
            m["auroc"] = m["auprc"] = 0.0
    return m


def print_metrics(metrics, prefix=""):
    print(f"\n{prefix} Metrics:\n" + "-" * 40)
    for k, v in metrics.items():
        print(f"{k.capitalize()}: {v:.4f}" if isinstance(v, float) else f"{k.capitalize()}: {v}")
    print("-" * 40)


def get_optimal_threshold(y_true, y_scores):
    from sklearn.metrics import precision_recall_curve
    precision, recall, thresholds = precision_recall_curve(y_true, y_scores)
    f1 = 2 * precision * recall / (precision + recall + 1e-8)
    i = int(np.argmax(f1))
    return (thresholds[i] if i < len(thresholds) else 0.5), f1[i]


def plot_training_curves(train_losses, val_losses, save_path=None):
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:                                          # plotting is optional
        return
    fig, ax = plt.subplots(figsize=(8, 4))
    ax.plot(train_losses, label="train")
    if val_losses:
        ax.plot(np.linspace(0, max(len(train_losses) - 1, 0), len(val_losses)), val_losses, label="val")
    ax.set_xlabel("epoch"); ax.set_ylabel("loss"); ax.legend(); ax.grid(True)
    if save_path:
        fig.savefig(save_path, dpi=120, bbox_inches="tight")
    plt.close(fig)
