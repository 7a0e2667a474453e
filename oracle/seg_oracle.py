"""CPU oracle for the multi-class segmentation head ("next" row f-1) -- TEST INFRASTRUCTURE.

Restates, with explicit arithmetic (fp64 inside, no call into torch's own cross_entropy), what
/root/reference/src/metrics.py computes for the Gear/Kolektor trainers:

* ``CombinedSegmentationLoss.forward``   metrics.py:300-335  (CE :312-320, Dice :323-326 via ``dice_loss`` :233-261,
                                                               focal :329-331 via ``focal_loss`` :264-282)
* ``SegmentationMetrics.update``         metrics.py:22-45    (argmax :31, sklearn confusion matrix :43)
* ``compute_iou/dice/...``               metrics.py:47-108

Pinned against outputs of the reference itself: tests/golden/seg_*.npz (tools/make_goldens_seg.py).
Only tests may import this module.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch


def log_softmax(logits: torch.Tensor) -> torch.Tensor:
    z = logits.double()
    m = z.max(dim=1, keepdim=True).values
    return z - m - (z - m).exp().sum(dim=1, keepdim=True).log()


def _nll(logits, target, ignore_index):
    """per-pixel -log p[target], 0 where ignored; plus the validity mask"""
    lp = log_softmax(logits)
    valid = torch.ones_like(target, dtype=torch.bool) if ignore_index is None else target != ignore_index
    t = torch.where(valid, target, torch.zeros_like(target))
    nll = -lp.gather(1, t.unsqueeze(1)).squeeze(1)
    return torch.where(valid, nll, torch.zeros_like(nll)), valid, t


def cross_entropy(logits, target, class_weights: Optional[Sequence[float]] = None, ignore_index: Optional[int] = None):
    """F.cross_entropy(pred, target, weight, ignore_index) with reduction='mean' (metrics.py:312-320):
    sum_i w[t_i] * nll_i / sum_i w[t_i] over the pixels that are not ignored."""
    nll, valid, t = _nll(logits, target, ignore_index)
    w = torch.ones(logits.shape[1], dtype=torch.float64) if class_weights is None else \
        torch.as_tensor(class_weights, dtype=torch.float64)
    wt = torch.where(valid, w[t], torch.zeros_like(nll))
    return (wt * nll).sum() / wt.sum()


def dice_loss(logits, target, smooth: float = 1e-8):
    """dice_loss(softmax(pred), target) (metrics.py:233-261, called at :324-325): per (image, class) Dice of the
    soft prediction against the one-hot target, loss = 1 - mean over images and classes."""
    p = log_softmax(logits).exp()
    n, c = p.shape[:2]
    onehot = torch.zeros_like(p).scatter_(1, target.unsqueeze(1), 1.0)
    pf, tf = p.reshape(n, c, -1), onehot.reshape(n, c, -1)
    inter = (pf * tf).sum(2)
    union = pf.sum(2) + tf.sum(2)
    return 1.0 - ((2 * inter + smooth) / (union + smooth)).mean()


def focal_loss(logits, target, alpha: float = 1.0, gamma: float = 2.0, ignore_index: Optional[int] = None):
    """focal_loss (metrics.py:264-282): CE per pixel (0 at ignored pixels), pt = exp(-ce), mean over ALL pixels."""
    ce, _, _ = _nll(logits, target, ignore_index)
    pt = (-ce).exp()
    return (alpha * (1 - pt) ** gamma * ce).mean()


def combined_segmentation_loss(logits, target, ce_weight=1.0, dice_weight=1.0, focal_weight=0.0,
                               ignore_index=None, class_weights=None):
    """CombinedSegmentationLoss.forward (metrics.py:300-335)."""
    loss = torch.zeros((), dtype=torch.float64)
    if ce_weight > 0:
        loss = loss + ce_weight * cross_entropy(logits, target, class_weights, ignore_index)
    if dice_weight > 0:
        loss = loss + dice_weight * dice_loss(logits, target)
    if focal_weight > 0:
        loss = loss + focal_weight * focal_loss(logits, target, ignore_index=ignore_index)
    return loss


def argmax_first(logits: torch.Tensor) -> torch.Tensor:
    """torch.argmax(pred, dim=1) (metrics.py:31): the FIRST maximum wins ties."""
    c = logits.shape[1]
    best = logits[:, 0]
    idx = torch.zeros_like(logits[:, 0], dtype=torch.int64)
    for k in range(1, c):
        gt = logits[:, k] > best
        best = torch.where(gt, logits[:, k], best)
        idx = torch.where(gt, torch.full_like(idx, k), idx)
    return idx


def confusion_matrix(pred_labels, target, num_classes: int, ignore_index: Optional[int] = None) -> np.ndarray:
    """SegmentationMetrics.update (metrics.py:33-45): rows = ground truth, columns = prediction, labels
    0..num_classes-1 (anything else is dropped, as sklearn does with ``labels=range(num_classes)``)."""
    p = np.asarray(pred_labels).reshape(-1).astype(np.int64)
    t = np.asarray(target).reshape(-1).astype(np.int64)
    keep = np.ones_like(t, dtype=bool) if ignore_index is None else t != ignore_index
    keep &= (t >= 0) & (t < num_classes) & (p >= 0) & (p < num_classes)
    cm = np.zeros((num_classes, num_classes), dtype=np.int64)
    np.add.at(cm, (t[keep], p[keep]), 1)
    return cm


def metrics_from_confusion(cm: np.ndarray) -> dict:
    """compute_all_metrics (metrics.py:110-140) on a confusion matrix."""
    cm = cm.astype(np.float64)
    tp = np.diag(cm)
    rows, cols = cm.sum(1), cm.sum(0)
    iou = tp / np.maximum(rows + cols - tp, 1e-8)
    dice = 2 * tp / np.maximum(rows + cols, 1e-8)
    precision = tp / np.maximum(cols, 1e-8)
    recall = tp / np.maximum(rows, 1e-8)
    f1 = 2 * precision * recall / np.maximum(precision + recall, 1e-8)
    return {"iou_per_class": iou, "mean_iou": np.nanmean(iou), "dice_per_class": dice, "mean_dice": np.nanmean(dice),
            "pixel_accuracy": tp.sum() / max(cm.sum(), 1e-8), "mean_accuracy": np.nanmean(tp / np.maximum(rows, 1e-8)),
            "precision_per_class": precision, "recall_per_class": recall, "f1_per_class": f1,
            "mean_precision": np.nanmean(precision), "mean_recall": np.nanmean(recall), "mean_f1": np.nanmean(f1)}
