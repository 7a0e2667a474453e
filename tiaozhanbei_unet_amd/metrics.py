"""Multi-class segmentation metrics and losses on the HIP path (mirror of /root/reference/src/metrics.py).

Same names, signatures and results as the reference module used by the Gear / Kolektor trainers:
``SegmentationMetrics`` (metrics.py:9-204), ``compute_metrics_from_predictions`` (:207-230), ``dice_loss``
(:233-261), ``focal_loss`` (:264-282), ``CombinedSegmentationLoss`` (:285-335).  The per-pixel work -- softmax, the
three loss terms and their gradient, argmax and the confusion-matrix counts -- runs in libunet_hip.so
(``unet_seg_loss`` / ``unet_seg_confusion``); the reference copies every prediction to the host and calls sklearn
per batch (metrics.py:33-43).  Only the C x C matrix and the derived ratios live on the host.

The plotting helper ``plot_confusion_matrix`` (matplotlib / seaborn GUI code, metrics.py:178-204) is out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .ops import _ptr, _require_cuda, _stream, _workspace

_NO_IGNORE = -1


class _SegLoss(torch.autograd.Function):
    """value + gradient in one call (unet_seg_loss); backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, pred, target, class_weights, ignore_index, is_prob, ce_w, dice_w, focal_w, alpha, gamma):
        _require_cuda(pred, target)
        if pred.dim() < 3 or target.dim() != pred.dim() - 1:
            raise ValueError(f"expected (N, C, ...) predictions and (N, ...) labels, got {tuple(pred.shape)} / {tuple(target.shape)}")
        n, c = pred.shape[0], pred.shape[1]
        hw = int(np.prod(pred.shape[2:]))
        x = pred.detach().contiguous().float()
        t = target.detach().contiguous().long()
        dev = x.device
        out = torch.empty(4, dtype=torch.float32, device=dev)
        need_grad = pred.requires_grad
        dl = torch.empty_like(x) if need_grad else None
        lib = L.lib()
        need = lib.unet_seg_loss_workspace(n, c, hw)
        ws = _workspace(need, dev)
        cw = None if class_weights is None else class_weights.to(device=dev, dtype=torch.float32).contiguous()
        L.check(lib.unet_seg_loss(_ptr(x), _ptr(t), n, c, hw, _ptr(cw), _NO_IGNORE if ignore_index is None else int(ignore_index),
                                  1 if is_prob else 0, ce_w, dice_w, focal_w, alpha, gamma, _ptr(out), _ptr(dl),
                                  _ptr(ws), ws.numel(), _stream()), "unet_seg_loss")
        ctx.dl = dl
        ctx.in_dtype = pred.dtype
        return out[0].clone(), out[1:].clone()

    @staticmethod
    def backward(ctx, g, _g_terms):
        dl = ctx.dl
        if dl is None:
            return (None,) * 10
        return ((dl * g).to(ctx.in_dtype),) + (None,) * 9


def dice_loss(pred, target, smooth=1e-8):
    """Dice loss of a probability map (N, C, H, W) against labels (N, H, W) (reference metrics.py:233-261)."""
    if smooth != 1e-8:
        raise NotImplementedError("the HIP kernel implements the reference's default smooth=1e-8")
    total, _ = _SegLoss.apply(pred, target, None, None, True, 0.0, 1.0, 0.0, 1.0, 2.0)
    return total


def focal_loss(pred, target, alpha=1, gamma=2, ignore_index=None):
    """Focal loss on logits (reference metrics.py:264-282).  As in the reference, ``ignore_index`` must be an int:
    it is handed to ``F.cross_entropy`` there, which rejects ``None``."""
    if ignore_index is None:
        raise TypeError("cross_entropy_loss(): argument 'ignore_index' (position 5) must be int, not NoneType")
    total, _ = _SegLoss.apply(pred, target, None, ignore_index, False, 0.0, 0.0, 1.0, float(alpha), float(gamma))
    return total


class CombinedSegmentationLoss(torch.nn.Module):
    """Combined loss for segmentation: CE (class weights, ignore_index) + Dice + focal (reference metrics.py:285-335)."""

    def __init__(self, ce_weight=1.0, dice_weight=1.0, focal_weight=0.0, ignore_index=None, class_weights=None):
        super().__init__()
        self.ce_weight = ce_weight
        self.dice_weight = dice_weight
        self.focal_weight = focal_weight
        self.ignore_index = ignore_index
        self.class_weights = class_weights
        if class_weights is not None:
            self.class_weights = torch.tensor(class_weights, dtype=torch.float32)

    def forward(self, pred, target):
        if self.focal_weight > 0 and self.ignore_index is None:      # the reference's own TypeError (metrics.py:278)
            raise TypeError("cross_entropy_loss(): argument 'ignore_index' (position 5) must be int, not NoneType")
        if not (self.ce_weight > 0 or self.dice_weight > 0 or self.focal_weight > 0):
            return 0
        cw = self.class_weights if self.ce_weight > 0 else None
        total, _ = _SegLoss.apply(pred, target, cw, self.ignore_index, False,
                                  float(self.ce_weight) if self.ce_weight > 0 else 0.0,
                                  float(self.dice_weight) if self.dice_weight > 0 else 0.0,
                                  float(self.focal_weight) if self.focal_weight > 0 else 0.0, 1.0, 2.0)
        return total


class SegmentationMetrics:
    """Comprehensive metrics for semantic segmentation tasks (reference metrics.py:9-175); the confusion matrix is
    counted on the GPU (first-maximum argmax + integer atomics) and kept there between updates."""

    def __init__(self, num_classes, ignore_index=None):
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.reset()

    def reset(self):
        self._dev_cm = None
        self._host_cm = np.zeros((self.num_classes, self.num_classes), dtype=np.int64)
        self.total_samples = 0

    @property
    def confusion_matrix(self):
        if self._dev_cm is not None:
            return self._host_cm + self._dev_cm.cpu().numpy()
        return self._host_cm

    @confusion_matrix.setter
    def confusion_matrix(self, value):
        """The reference's attribute is a plain numpy array callers may assign or ``+=`` (metrics.py:19,44): an
        assignment replaces the counts held so far (device-side counts included)."""
        value = np.asarray(value, dtype=np.int64)
        if value.shape != (self.num_classes, self.num_classes):
            raise ValueError(f"confusion matrix must be {self.num_classes}x{self.num_classes}, got {value.shape}")
        self._host_cm = value.copy()
        self._dev_cm = None

    def update(self, pred, target):
        """pred: (N, C, H, W) scores or (N, H, W) labels; target: (N, H, W) labels."""
        _require_cuda(pred, target)
        t = target.contiguous().long()
        ign = _NO_IGNORE if self.ignore_index is None else int(self.ignore_index)
        if pred.dim() == 4:
            n, c = pred.shape[0], pred.shape[1]
            hw = int(np.prod(pred.shape[2:]))
            if c != self.num_classes:
                raise ValueError(f"{c} score maps for {self.num_classes} classes")
            if self._dev_cm is None or self._dev_cm.device != pred.device:
                if self._dev_cm is not None:
                    self._host_cm = self._host_cm + self._dev_cm.cpu().numpy()
                self._dev_cm = torch.zeros((c, c), dtype=torch.int64, device=pred.device)
            x = pred.detach().contiguous().float()
            L.check(L.lib().unet_seg_confusion(_ptr(x), _ptr(t), n, c, hw, ign, None, _ptr(self._dev_cm), _stream()),
                    "unet_seg_confusion")
        else:           # label maps: a C x C histogram, integer arithmetic on the device
            p = pred.contiguous().long().reshape(-1)
            tt = t.reshape(-1)
            keep = (tt >= 0) & (tt < self.num_classes) & (p >= 0) & (p < self.num_classes)
            if self.ignore_index is not None:
                keep &= tt != self.ignore_index
            idx = (tt[keep] * self.num_classes + p[keep])
            cm = torch.bincount(idx, minlength=self.num_classes ** 2).reshape(self.num_classes, self.num_classes)
            self._host_cm = self._host_cm + cm.cpu().numpy()
        valid = t != self.ignore_index if self.ignore_index is not None else torch.ones_like(t, dtype=torch.bool)
        self.total_samples += int(valid.sum())

    def argmax(self, pred):
        """Label map of (N, C, H, W) scores, the first maximum winning ties like torch.argmax (metrics.py:31)."""
        _require_cuda(pred)
        n, c = pred.shape[0], pred.shape[1]
        hw = int(np.prod(pred.shape[2:]))
        x = pred.detach().contiguous().float()
        out = torch.empty((n,) + tuple(pred.shape[2:]), dtype=torch.int64, device=pred.device)
        L.check(L.lib().unet_seg_confusion(_ptr(x), None, n, c, hw, _NO_IGNORE, _ptr(out), None, _stream()),
                "unet_seg_confusion")
        return out

    # ---- ratios of the C x C matrix (host arithmetic; reference metrics.py:47-140 computes the same quantities)
    def _ratios(self):
        """Every per-class ratio from one pass over the matrix: rows = ground truth, columns = prediction."""
        cm = self.confusion_matrix.astype(np.float64)
        hit = np.diagonal(cm)
        truth_total, pred_total = cm.sum(axis=1), cm.sum(axis=0)
        floor = 1e-8                                   # the reference's guard against empty classes

        def ratio(num, den):
            return num / np.maximum(den, floor)

        precision, recall = ratio(hit, pred_total), ratio(hit, truth_total)
        return {"iou": ratio(hit, truth_total + pred_total - hit), "dice": ratio(2 * hit, truth_total + pred_total),
                "precision": precision, "recall": recall, "f1": ratio(2 * precision * recall, precision + recall),
                "class_acc": recall, "pixel_acc": hit.sum() / max(cm.sum(), floor)}

    def compute_iou(self, per_class=True):
        v = self._ratios()["iou"]
        return v if per_class else np.nanmean(v)

    def compute_dice(self, per_class=True):
        v = self._ratios()["dice"]
        return v if per_class else np.nanmean(v)

    def compute_pixel_accuracy(self):
        return self._ratios()["pixel_acc"]

    def compute_mean_accuracy(self):
        return np.nanmean(self._ratios()["class_acc"])

    def compute_precision_recall_f1(self, per_class=True):
        r = self._ratios()
        trio = (r["precision"], r["recall"], r["f1"])
        return trio if per_class else tuple(np.nanmean(t) for t in trio)

    def compute_all_metrics(self):
        r = self._ratios()
        out = {"confusion_matrix": self.confusion_matrix, "pixel_accuracy": r["pixel_acc"],
               "mean_accuracy": np.nanmean(r["class_acc"])}
        for key, name in (("iou", "iou"), ("dice", "dice"), ("precision", "precision"), ("recall", "recall"), ("f1", "f1")):
            out[f"{name}_per_class"] = r[key]
            out[f"mean_{name}"] = np.nanmean(r[key])
        return out

    def print_metrics(self, class_names=None):
        m = self.compute_all_metrics()
        names = class_names or [f"Class {i}" for i in range(self.num_classes)]
        print("=" * 60)
        print("SEGMENTATION METRICS")
        print("=" * 60)
        print(f"Pixel Accuracy: {m['pixel_accuracy']:.4f}")
        print(f"Mean Accuracy:  {m['mean_accuracy']:.4f}")
        print(f"Mean IoU:       {m['mean_iou']:.4f}")
        print(f"Mean Dice:      {m['mean_dice']:.4f}")
        print(f"Mean F1:        {m['mean_f1']:.4f}")
        for i, name in enumerate(names):
            print(f"{name:<15} IoU {m['iou_per_class'][i]:.4f}  Dice {m['dice_per_class'][i]:.4f}  "
                  f"P {m['precision_per_class'][i]:.4f}  R {m['recall_per_class'][i]:.4f}  F1 {m['f1_per_class'][i]:.4f}")


def compute_metrics_from_predictions(predictions, targets, num_classes, class_names=None):
    """Metrics of a batch of predictions (reference metrics.py:207-230)."""
    calc = SegmentationMetrics(num_classes)
    calc.update(predictions, targets)
    return calc.compute_all_metrics()
