"""Loss heads and the train / validate loops of the reference, on the HIP path.

Same call surface as /root/reference/src/train_utils.py: ``CombinedLoss`` (:10-44),
``SSIMLoss`` (:47-104), ``train_epoch`` (:107-152), ``validate_epoch`` (:155-260),
``get_optimizer`` (:263-272), ``get_scheduler`` (:275-284).  The loss arithmetic runs in
libunet_hip.so (ops.MseFocal / ops.Ssim); the loops are host plumbing with the reference's
return-dict keys.  Two deliberate differences, both value-preserving: the three per-step
``.item()`` host syncs (:137-139) become device-side running sums read once per epoch, and
``--use_ssim`` (dead in the reference, src/train.py:191-194) is honoured when a caller opts in
through ``CombinedLoss(recon_criterion=SSIMLoss())``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .utils import AverageMeter, calculate_metrics, compute_anomaly_score, metrics_from_counts


class SSIMLoss(nn.Module):
    """1 - mean SSIM with an 11x11 Gaussian window (sigma 1.5), zero padded."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size = window_size
        self.size_average = size_average

    def forward(self, img1, img2):
        """0-d loss, or one loss per image when ``size_average=False`` (reference :84-87)."""
        return ops.Ssim.apply(img1, img2, self.window_size, bool(self.size_average))


class CombinedLoss(nn.Module):
    """recon_weight * MSE(recon, image) + seg_weight * focal(anomaly_map, mask)."""

    def __init__(self, recon_weight=1.0, seg_weight=1.0, focal_alpha=0.25, focal_gamma=2.0,
                 recon_criterion=None):
        super().__init__()
        self.recon_weight = recon_weight
        self.seg_weight = seg_weight
        self.focal_alpha = focal_alpha
        self.focal_gamma = focal_gamma
        self.recon_criterion = recon_criterion      # None = MSE (what the reference always uses)

    def focal_loss(self, pred, target):
        zero = pred.new_zeros((1, 1, 1, 1))
        return ops.MseFocal.apply(zero, pred, zero, target, self.focal_alpha, self.focal_gamma)[1]

    def forward(self, reconstruction, anomaly_map, original_image, true_mask):
        if self.recon_criterion is not None:
            # (the reference's dead --use_ssim switch, src/train.py:191-194, made live: SSIM head + focal)
            recon_loss = self.recon_criterion(reconstruction, original_image)
            seg_loss = self.focal_loss(anomaly_map, true_mask)
        else:
            recon_loss, seg_loss = ops.MseFocal.apply(reconstruction, anomaly_map, original_image, true_mask,
                                                      self.focal_alpha, self.focal_gamma)
        total = self.recon_weight * recon_loss + self.seg_weight * seg_loss
        return {"total_loss": total, "recon_loss": recon_loss, "seg_loss": seg_loss}


PIXEL_THRESHOLDS = (0.3, 0.5, 0.7)          # src/train_utils.py:234


def _batches(loader, device):
    """(batch dict, images, masks) on the device.  Loaders built with ``device_preprocess`` ship the decoded uint8
    images / masks: the whole image transform of src/dataset.py:134-151 (resize, flip, rotation, colour jitter, ToTensor,
    Normalize) then runs on the GPU (augment.DeviceTransform of the loader's dataset) and the batch regains the
    reference's ``image`` / ``mask`` keys."""
    ds, tf = getattr(loader, "dataset", None), None
    while ds is not None and tf is None:            # (a torch Subset wraps the dataset that owns the transform)
        tf, ds = getattr(ds, "device_transform", None), getattr(ds, "dataset", None)
    for batch in loader:
        if "image_raw" in batch:
            if tf is None:
                raise RuntimeError("raw batches need the dataset's device_transform (dataset.MVTecDataset(device_preprocess=True))")
            images = tf(batch.pop("image_raw"), device=device)
            masks = tf.masks(batch.pop("mask_raw"), device=device)
            batch["image"], batch["mask"] = images, masks
            yield batch, images, masks
            continue
        yield batch, batch["image"].to(device, non_blocking=True), batch["mask"].to(device, non_blocking=True)


def train_epoch(model, train_loader, criterion, optimizer, device, epoch, step_hook=None):
    """One epoch; returns batch-size-weighted means under the reference's keys."""
    model.train()
    sums = torch.zeros(3, dtype=torch.float64, device=device)
    count = 0
    for _, images, masks in _batches(train_loader, device):
        reconstruction, anomaly_map = model(images)
        losses = criterion(reconstruction, anomaly_map, images, masks)
        optimizer.zero_grad(set_to_none=True)
        losses["total_loss"].backward()
        if step_hook is not None:
            step_hook()                     # data-parallel gradient exchange completes here
        optimizer.step()
        bs = images.size(0)
        with torch.no_grad():
            sums += bs * torch.stack([losses["total_loss"].detach(), losses["recon_loss"].detach(),
                                      losses["seg_loss"].detach()]).double()
        count += bs
    t, r, s = (sums / max(count, 1)).tolist()
    return {"total_loss": t, "recon_loss": r, "seg_loss": s}


def validate_epoch(model, val_loader, criterion, device):
    """Eval-mode pass over ``val_loader`` with the reference's return dict (src/train_utils.py:155-260): batch-size
    weighted losses, ``image_metrics``, ``pixel_metrics`` and ``predictions`` = {labels [N], scores [N, H, W] per-pixel
    error maps (src/utils.py:205-215), masks_true, masks_pred [N, 1, H, W]}.

    One branch cannot mirror the reference because the reference cannot run it: with both classes present it hands the
    N labels and the N*H*W thresholded map pixels to sklearn (:206-210), which raises.  There the image-level score is
    the mean of an image's error map, thresholded at the 95th percentile of those N scores (``predictions`` gains the
    key ``image_scores``).  The all-one-class branch (:217-227) -- what MVTec's all-normal split exercises -- is the
    reference's, value for value (tests/golden/validate_epoch_all_normal.npz)."""
    model.eval()
    meters = {k: AverageMeter() for k in ("total_loss", "recon_loss", "seg_loss")}
    labels, scores, masks_true, masks_pred = [], [], [], []
    pix_counts = None
    with torch.no_grad():
        for batch, images, masks in _batches(val_loader, device):
            reconstruction, anomaly_map = model(images)
            losses = criterion(reconstruction, anomaly_map, images, masks)
            bs = images.size(0)
            vals = torch.stack([losses[k].detach().float() for k in meters]).tolist()     # one host sync per batch
            for (k, m), v in zip(meters.items(), vals):
                m.update(v, bs)
            labels.extend(np.asarray(batch["label"]))
            scores.extend(compute_anomaly_score(reconstruction, images).cpu().numpy())
            masks_true.extend(masks.cpu().numpy())
            masks_pred.extend(anomaly_map.cpu().numpy())
            # pixel metrics of the anomalous images (:232-245): confusion counts at the three thresholds on the device
            pix_counts = ops.threshold_confusion(anomaly_map, masks, PIXEL_THRESHOLDS,
                                                 select=torch.as_tensor(np.asarray(batch["label"]) == 1), counts=pix_counts)
    labels, scores = np.array(labels), np.array(scores)
    masks_true, masks_pred = np.array(masks_true), np.array(masks_pred)
    predictions = {"labels": labels, "scores": scores, "masks_true": masks_true, "masks_pred": masks_pred}

    if len(np.unique(labels)) > 1:
        image_scores = scores.reshape(len(scores), -1).mean(1)
        threshold = np.percentile(image_scores, 95)
        image_metrics = calculate_metrics(labels, (image_scores > threshold).astype(int), image_scores)
        predictions["image_scores"] = image_scores
    else:
        normal = 1.0 if (len(labels) and labels[0] == 0) else 0.0
        image_metrics = {"accuracy": normal, "precision": 0.0, "recall": 0.0, "specificity": normal,
                         "f1_score": 0.0, "auroc": 0.0, "auprc": 0.0}

    pixel_metrics = {}
    if (labels == 1).sum() > 0 and pix_counts is not None:
        for thr, (tp, fp, fn, tn) in zip(PIXEL_THRESHOLDS, pix_counts.cpu().tolist()):
            if tp + fn > 0 and fp + tn > 0:                       # both classes among the true pixels (:241)
                pixel_metrics[f"pixel_f1_@{thr}"] = metrics_from_counts(tp, fp, fn, tn)["f1_score"]

    return {"total_loss": meters["total_loss"].avg, "recon_loss": meters["recon_loss"].avg,
            "seg_loss": meters["seg_loss"].avg, "image_metrics": image_metrics, "pixel_metrics": pixel_metrics,
            "predictions": predictions}


def get_optimizer(model, optimizer_name="adam", learning_rate=1e-3, weight_decay=1e-4):
    name = optimizer_name.lower()
    params = list(model.parameters())
    # same update rule and state_dict layout as the reference's torch.optim.Adam(...) (src/train_utils.py:266); on the
    # GPU: optim.FusedAdam = ONE libunet_hip launch over all parameter tensors per step
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if name in ("adam", "adamw") and on_gpu:
        from .optim import FusedAdam
        return FusedAdam(params, lr=learning_rate, weight_decay=weight_decay, decoupled=(name == "adamw"))
    if name == "adam":
        return torch.optim.Adam(params, lr=learning_rate, weight_decay=weight_decay)
    if name == "adamw":
        return torch.optim.AdamW(params, lr=learning_rate, weight_decay=weight_decay)
    if name == "sgd":
        return torch.optim.SGD(params, lr=learning_rate, momentum=0.9, weight_decay=weight_decay)
    raise ValueError(f"Unknown optimizer: {optimizer_name}")


def get_scheduler(optimizer, scheduler_name="cosine", num_epochs=100, eta_min=1e-6):
    name = scheduler_name.lower()
    sched = torch.optim.lr_scheduler
    if name == "cosine":
        return sched.CosineAnnealingLR(optimizer, T_max=num_epochs, eta_min=eta_min)
    if name == "step":
        return sched.StepLR(optimizer, step_size=num_epochs // 3, gamma=0.1)
    if name == "plateau":
        return sched.ReduceLROnPlateau(optimizer, mode="min", patience=10, factor=0.5)
    return None
