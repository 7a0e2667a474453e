"""MI355X-native U-Net anomaly-segmentation training path (gfx950 HIP kernels behind a C-ABI).

Drop-in for the hot path of ukeSJTU/tiaozhanbei-unet: ``model`` mirrors src/model.py,
``train_utils`` mirrors src/train_utils.py, ``metrics`` mirrors src/metrics.py (multi-class segmentation
loss / argmax / confusion matrix).  Compute happens only in libunet_hip.so.
"""
from . import _lib
from .model import (AnomalyUNet, DoubleConv, Down, OutConv, SegmentationUNet, UNet, Up, set_default_precision,
                    set_precision)
from .train_utils import CombinedLoss, SSIMLoss, get_optimizer, get_scheduler, train_epoch, validate_epoch
from .metrics import CombinedSegmentationLoss, SegmentationMetrics

__all__ = ["AnomalyUNet", "UNet", "SegmentationUNet", "DoubleConv", "Down", "Up", "OutConv", "CombinedLoss", "SSIMLoss",
           "train_epoch", "validate_epoch", "get_optimizer", "get_scheduler", "set_precision",
           "set_default_precision", "CombinedSegmentationLoss", "SegmentationMetrics"]
