"""Case table and seeded inputs of the segmentation-head fixtures (shared by tools/make_goldens_seg.py, which runs
the reference on them, and by the tests, which run the oracle / the HIP path on the same inputs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import weights as W  # noqa: E402

# name -> (N, C, H, W, loss kwargs, fraction of ignored pixels)
CASES = {
    "seg_c4_default": (2, 4, 24, 40, dict(ce_weight=1.0, dice_weight=1.0, focal_weight=0.0), 0.0),
    # the reference's focal term only runs with an integer ignore_index (metrics.py:278 passes it straight to
    # F.cross_entropy, which rejects None); Dice needs targets without ignored pixels (F.one_hot, metrics.py:247)
    "seg_c4_all_terms_weighted": (2, 4, 17, 23, dict(ce_weight=0.7, dice_weight=0.5, focal_weight=2.0, ignore_index=255,
                                                     class_weights=[0.2, 1.0, 3.0, 0.5]), 0.0),
    "seg_c2_focal_only": (3, 2, 16, 16, dict(ce_weight=0.0, dice_weight=0.0, focal_weight=1.0, ignore_index=255), 0.0),
    "seg_c3_ignore": (2, 3, 20, 12, dict(ce_weight=1.0, dice_weight=0.0, focal_weight=0.5, ignore_index=255,
                                        class_weights=[1.0, 2.0, 0.5]), 0.15),
    "seg_c8_dice_only": (1, 8, 32, 32, dict(ce_weight=0.0, dice_weight=1.0, focal_weight=0.0), 0.0),
}


def inputs(name, n, c, h, w, ignore_frac, ignore_index):
    logits = W.make_input(name + ":logits", (n, c, h, w)) * 3.0
    target = (W.make_input(name + ":target", (n, h, w), kind="uniform") * c).long().clamp_(0, c - 1)
    if ignore_frac > 0:
        drop = W.make_input(name + ":drop", (n, h, w), kind="uniform") < ignore_frac
        target = torch.where(drop, torch.full_like(target, ignore_index), target)
    # exact ties for the argmax rule: copy channel 0 into channel 1 on a stripe
    logits[:, 1, :2, :] = logits[:, 0, :2, :]
    return logits, target


