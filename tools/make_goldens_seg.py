#!/usr/bin/env python3
"""Generate tests/golden/seg_*.npz by running the REFERENCE's src/metrics.py (build container only).

Same rules as tools/make_goldens.py: the reference is imported read-only (no bytecode; `seaborn`, absent from the
image and used only for plotting, is stubbed), inputs come from the key-seeded generator of oracle/weights.py, only
inputs-by-seed and the reference's OUTPUTS are stored.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_goldens_seg.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))

import metrics as ref_metrics        # noqa: E402  (the reference)

sys.path.insert(0, os.path.join(ROOT, "tools"))

OUT = os.path.join(ROOT, "tests", "golden")

from seg_cases import CASES, inputs  # noqa: E402


def main():
    for name, (n, c, h, w, kw, ign) in CASES.items():
        logits, target = inputs(name, n, c, h, w, ign, kw.get("ignore_index"))
        x = logits.clone().requires_grad_(True)
        crit = ref_metrics.CombinedSegmentationLoss(**kw)
        loss = crit(x, target)
        loss.backward()
        m = ref_metrics.SegmentationMetrics(c, ignore_index=kw.get("ignore_index"))
        m.update(logits, target)
        allm = m.compute_all_metrics()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), loss=loss.detach().numpy(), dlogits=x.grad.numpy(),
                            argmax=torch.argmax(logits, dim=1).numpy().astype(np.int64),
                            confusion=m.confusion_matrix.astype(np.int64), mean_iou=np.float64(allm["mean_iou"]),
                            mean_dice=np.float64(allm["mean_dice"]), pixel_accuracy=np.float64(allm["pixel_accuracy"]),
                            mean_f1=np.float64(allm["mean_f1"]))
        print(f"  wrote {name}.npz  loss={float(loss):.6f}")


if __name__ == "__main__":
    main()
