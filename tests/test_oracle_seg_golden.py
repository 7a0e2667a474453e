"""CPU: the segmentation-head oracle (oracle/seg_oracle.py) against outputs of the reference's src/metrics.py
(tests/golden/seg_*.npz, tools/make_goldens_seg.py)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import seg_oracle as S
from oracle import weights as W

_spec = importlib.util.spec_from_file_location(
    "make_goldens_seg_cases", os.path.join(os.path.dirname(__file__), "..", "tools", "seg_cases.py"))
seg_cases = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(seg_cases)


@pytest.mark.parametrize("name", sorted(seg_cases.CASES))
def test_seg_oracle_matches_reference(name):
    n, c, h, w, kw, ign = seg_cases.CASES[name]
    g = {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in load_golden(name).items()}
    logits, target = seg_cases.inputs(name, n, c, h, w, ign, kw.get("ignore_index"))
    x = logits.double().requires_grad_(True)
    loss = S.combined_segmentation_loss(x, target, **kw)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 2e-5 * max(1.0, abs(float(g["loss"])))
    assert np.abs(x.grad.numpy() - g["dlogits"]).max() <= 1e-6
    am = S.argmax_first(logits)
    assert np.array_equal(am.numpy(), g["argmax"]), "argmax indices must be bit-exact (first maximum wins ties)"
    cm = S.confusion_matrix(am.numpy(), target.numpy(), c, kw.get("ignore_index"))
    assert np.array_equal(cm, g["confusion"])
    m = S.metrics_from_confusion(cm)
    for k in ("mean_iou", "mean_dice", "pixel_accuracy", "mean_f1"):
        assert abs(m[k] - float(g[k])) < 1e-12, k
