"""GPU parity of the multi-class segmentation head: tiaozhanbei_unet_amd.metrics (HIP kernels
unet_seg_loss / unet_seg_confusion through the C-ABI) against outputs of the reference's src/metrics.py
(tests/golden/seg_*.npz) and against the CPU oracle."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import seg_oracle as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

_spec = importlib.util.spec_from_file_location("seg_cases", os.path.join(os.path.dirname(__file__), "..", "tools", "seg_cases.py"))
seg_cases = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(seg_cases)


def golden(name):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in load_golden(name).items()}


@pytest.mark.parametrize("name", sorted(seg_cases.CASES))
def test_combined_segmentation_loss_and_metrics_match_reference(name):
    from tiaozhanbei_unet_amd import metrics as M
    n, c, h, w, kw, ign = seg_cases.CASES[name]
    g = golden(name)
    logits, target = seg_cases.inputs(name, n, c, h, w, ign, kw.get("ignore_index"))
    x = logits.to(DEV).requires_grad_(True)
    t = target.to(DEV)
    crit = M.CombinedSegmentationLoss(**kw)
    loss = crit(x, t)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(1.0, abs(float(g["loss"]))), (float(loss), float(g["loss"]))
    gerr = np.abs(x.grad.cpu().numpy() - g["dlogits"]).max()
    assert gerr <= 2e-5 * max(1e-3, np.abs(g["dlogits"]).max()) + 1e-9, gerr
    m = M.SegmentationMetrics(c, ignore_index=kw.get("ignore_index"))
    assert np.array_equal(m.argmax(logits.to(DEV)).cpu().numpy(), g["argmax"]), "argmax indices must be bit-exact"
    m.update(logits.to(DEV), t)
    assert np.array_equal(m.confusion_matrix, g["confusion"])
    allm = m.compute_all_metrics()
    for k in ("mean_iou", "mean_dice", "pixel_accuracy", "mean_f1"):
        assert abs(allm[k] - float(g[k])) < 1e-12, k
    # a second update accumulates; label-map predictions take the other branch of update()
    m.update(torch.from_numpy(g["argmax"]).to(DEV), t)
    assert np.array_equal(m.confusion_matrix, 2 * g["confusion"])


def test_standalone_dice_and_focal_against_the_oracle():
    from tiaozhanbei_unet_amd import metrics as M
    n, c, h, w, _, _ = seg_cases.CASES["seg_c4_default"]
    logits, target = seg_cases.inputs("seg_c4_default", n, c, h, w, 0.0, None)
    prob = torch.softmax(logits, 1)
    pd = prob.to(DEV).requires_grad_(True)
    d = M.dice_loss(pd, target.to(DEV))
    d.backward()
    pr = prob.double().requires_grad_(True)
    onehot = torch.zeros_like(pr).scatter_(1, target.unsqueeze(1), 1.0)
    inter = (pr * onehot).flatten(2).sum(2)
    union = pr.flatten(2).sum(2) + onehot.flatten(2).sum(2)
    ref = 1 - ((2 * inter + 1e-8) / (union + 1e-8)).mean()       # reference metrics.py:233-261 on probabilities
    ref.backward()
    assert abs(float(d) - float(ref)) < 1e-5
    assert float((pd.grad.cpu().double() - pr.grad).abs().max()) < 1e-8
    xf = logits.to(DEV).requires_grad_(True)
    f = M.focal_loss(xf, target.to(DEV), alpha=0.75, gamma=2, ignore_index=255)
    f.backward()
    xo = logits.double().requires_grad_(True)
    fo = S.focal_loss(xo, target, alpha=0.75, gamma=2.0, ignore_index=255)
    fo.backward()
    assert abs(float(f) - float(fo)) < 1e-5 * max(1.0, float(fo))
    assert float((xf.grad.cpu().double() - xo.grad).abs().max()) < 1e-7
    with pytest.raises(TypeError):
        M.focal_loss(xf, target.to(DEV))                          # the reference's own failure mode (ignore_index=None)
    with pytest.raises(TypeError):
        M.CombinedSegmentationLoss(focal_weight=1.0)(xf, target.to(DEV))
    with pytest.raises(RuntimeError):
        M.CombinedSegmentationLoss()(logits, target)              # CPU tensors: no fallback


def test_unet_multiclass_training_step_with_segmentation_loss():
    """The 4-class UNet of the Kolektor trainer (n_classes=4, train_kolektorsdd.py) trains through the HIP backbone
    with the HIP loss: finite decreasing loss, confusion counts add up; a KolektorSDD-shaped frame (non-square)."""
    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd import metrics as M
    torch.manual_seed(0)
    net = P.UNet(3, 4, precision="bf16").to(DEV).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    crit = M.CombinedSegmentationLoss(ce_weight=1.0, dice_weight=1.0, focal_weight=0.5, ignore_index=255,
                                      class_weights=[0.5, 1.0, 2.0, 1.0])
    x = torch.randn(2, 3, 176, 64, device=DEV)
    t = torch.randint(0, 4, (2, 176, 64), device=DEV)
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = crit(net(x), t)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    m = M.SegmentationMetrics(4)
    with torch.no_grad():
        out = net(x)
    m.update(out, t)
    assert int(m.confusion_matrix.sum()) == t.numel() and m.total_samples == t.numel()
    assert torch.equal(m.argmax(out), torch.argmax(out, dim=1))


def test_segmentation_unet_is_unet_plus_bottleneck_dropout():
    """SegmentationUNet (reference src/model.py:111-153): identical to UNet in eval mode and with dropout=0; in training
    the bottleneck is multiplied by the [N, C, 1, 1] bernoulli(1-p)/(1-p) noise torch's Dropout2d draws."""
    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd import model as Mdl
    torch.manual_seed(3)
    seg = P.SegmentationUNet(3, 4, dropout=0.5, precision="fp32").to(DEV)
    ref = P.UNet(3, 4, precision="fp32").to(DEV)
    assert list(seg.state_dict().keys()) == list(ref.state_dict().keys())
    ref.load_state_dict(seg.state_dict())
    x = torch.randn(2, 3, 32, 32, device=DEV)
    seg.eval(); ref.eval()
    with torch.no_grad():
        assert torch.equal(seg(x), ref(x))
    seg.train(); ref.train()
    torch.manual_seed(11)
    y = seg(x)
    torch.manual_seed(11)
    noise = torch.empty(2, 1024, 1, 1, device=DEV).bernoulli_(0.5).div_(0.5)
    assert 0.3 < float((noise == 0).float().mean()) < 0.7
    ref.load_state_dict(seg.state_dict())                      # (running statistics moved in seg's forward: restore both)
    seg2 = P.SegmentationUNet(3, 4, dropout=0.5, precision="fp32").to(DEV).train()
    seg2.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    # expected: the same blocks with x5 scaled by the noise in torch arithmetic
    Mdl._pack_cache(seg2)
    x1, x2, x3, x4, x5 = Mdl._encoder(seg2, x)
    x5s = (x5.float() * noise).to(x5.dtype).contiguous(memory_format=torch.channels_last)
    exp = seg2.outc(seg2.up4(seg2.up3(seg2.up2(seg2.up1(x5s, x4), x3), x2), x1))
    # y was produced with running stats one step behind exp's model, but training-mode BN uses batch statistics: equal
    assert float((y - exp).abs().max()) <= 1e-5 * max(1.0, float(exp.abs().max()))
    y.square().mean().backward()
    g = [p.grad for p in seg.parameters()]
    assert all(t is not None and bool(torch.isfinite(t).all()) for t in g)
    seg0 = P.SegmentationUNet(3, 4, dropout=0.0, precision="fp32").to(DEV).train()
    assert isinstance(seg0.dropout, torch.nn.Identity)


def test_anomaly_score_kernel_matches_the_reference_formula():
    from tiaozhanbei_unet_amd import ops, utils
    torch.manual_seed(5)
    recon, img = torch.rand(3, 3, 40, 24, device=DEV), torch.rand(3, 3, 40, 24, device=DEV)
    for l1 in (False, True):
        score, per_image = ops.anomaly_score(recon, img, l1=l1)
        d = recon - img
        ref = d.abs().mean(1) if l1 else (d * d).mean(1)         # src/utils.py:205-215
        assert float((score - ref).abs().max()) < 1e-6
        assert float((per_image - ref.flatten(1).mean(1)).abs().max()) < 1e-6
    assert float((utils.compute_anomaly_score(recon, img) - ((recon - img) ** 2).mean(1)).abs().max()) < 1e-6
    assert torch.allclose(utils.compute_anomaly_score(recon.cpu(), img.cpu(), "l1"), (recon - img).abs().mean(1).cpu())
