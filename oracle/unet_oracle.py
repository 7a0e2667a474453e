"""CPU oracle: functional restatement of the reference hot path (TEST INFRASTRUCTURE).

Pure functions over a ``state`` mapping (the reference's ``state_dict`` key
layout) -- no nn.Module, no HIP, plain fp32/fp64 torch on the CPU.  Each function
cites the reference lines it follows.  Pinned against outputs of the reference
itself: see ``oracle/__init__.py`` and ``tests/test_oracle_golden.py``.

Only tests, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

State = Mapping[str, torch.Tensor]

BN_EPS = 1e-5        # nn.BatchNorm2d default, /root/reference/src/model.py:15,18
BN_MOMENTUM = 0.1

# ---- optional emulation of the HIP path's bf16 STORAGE (arithmetic stays fp32, like the MFMA accumulators) ----------
# Inside ``with bf16_storage():`` every tensor the bf16 mode of the HIP path keeps in bf16 is rounded to bf16 at the
# point where that path stores it: convolution / transposed-convolution operands (weights, the input image) and
# outputs, activations after ReLU, up-sampled tensors -- and, in the backward pass, the gradients of those same
# tensors.  The reference has no bf16 mode; this is the yardstick that separates the rounding noise of the MODE (large
# on the deep layers' gradients: ~0.3 L2-relative at N = 2) from an error of the KERNELS (tests/test_gpu_round2.py).
_EMULATE_BF16 = False


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _q(x):
    """a stored activation / gradient pair"""
    return _RoundBF16.apply(x) if _EMULATE_BF16 else x


def _qw(w):
    """an operand that is only READ in bf16 (packed weights, the image): its gradient stays fp32"""
    return w + (w.to(torch.bfloat16).to(w.dtype) - w).detach() if _EMULATE_BF16 else w


class bf16_storage:
    def __enter__(self):
        global _EMULATE_BF16
        self.prev, _EMULATE_BF16 = _EMULATE_BF16, True
        return self

    def __exit__(self, *exc):
        global _EMULATE_BF16
        _EMULATE_BF16 = self.prev
        return False


# --------------------------------------------------------------------------- BN
def batch_norm(state: State, prefix: str, x: torch.Tensor, training: bool,
               new_stats: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """BatchNorm2d as torch runs it for model.py:15,18.

    train: normalise with the BIASED batch variance; running stats move with
    momentum 0.1 towards the batch mean / UNBIASED variance;
    ``num_batches_tracked`` += 1.  eval: running stats.
    Updated statistics are returned through ``new_stats`` (the oracle is pure).
    """
    g, b = state[f"{prefix}.weight"], state[f"{prefix}.bias"]
    if training:
        m = x.shape[0] * x.shape[2] * x.shape[3]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if new_stats is not None:
            with torch.no_grad():
                unbiased = var * (m / max(m - 1, 1))
                new_stats[f"{prefix}.running_mean"] = (
                    (1 - BN_MOMENTUM) * state[f"{prefix}.running_mean"] + BN_MOMENTUM * mean.detach())
                new_stats[f"{prefix}.running_var"] = (
                    (1 - BN_MOMENTUM) * state[f"{prefix}.running_var"] + BN_MOMENTUM * unbiased.detach())
                new_stats[f"{prefix}.num_batches_tracked"] = state[f"{prefix}.num_batches_tracked"] + 1
    else:
        mean, var = state[f"{prefix}.running_mean"], state[f"{prefix}.running_var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]


# ------------------------------------------------------------------ the blocks
def double_conv(state: State, prefix: str, x, training, new_stats=None):
    """(conv3x3 pad1 no-bias -> BN -> ReLU) x 2 -- model.py:13-20."""
    for conv_idx in (0, 3):
        x = _q(F.conv2d(x, _qw(state[f"{prefix}.double_conv.{conv_idx}.weight"]), None, 1, 1))
        x = batch_norm(state, f"{prefix}.double_conv.{conv_idx + 1}", x, training, new_stats)
        x = _q(torch.clamp_min(x, 0.0))
    return x


def down(state: State, prefix: str, x, training, new_stats=None):
    """MaxPool2d(2) then DoubleConv -- model.py:31-37 (floor, stride 2)."""
    x = F.max_pool2d(x, 2)
    return double_conv(state, f"{prefix}.maxpool_conv.1", x, training, new_stats)


def upsample_bilinear2x(x):
    """nn.Upsample(scale 2, bilinear, align_corners=True) -- model.py:48."""
    n, c, h, w = x.shape
    oh, ow = 2 * h, 2 * w

    def axis(n_in, n_out):
        if n_out == 1 or n_in == 1:
            src = torch.zeros(n_out, dtype=x.dtype)
        else:
            src = torch.arange(n_out, dtype=x.dtype) * ((n_in - 1) / (n_out - 1))
        i0 = src.floor().long().clamp(0, n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        return i0, i1, (src - i0.to(x.dtype))

    y0, y1, fy = axis(h, oh)
    x0, x1, fx = axis(w, ow)
    top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
    bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
    return _q(top * (1 - fy)[None, None, :, None] + bot * fy[None, None, :, None])


def conv_transpose2x2(x, w, b):
    """ConvTranspose2d(k=2,s=2) as one GEMM + pixel shuffle -- model.py:51
    (weight [Cin, Cout, 2, 2])."""
    n, _, h, wd = x.shape
    co = w.shape[1]
    y = torch.einsum("nchw,cokl->nohkwl", x, _qw(w)).reshape(n, co, 2 * h, 2 * wd)
    return _q(y + b[None, :, None, None])


def up(state: State, prefix: str, x1, x2, training, bilinear, new_stats=None):
    """Up.forward -- model.py:54-66: upsample x1, centre-pad to x2's size
    (diff//2 before, remainder after), cat([x2, x1]) skip FIRST, DoubleConv."""
    if bilinear:
        x1 = upsample_bilinear2x(x1)
    else:
        x1 = conv_transpose2x2(x1, state[f"{prefix}.up.weight"], state[f"{prefix}.up.bias"])
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(state, f"{prefix}.conv", torch.cat([x2, x1], dim=1), training, new_stats)


def out_conv(state: State, prefix: str, x):
    """1x1 conv with bias -- model.py:72."""
    return F.conv2d(x, state[f"{prefix}.conv.weight"], state[f"{prefix}.conv.bias"])


# ------------------------------------------------------------------ the models
def _encoder(state, x, training, new_stats):
    x1 = double_conv(state, "inc", _qw(x), training, new_stats)
    x2 = down(state, "down1", x1, training, new_stats)
    x3 = down(state, "down2", x2, training, new_stats)
    x4 = down(state, "down3", x3, training, new_stats)
    x5 = down(state, "down4", x4, training, new_stats)
    return x1, x2, x3, x4, x5


def _decoder(state, feats, suffix, training, bilinear, new_stats):
    x1, x2, x3, x4, x5 = feats
    y = up(state, f"up1{suffix}", x5, x4, training, bilinear, new_stats)
    y = up(state, f"up2{suffix}", y, x3, training, bilinear, new_stats)
    y = up(state, f"up3{suffix}", y, x2, training, bilinear, new_stats)
    return up(state, f"up4{suffix}", y, x1, training, bilinear, new_stats)


def unet_forward(state: State, x, training=True, bilinear=False, new_stats=None):
    """UNet.forward -- model.py:97-108: raw logits [N, n_classes, H, W]."""
    feats = _encoder(state, x, training, new_stats)
    return out_conv(state, "outc", _decoder(state, feats, "", training, bilinear, new_stats))


def segmentation_unet_forward(state: State, x, training=True, bilinear=False, new_stats=None, drop_noise=None):
    """SegmentationUNet.forward -- model.py:133-153: UNet with nn.Dropout2d on the bottleneck x5 (:146).
    ``drop_noise`` [N, C] = the per-(image, channel) keep/scale factors of the dropout (None: identity, i.e. eval
    mode or dropout=0)."""
    x1, x2, x3, x4, x5 = _encoder(state, x, training, new_stats)
    if drop_noise is not None:
        x5 = x5 * drop_noise[:, :, None, None]
    return out_conv(state, "outc", _decoder(state, (x1, x2, x3, x4, x5), "", training, bilinear, new_stats))


def anomaly_unet_forward(state: State, x, training=True, bilinear=False, new_stats=None
                         ) -> Tuple[torch.Tensor, torch.Tensor]:
    """AnomalyUNet.forward -- model.py:188-210: shared encoder, recon decoder
    first then seg decoder, sigmoid on both heads."""
    feats = _encoder(state, x, training, new_stats)
    recon = torch.sigmoid(out_conv(state, "outc_recon",
                                   _decoder(state, feats, "_recon", training, bilinear, new_stats)))
    amap = torch.sigmoid(out_conv(state, "outc_seg",
                                  _decoder(state, feats, "_seg", training, bilinear, new_stats)))
    return recon, amap


# ------------------------------------------------------------------ loss heads
class _BCE(torch.autograd.Function):
    """F.binary_cross_entropy(reduction='none') as ATen computes it (the arithmetic
    the reference reaches through train_utils.py:25): forward clamps each log term at
    -100; backward is g * (p - t) / max((1 - p) * p, 1e-12) -- finite at p = 0 and 1."""

    @staticmethod
    def forward(ctx, p, t):
        ctx.save_for_backward(p, t)
        log_p = torch.clamp(torch.log(p), min=-100.0)
        log_1p = torch.clamp(torch.log(1.0 - p), min=-100.0)
        return -(t * log_p + (1.0 - t) * log_1p)

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        return g * (p - t) / torch.clamp_min((1.0 - p) * p, 1e-12), None


def focal_loss(pred, target, alpha=0.25, gamma=2.0):
    """CombinedLoss.focal_loss -- train_utils.py:23-28.  BCE on probabilities
    (see _BCE); pt = exp(-bce); mean(alpha * (1-pt)^gamma * bce); gradient flows
    through pt."""
    bce = _BCE.apply(pred, target)
    pt = torch.exp(-bce)
    return (alpha * (1.0 - pt) ** gamma * bce).mean()


def combined_loss(recon, amap, image, mask, recon_weight=1.0, seg_weight=1.0,
                  focal_alpha=0.25, focal_gamma=2.0):
    """CombinedLoss.forward -- train_utils.py:30-44."""
    recon_loss = ((recon - image) ** 2).mean()
    seg_loss = focal_loss(amap, mask, focal_alpha, focal_gamma)
    return {"total_loss": recon_weight * recon_loss + seg_weight * seg_loss,
            "recon_loss": recon_loss, "seg_loss": seg_loss}


def gaussian_window(window_size=11, sigma=1.5, dtype=torch.float32):
    """SSIMLoss.gaussian / create_window -- train_utils.py:57-65 (fp32 1-D taps,
    normalised, outer product)."""
    g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2))
                      for i in range(window_size)], dtype=torch.float32)
    g = g / g.sum()
    return torch.outer(g, g).to(dtype)


def ssim_loss(img1, img2, window_size=11, size_average=True):
    """SSIMLoss.forward / _ssim -- train_utils.py:67-104.  Depthwise Gaussian,
    zero padding window_size//2, variances as E[x^2]-mu^2, C1=1e-4, C2=9e-4."""
    c = img1.shape[1]
    win = gaussian_window(window_size, 1.5, img1.dtype)[None, None].expand(c, 1, -1, -1).contiguous()
    p = window_size // 2

    def blur(t):
        return F.conv2d(t, win, padding=p, groups=c)

    mu1, mu2 = blur(img1), blur(img2)
    s11 = blur(img1 * img1) - mu1 * mu1
    s22 = blur(img2 * img2) - mu2 * mu2
    s12 = blur(img1 * img2) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    if size_average:
        return 1 - smap.mean()
    return 1 - smap.mean(1).mean(1).mean(1)


def anomaly_score(recon, image):
    """compute_anomaly_score(method='mse') -- src/utils.py:205-208."""
    return ((recon - image) ** 2).mean(dim=1)


def validate_pass(state: State, batches, recon_weight=1.0, seg_weight=1.0):
    """The arithmetic of validate_epoch's all-one-class branch -- train_utils.py:155-204,217-227: eval-mode forward,
    batch-size weighted mean of the three losses (:184-187), per-pixel error maps (:190), predicted masks."""
    tot = [0.0, 0.0, 0.0]
    count = 0
    scores, masks_pred = [], []
    with torch.no_grad():
        for b in batches:
            recon, amap = anomaly_unet_forward(state, b["image"], training=False)
            d = combined_loss(recon, amap, b["image"], b["mask"], recon_weight, seg_weight)
            n = b["image"].shape[0]
            for i, k in enumerate(("total_loss", "recon_loss", "seg_loss")):
                tot[i] += float(d[k]) * n
            count += n
            scores.append(anomaly_score(recon, b["image"]))
            masks_pred.append(amap)
    return {"total_loss": tot[0] / count, "recon_loss": tot[1] / count, "seg_loss": tot[2] / count,
            "scores": torch.cat(scores), "masks_pred": torch.cat(masks_pred)}


# ------------------------------------------------------- one training step
def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], opt_state: dict,
              lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4):
    """torch.optim.Adam as configured by get_optimizer -- train_utils.py:266:
    L2-coupled weight decay (g += wd * p before the moments), bias-corrected."""
    b1, b2 = betas
    opt_state["step"] = opt_state.get("step", 0) + 1
    t = opt_state["step"]
    out = {}
    for k, p in params.items():
        g = grads[k] + weight_decay * p
        m = opt_state.setdefault(("m", k), torch.zeros_like(p))
        v = opt_state.setdefault(("v", k), torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        out[k] = p - (lr / (1 - b1 ** t)) * (m / denom)
    return out


def is_trainable(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


def train_step(state: Dict[str, torch.Tensor], opt_state: dict, image, mask,
               model="anomaly_unet", bilinear=False, lr=1e-3, weight_decay=1e-4,
               recon_weight=1.0, seg_weight=1.0):
    """One iteration of train_epoch's body -- train_utils.py:117-133 -- for
    AnomalyUNet (MSE + focal) or UNet (the seg-only focal path on sigmoid(logits),
    BASELINE.md section 5).  Returns (new_state, loss dict of floats)."""
    work = {k: (v.detach().clone().requires_grad_(True) if is_trainable(k) else v)
            for k, v in state.items()}
    new_stats: Dict[str, torch.Tensor] = {}
    if model == "anomaly_unet":
        recon, amap = anomaly_unet_forward(work, image, True, bilinear, new_stats)
        losses = combined_loss(recon, amap, image, mask, recon_weight, seg_weight)
    else:
        amap = torch.sigmoid(unet_forward(work, image, True, bilinear, new_stats))
        seg = focal_loss(amap, mask)
        losses = {"total_loss": seg, "recon_loss": torch.zeros(()), "seg_loss": seg}
    keys = [k for k in work if is_trainable(k)]
    grads = torch.autograd.grad(losses["total_loss"], [work[k] for k in keys])
    new_params = adam_step({k: state[k] for k in keys}, dict(zip(keys, grads)), opt_state,
                           lr=lr, weight_decay=weight_decay)
    new_state = dict(state)
    new_state.update(new_params)
    new_state.update(new_stats)
    return new_state, {k: float(v.detach()) for k, v in losses.items()}
