"""U-Net model zoo of the reference, running on hand-written gfx950 kernels.

Drop-in surface (same class names, constructor / forward signatures, attributes and
``state_dict`` key layout as /root/reference/src/model.py): ``DoubleConv`` (:6-23),
``Down`` (:26-37), ``Up`` (:40-66), ``OutConv`` (:69-75), ``UNet`` (:78-108),
``SegmentationUNet`` (:111-153), ``AnomalyUNet`` (:156-210).

The stock ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` children are kept
ONLY as parameter holders -- same registration order, hence the same default
initialisation under ``torch.manual_seed`` and the same checkpoint keys as the reference --
their ``forward`` is never called.  All arithmetic goes through ``ops`` (libunet_hip.so).

Extra, optional, keyword: ``precision`` in {"fp32", "bf16"} (default: the package default,
fp32 = the parity mode; bf16 = the throughput mode with fp32 accumulation, fp32 BatchNorm
statistics and fp32 master parameters).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops

_DTYPES = {"fp32": torch.float32, "bf16": torch.bfloat16}
_default_precision = "fp32"


def set_default_precision(precision: str) -> None:
    global _default_precision
    if precision not in _DTYPES:
        raise ValueError(f"precision must be one of {sorted(_DTYPES)}")
    _default_precision = precision


def set_precision(module: nn.Module, precision: str) -> nn.Module:
    """Switch every block of ``module`` to ``precision`` (parameters stay fp32)."""
    if precision not in _DTYPES:
        raise ValueError(f"precision must be one of {sorted(_DTYPES)}")
    for m in module.modules():
        if isinstance(m, _HipBlock):
            m.precision = precision
    return module


class _HipBlock(nn.Module):
    precision = None

    @property
    def compute_dtype(self) -> torch.dtype:
        return _DTYPES[self.precision or _default_precision]


def _conv_bn_relu(conv: nn.Conv2d, bn: nn.BatchNorm2d, x, x_up, first: bool = False, in_link=None, out_link=None,
                  head=None, pool=False):
    """One conv3x3 -> BatchNorm2d -> ReLU third of DoubleConv.  Batch vs running statistics follow ``bn.training``
    (the holder module itself, so a frozen ``bn.eval()`` inside a training model is honoured like in the reference's
    nn.Sequential); ``bn.momentum is None`` is torch's cumulative moving average (factor 1 / num_batches_tracked).
    ``head`` = (OutConv module, sigmoid): fuse the following 1x1 head (training path only)."""
    training = bn.training
    track = training or bn.running_mean is None
    if bn.momentum is not None:
        momentum = bn.momentum
    elif training and bn.num_batches_tracked is not None:
        momentum = 1.0 / float(int(bn.num_batches_tracked) + 1)
    else:
        momentum = 0.0
    if first:
        ops._require_cuda(x)
        out = ops.FirstConvBnRelu.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                        track, momentum, out_link)
    else:
        # inference (eval mode, no autograd recording): BatchNorm folded into the layer, one kernel
        fold = (not track) and not torch.is_grad_enabled()
        hw, hb, hs = (head[0].conv.weight, head[0].conv.bias, head[1]) if head is not None else (None, None, False)
        out = ops.ConvBnRelu.apply(x, x_up, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   track, momentum, fold, in_link, out_link, hw, hb, hs, pool)
    if training and bn.num_batches_tracked is not None:
        if _deferred_counters is not None:
            _deferred_counters.append(bn.num_batches_tracked)
        else:
            with torch.no_grad():
                bn.num_batches_tracked.add_(1)
    return out


# Inside a whole-model forward the 46 `num_batches_tracked += 1` updates are issued as ONE multi-tensor launch.
_deferred_counters = None


class _BatchedCounters:
    def __enter__(self):
        global _deferred_counters
        self.outer = _deferred_counters
        _deferred_counters = []
        return self

    def __exit__(self, *exc):
        global _deferred_counters
        mine, _deferred_counters = _deferred_counters, self.outer
        if mine and exc[0] is None:
            with torch.no_grad():
                torch._foreach_add_(mine, 1)
        return False


class DoubleConv(_HipBlock):
    """(conv3x3 => BatchNorm => ReLU) * 2 on MFMA implicit-GEMM kernels."""

    def __init__(self, in_channels, out_channels, mid_channels=None, precision=None):
        super().__init__()
        mid_channels = mid_channels or out_channels
        if mid_channels % 64 or out_channels % 64:
            raise ValueError("the gfx950 kernels need output channel counts that are multiples of 64 "
                             f"(got {mid_channels}, {out_channels})")
        self.precision = precision
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def forward(self, x, x_up=None, head=None, pool=False, out_link=None):
        """``out_link`` (internal, optional): ops.BnLink to the ONE consumer of the result (the transposed convolution of
        the next Up block), which then produces this block's last ReLU mask / BatchNorm-backward sums in its data gradient.
        ``x_up`` (internal, optional): second channel block of the input, i.e. the up-sampled
        tensor of ``Up`` -- concatenated after ``x`` and centre-padded to its size on the fly.
        ``head`` (internal, optional): (OutConv, sigmoid) applied to the result -- in training mode the 1x1 head
        is fused with the last BatchNorm + ReLU (the activation is never written).
        ``pool`` (internal, optional): also return ``max_pool2d(result, 2)`` -- in training mode the pool is fused with
        the last BatchNorm + ReLU (forward) and with its backward (see ops.ConvBnRelu)."""
        seq = self.double_conv
        # the intermediate activation has exactly one consumer (the second convolution): its ReLU mask and
        # BatchNorm-backward sums are produced by that convolution's data-gradient kernel (ops.BnLink)
        link = ops.BnLink() if (seq[1].training and torch.is_grad_enabled()) else None
        fuse_head = head is not None and ops.FUSE_BN_HEAD and seq[4].training
        if x_up is None and ops.first_layer_ok(x, seq[0], self.compute_dtype):
            a = _conv_bn_relu(seq[0], seq[1], x, None, first=True, out_link=link)      # the image layer
        else:
            x = ops.to_operator_layout(x, self.compute_dtype)
            if x_up is not None:
                x_up = ops.to_operator_layout(x_up, self.compute_dtype)
            a = _conv_bn_relu(seq[0], seq[1], x, x_up, out_link=link)
        fuse_pool = pool and ops.FUSE_BN_POOL and seq[4].training and a.shape[2] >= 2 and a.shape[3] >= 2 and \
            bool(ops.L.lib().unet_bn_relu_pool_supported(ops._DT[a.dtype], seq[3].out_channels))
        if fuse_head or fuse_pool or head is not None or pool:
            out_link = None
        out = _conv_bn_relu(seq[3], seq[4], a, None, in_link=link, out_link=out_link, head=head if fuse_head else None,
                            pool=fuse_pool)
        if head is not None and not fuse_head:
            out = head[0](out, sigmoid=head[1])
        if pool and not fuse_pool:
            return out, None
        return out


class Down(_HipBlock):
    """MaxPool2d(2) then DoubleConv."""

    def __init__(self, in_channels, out_channels, precision=None):
        super().__init__()
        self.precision = precision
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels, precision=precision))

    def forward(self, x, pooled=None, pool=False):
        """``pooled`` (internal, optional): ``max_pool2d(x, 2)`` already computed by the producer of ``x`` (fused with its
        BatchNorm + ReLU); ``pool``: see DoubleConv.forward."""
        if pooled is None:
            x = ops.to_operator_layout(x, self.compute_dtype)
            pooled = ops.MaxPool2.apply(x)
        return self.maxpool_conv[1](pooled, pool=pool)


class Up(_HipBlock):
    """Up-sample (transposed conv or bilinear), centre-pad to the skip, concat skip-first, DoubleConv."""

    def __init__(self, in_channels, out_channels, bilinear=True, precision=None):
        super().__init__()
        self.precision = precision
        self.bilinear = bool(bilinear)
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2, precision=precision)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels, precision=precision)

    def forward(self, x1, x2, head=None, in_link=None, out_link=None):
        """``in_link`` / ``out_link`` (internal, optional): ops.BnLink from the block that produced ``x1`` / to the one
        consumer of the result (see DoubleConv.forward)."""
        dt = self.compute_dtype
        x1 = ops.to_operator_layout(x1, dt)
        x2 = ops.to_operator_layout(x2, dt)
        if self.bilinear:
            u = ops.Bilinear2x.apply(x1)
        else:
            u = ops.ConvT2x2.apply(x1, self.up.weight, self.up.bias, in_link)
        return self.conv(x2, u, head=head, out_link=out_link)


class OutConv(_HipBlock):
    """1x1 convolution with bias; returns NCHW fp32 logits."""

    def __init__(self, in_channels, out_channels, precision=None):
        super().__init__()
        self.precision = precision
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x, sigmoid=False):
        x = ops.to_operator_layout(x, self.compute_dtype)
        return ops.Head.apply(x, self.conv.weight, self.conv.bias, sigmoid)


def _pack_cache(model):
    """One batched weight-pack launch per parameter update for the whole model (ops.PackCache)."""
    dtype = model.compute_dtype
    cache = model.__dict__.get("_packs")
    if cache is None or cache.dtype != dtype:
        cache = ops.PackCache(dtype)
        for mod in model.modules():
            if isinstance(mod, DoubleConv):
                for idx in (0, 3):
                    w = mod.double_conv[idx].weight
                    co, ci = w.shape[0], w.shape[1]
                    ctot = (ci + 63) // 64 * 64
                    cache.add(w, ops.L.PACK_CONV_FWD, co, ctot)
                    cache.add(w, ops.L.PACK_CONV_DGRAD, ctot, co)
            elif isinstance(mod, Up) and not mod.bilinear:
                w = mod.up.weight
                ci, co = w.shape[0], w.shape[1]
                cache.add(w, ops.L.PACK_CONVT_FWD, co, ci)
                cache.add(w, ops.L.PACK_CONVT_DGRAD, ci, co)
        model.__dict__["_packs"] = cache
    cache.refresh(force=model.training)
    ops.set_active_packs(cache)
    return cache


_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _encoder(m, x):
    # x1..x4 feed the next level AND the decoder(s): their gradients meet in one buffer (ops.GradSink); each level
    # also hands its max-pooled activation to the next one (fused with its last BatchNorm + ReLU when training)
    x1, p1 = m.inc(x, pool=True)
    x1 = ops.share_grad(x1)
    x2, p2 = m.down1(x1, pooled=p1, pool=True)
    x2 = ops.share_grad(x2)
    x3, p3 = m.down2(x2, pooled=p2, pool=True)
    x3 = ops.share_grad(x3)
    x4, p4 = m.down3(x3, pooled=p3, pool=True)
    x4 = ops.share_grad(x4)
    x5 = m.down4(x4, pooled=p4)
    return x1, x2, x3, x4, x5


def _decoder_chain(up1, up2, up3, up4, feats, head):
    """up1..up4 of one decoder (reference src/model.py:103-107 / :195-199 / :203-207).  The output of up1..up3 has exactly
    one consumer, the next block's transposed convolution: a BnLink lets that kernel's data gradient produce the ReLU
    mask and the BatchNorm-backward sums of the block's last conv-BN-ReLU (training only)."""
    x1, x2, x3, x4, x5 = feats
    links = [ops.BnLink() if (torch.is_grad_enabled() and up.training and not up.bilinear) else None for up in (up1, up2, up3)]
    y = up1(x5, x4, out_link=links[0])
    y = up2(y, x3, in_link=links[0], out_link=links[1])
    y = up3(y, x2, in_link=links[1], out_link=links[2])
    return up4(y, x1, head=head, in_link=links[2])


class UNet(_HipBlock):
    def __init__(self, n_channels=3, n_classes=1, bilinear=False, precision=None):
        super().__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        self.precision = precision
        factor = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64, precision=precision)
        self.down1 = Down(64, 128, precision=precision)
        self.down2 = Down(128, 256, precision=precision)
        self.down3 = Down(256, 512, precision=precision)
        self.down4 = Down(512, 1024 // factor, precision=precision)
        self.up1 = Up(1024, 512 // factor, bilinear, precision=precision)
        self.up2 = Up(512, 256 // factor, bilinear, precision=precision)
        self.up3 = Up(256, 128 // factor, bilinear, precision=precision)
        self.up4 = Up(128, 64, bilinear, precision=precision)
        self.outc = OutConv(64, n_classes, precision=precision)

    def forward(self, x, sigmoid=False):
        """``sigmoid`` (extra, optional): apply the sigmoid inside the head kernel (the seg-only trainer and
        tester use probabilities, train.py / test.py); default = raw logits as the reference."""
        ops._require_cuda(x)
        _pack_cache(self)
        with _BatchedCounters():
            feats = _encoder(self, x)
            return _decoder_chain(self.up1, self.up2, self.up3, self.up4, feats, (self.outc, bool(sigmoid)))


class SegmentationUNet(_HipBlock):
    """UNet for multi-class semantic segmentation (Gear / Kolektor trainers, reference src/model.py:111-153): the UNet
    above with ``nn.Dropout2d(dropout)`` on the bottleneck x5.  Same attributes and state_dict keys as the reference."""

    def __init__(self, n_channels=3, n_classes=4, bilinear=False, dropout=0.1, precision=None):
        super().__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        self.precision = precision
        self.inc = DoubleConv(n_channels, 64, precision=precision)
        self.down1 = Down(64, 128, precision=precision)
        self.down2 = Down(128, 256, precision=precision)
        self.down3 = Down(256, 512, precision=precision)
        factor = 2 if bilinear else 1
        self.down4 = Down(512, 1024 // factor, precision=precision)
        self.dropout = nn.Dropout2d(dropout) if dropout > 0 else nn.Identity()      # holder of p / training flag
        self.up1 = Up(1024, 512 // factor, bilinear, precision=precision)
        self.up2 = Up(512, 256 // factor, bilinear, precision=precision)
        self.up3 = Up(256, 128 // factor, bilinear, precision=precision)
        self.up4 = Up(128, 64, bilinear, precision=precision)
        self.outc = OutConv(64, n_classes, precision=precision)

    def _bottleneck_dropout(self, x5):
        d = self.dropout
        if not isinstance(d, nn.Dropout2d) or not self.training or d.p == 0:
            return x5
        if d.p >= 1:
            return ops.ChannelDropout.apply(x5, torch.zeros(x5.shape[0], x5.shape[1], device=x5.device))
        # the noise tensor torch's feature dropout draws: [N, C, 1, 1] bernoulli(1-p) / (1-p) in the input's arithmetic
        noise = torch.empty((x5.shape[0], x5.shape[1], 1, 1), dtype=torch.float32, device=x5.device)
        noise.bernoulli_(1 - d.p).div_(1 - d.p)
        return ops.ChannelDropout.apply(x5, noise.view(x5.shape[0], x5.shape[1]))

    def forward(self, x, sigmoid=False):
        ops._require_cuda(x)
        _pack_cache(self)
        with _BatchedCounters():
            x1, x2, x3, x4, x5 = _encoder(self, x)
            x5 = self._bottleneck_dropout(x5)
            return _decoder_chain(self.up1, self.up2, self.up3, self.up4, (x1, x2, x3, x4, x5), (self.outc, bool(sigmoid)))


class AnomalyUNet(_HipBlock):
    """Shared encoder + reconstruction decoder + anomaly-segmentation decoder; both heads sigmoid."""

    def __init__(self, n_channels=3, bilinear=False, precision=None):
        super().__init__()
        self.n_channels = n_channels
        self.bilinear = bilinear
        self.precision = precision
        factor = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64, precision=precision)
        self.down1 = Down(64, 128, precision=precision)
        self.down2 = Down(128, 256, precision=precision)
        self.down3 = Down(256, 512, precision=precision)
        self.down4 = Down(512, 1024 // factor, precision=precision)
        for branch, n_out in (("recon", n_channels), ("seg", 1)):
            setattr(self, f"up1_{branch}", Up(1024, 512 // factor, bilinear, precision=precision))
            setattr(self, f"up2_{branch}", Up(512, 256 // factor, bilinear, precision=precision))
            setattr(self, f"up3_{branch}", Up(256, 128 // factor, bilinear, precision=precision))
            setattr(self, f"up4_{branch}", Up(128, 64, bilinear, precision=precision))
            setattr(self, f"outc_{branch}", OutConv(64, n_out, precision=precision))

    def _decode(self, feats, branch):
        ups = [getattr(self, f"up{i}_{branch}") for i in (1, 2, 3, 4)]
        return _decoder_chain(*ups, feats, (getattr(self, f"outc_{branch}"), True))

    two_streams = os.environ.get("UNET_TWO_STREAMS", "1") != "0"   # run the two independent decoders on two HIP streams (their kernels fill each
                           # other's ramp-up / tail; autograd replays each branch's backward on its own stream)

    def forward(self, x):
        with _BatchedCounters():
            return self._forward(x)

    def _forward(self, x):
        ops._require_cuda(x)
        _pack_cache(self)
        feats = _encoder(self, x)
        if not self.two_streams:
            return self._decode(feats, "recon"), self._decode(feats, "seg")
        main = torch.cuda.current_stream(x.device)
        side = _side_stream(x.device)
        side.wait_stream(main)                       # encoder features are ready
        with torch.cuda.stream(side):
            anomaly_map = self._decode(feats, "seg")
        reconstruction = self._decode(feats, "recon")
        main.wait_stream(side)
        anomaly_map.record_stream(main)
        if torch.is_grad_enabled() and (reconstruction.requires_grad or anomaly_map.requires_grad):
            setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if setter is not None and os.environ.get("UNET_KEEP_STREAM_WARNING", "0") == "0":
                setter(False)                    # (the engine checks when a backward pass STARTS: see _QuietStreamMismatch)
                reconstruction, anomaly_map = _QuietStreamMismatch.apply(reconstruction, anomaly_map)
        return reconstruction, anomaly_map


class _QuietStreamMismatch(torch.autograd.Function):
    """Identity on the two outputs of a two-stream AnomalyUNet forward.  The segmentation decoder's weight gradients are
    produced on the side stream while their AccumulateGrad nodes belong to the stream the parameters live on: autograd
    orders the two with an event wait -- exactly the dependency the optimiser step needs -- and warns about it.  The
    wait is wanted, the warning is not.  The engine looks at the switch when a backward pass starts, so the forward pass
    switches the warning off and this node -- part of that backward pass -- queues an engine callback that switches it
    back on when the pass has finished: the process-wide setting is off only between this model's forward and the end of
    its backward (``UNET_KEEP_STREAM_WARNING=1`` leaves it alone altogether)."""

    @staticmethod
    def forward(ctx, a, b):
        return a.view_as(a), b.view_as(b)

    @staticmethod
    def backward(ctx, ga, gb):
        setter = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if setter is not None:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: setter(True))
        return ga, gb
