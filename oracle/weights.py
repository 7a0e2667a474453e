"""Deterministic, key-seeded state dicts for parity tests (test infrastructure).

Full AnomalyUNet weights are 173 MB -- too large to commit -- so goldens store
outputs only and both sides (``tools/make_goldens.py`` with the imported
reference, and the tests with the oracle / the HIP path) rebuild the SAME
weights from ``(key name, base seed)``.  numpy's PCG64 stream is platform
independent, so the GPU box regenerates bit-identical tensors.

``state_spec`` restates the parameter / buffer naming of the reference model
zoo (/root/reference/src/model.py:13-20 DoubleConv, :31-34 Down, :47-52 Up,
:72 OutConv, :85-95 UNet, :166-186 AnomalyUNet); the golden script asserts it
equals ``reference_model.state_dict()`` key-for-key and shape-for-shape.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch


def _double_conv_spec(spec, prefix, cin, cout, mid=None):
    mid = mid or cout
    for idx, (ci, co) in ((0, (cin, mid)), (3, (mid, cout))):
        spec[f"{prefix}.double_conv.{idx}.weight"] = (co, ci, 3, 3)
        bn = f"{prefix}.double_conv.{idx + 1}"
        spec[f"{bn}.weight"] = (co,)
        spec[f"{bn}.bias"] = (co,)
        spec[f"{bn}.running_mean"] = (co,)
        spec[f"{bn}.running_var"] = (co,)
        spec[f"{bn}.num_batches_tracked"] = ()


def _up_spec(spec, prefix, cin, cout, bilinear):
    if bilinear:
        _double_conv_spec(spec, f"{prefix}.conv", cin, cout, cin // 2)
    else:
        spec[f"{prefix}.up.weight"] = (cin, cin // 2, 2, 2)
        spec[f"{prefix}.up.bias"] = (cin // 2,)
        _double_conv_spec(spec, f"{prefix}.conv", cin, cout)


def _encoder_spec(spec, n_channels, bilinear):
    f = 2 if bilinear else 1
    _double_conv_spec(spec, "inc", n_channels, 64)
    for i, (ci, co) in enumerate(((64, 128), (128, 256), (256, 512), (512, 1024 // f)), 1):
        _double_conv_spec(spec, f"down{i}.maxpool_conv.1", ci, co)


def _decoder_spec(spec, suffix, bilinear):
    f = 2 if bilinear else 1
    for i, (ci, co) in enumerate(((1024, 512 // f), (512, 256 // f), (256, 128 // f), (128, 64)), 1):
        _up_spec(spec, f"up{i}{suffix}", ci, co, bilinear)


def state_spec(kind: str, n_channels: int = 3, n_classes: int = 1, bilinear: bool = False):
    """Ordered ``{key: shape}`` for ``kind`` in {'unet', 'anomaly_unet'}."""
    spec = OrderedDict()
    _encoder_spec(spec, n_channels, bilinear)
    if kind == "unet":
        _decoder_spec(spec, "", bilinear)
        spec["outc.conv.weight"] = (n_classes, 64, 1, 1)
        spec["outc.conv.bias"] = (n_classes,)
    elif kind == "anomaly_unet":
        _decoder_spec(spec, "_recon", bilinear)
        spec["outc_recon.conv.weight"] = (n_channels, 64, 1, 1)
        spec["outc_recon.conv.bias"] = (n_channels,)
        _decoder_spec(spec, "_seg", bilinear)
        spec["outc_seg.conv.weight"] = (1, 64, 1, 1)
        spec["outc_seg.conv.bias"] = (1,)
    else:
        raise ValueError(kind)
    return spec


def block_spec(block: str, *args):
    """Spec of a single building block with the reference's local key names."""
    spec = OrderedDict()
    if block == "double_conv":
        cin, cout, mid = (list(args) + [None])[:3]
        _double_conv_spec(spec, "m", cin, cout, mid)
    elif block == "down":
        cin, cout = args
        _double_conv_spec(spec, "m.maxpool_conv.1", cin, cout)
    elif block == "up":
        cin, cout, bilinear = args
        _up_spec(spec, "m", cin, cout, bilinear)
    elif block == "outconv":
        cin, cout = args
        spec["m.conv.weight"] = (cout, cin, 1, 1)
        spec["m.conv.bias"] = (cout,)
    else:
        raise ValueError(block)
    return OrderedDict((k[2:], v) for k, v in spec.items())  # strip "m."


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def tensor_for(key: str, shape, seed: int = 0) -> torch.Tensor:
    """One deterministic tensor.  Conv/ConvT/1x1 weights ~ U(+-1/sqrt(fan_in)) like
    torch's default init; BN affine and running stats are non-trivial on purpose."""
    g = _rng(key, seed)
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.long)
    if key.endswith("running_mean"):
        a = g.uniform(-0.2, 0.2, shape)
    elif key.endswith("running_var"):
        a = g.uniform(0.5, 1.5, shape)
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        if ".up.weight" in key:              # ConvTranspose2d: fan_in = Cout*k*k in torch
            fan_in = shape[1] * shape[2] * shape[3]
        b = 1.0 / np.sqrt(fan_in)
        a = g.uniform(-b, b, shape)
    elif key.endswith(".weight"):            # BN gamma
        a = g.uniform(0.5, 1.5, shape)
    elif ".double_conv." in key:             # BN beta
        a = g.uniform(-0.3, 0.3, shape)
    else:                                    # conv / convT bias
        a = g.uniform(-0.05, 0.05, shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def make_state(spec, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, tensor_for(k, shp, seed)) for k, shp in spec.items())


def make_input(name: str, shape, seed: int = 0, kind: str = "normal") -> torch.Tensor:
    g = _rng("input:" + name, seed)
    if kind == "normal":
        a = g.standard_normal(shape)
    elif kind == "uniform":
        a = g.uniform(0.0, 1.0, shape)
    elif kind == "bernoulli":
        a = (g.uniform(0.0, 1.0, shape) < 0.1).astype(np.float32)
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.asarray(a, dtype=np.float32))
