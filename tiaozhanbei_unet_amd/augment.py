"""The loaders' image transform on the GPU: host side of csrc/augment.hip.

The reference builds its transforms from torchvision on PIL images (/root/reference/src/dataset.py:130-154,
src/kolektorsdd_dataset.py:133-155):

    Resize -> RandomHorizontalFlip(0.5) -> RandomRotation(10 | 5) -> ColorJitter(0.1, 0.1, 0.1, 0.05) -> ToTensor -> Normalize

Every one of those ends in a Pillow C kernel; libunet_hip restates that arithmetic on the device (bit-exact against PIL,
tests/golden/aug_*.npz), so loader workers only DECODE and ship the raw uint8 image.  What stays on the host is what is
host work in the reference too: the random draws (same distributions, same per-sample order as torchvision's
``get_params``; the RNG *stream* is this module's own generator -- torchvision is not installable here, so the stream
itself is "parity unpinned") and the parameter set-up PIL does in Python (``Image.rotate``'s matrix) or per axis
(the resampling tables, computed by the library's host functions).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .ops import _ptr, _require_cuda, _stream

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)

JITTER_DTYPE = np.dtype([("order", "<i4", (4,)), ("brightness", "<f4"), ("contrast", "<f4"), ("saturation", "<f4"),
                         ("hue_shift", "<i4")])          # struct unet_jitter_desc

_tables = {}


def _axis_tables(kind: str, in_size: int, out_size: int, device):
    """PIL's per-axis tables in device memory, cached per (kind, sizes, device): ('bilinear' -> bounds, kk, ksize),
    ('nearest' -> idx)."""
    key = (kind, in_size, out_size, device.index)
    hit = _tables.get(key)
    if hit is not None:
        return hit
    lib = L.lib()
    if kind == "bilinear":
        ksize = lib.unet_resize_bilinear_ksize(in_size, out_size)
        bounds = np.zeros((out_size, 2), dtype=np.int32)
        kk = np.zeros((out_size, ksize), dtype=np.int32)
        L.check(lib.unet_resize_bilinear_coeffs(in_size, out_size, bounds.ctypes.data_as(C.c_void_p),
                                                kk.ctypes.data_as(C.c_void_p)), "unet_resize_bilinear_coeffs")
        hit = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    else:
        idx = np.zeros(out_size, dtype=np.int32)
        L.check(lib.unet_resize_nearest_index(in_size, out_size, idx.ctypes.data_as(C.c_void_p)),
                "unet_resize_nearest_index")
        hit = (torch.from_numpy(idx).to(device),)
    _tables[key] = hit
    return hit


def _check_u8(images: torch.Tensor, what: str):
    _require_cuda(images)
    if images.dtype != torch.uint8 or images.dim() != 4:
        raise ValueError(f"{what} expects a uint8 [N, H, W, C] device tensor")
    return images.contiguous()


def resize_bilinear_u8(images: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """``transforms.Resize((out_h, out_w))`` of PIL images = ``Image.resize((out_w, out_h), BILINEAR)`` for a batch of
    equally sized uint8 [N, H, W, C] images (C = 1 or 3) on the device (unet_resize_bilinear_u8)."""
    images = _check_u8(images, "resize_bilinear_u8")
    n, h, w, c = images.shape
    dev = images.device
    out = torch.empty((n, out_h, out_w, c), dtype=torch.uint8, device=dev)
    xb = xk = yb = yk = None
    xs = ys = 0
    if out_w != w:
        xb, xk, xs = _axis_tables("bilinear", w, out_w, dev)
    if out_h != h:
        yb, yk, ys = _axis_tables("bilinear", h, out_h, dev)
    tmp = torch.empty((n, h, out_w, c), dtype=torch.uint8, device=dev) if (out_w != w and out_h != h) else None
    L.check(L.lib().unet_resize_bilinear_u8(_ptr(images), n, h, w, c, out_h, out_w, _ptr(xb), _ptr(xk), xs, _ptr(yb), _ptr(yk),
                                            ys, _ptr(tmp), _ptr(out), _stream()), "unet_resize_bilinear_u8")
    return out


def resize_nearest_u8(images: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """``Image.resize((out_w, out_h), NEAREST)`` (the Kolektor mask transform, src/kolektorsdd_dataset.py:116-117,147-150)."""
    images = _check_u8(images, "resize_nearest_u8")
    n, h, w, c = images.shape
    dev = images.device
    (yi,) = _axis_tables("nearest", h, out_h, dev)
    (xi,) = _axis_tables("nearest", w, out_w, dev)
    out = torch.empty((n, out_h, out_w, c), dtype=torch.uint8, device=dev)
    L.check(L.lib().unet_resize_nearest_u8(_ptr(images), n, h, w, c, out_h, out_w, _ptr(yi), _ptr(xi), _ptr(out), _stream()),
            "unet_resize_nearest_u8")
    return out


def rotation_matrix_fixed(w: int, h: int, angle: float) -> List[int]:
    """The six 16.16 fixed-point coefficients Pillow's ``Image.rotate(angle, NEAREST)`` ends up with: the Python half
    (Image.py: angle % 360, cos / sin rounded to 15 decimals, centre (w/2, h/2)) and the C half (Geometry.c affine_fixed:
    FIX(v) = floor(v * 65536 + 0.5), half-pixel terms folded into the translation)."""
    angle = angle % 360.0
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))

    out = [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]
    if any(abs(v) >= 1 << 31 for v in out):
        raise ValueError("rotation matrix outside 16.16 fixed point")
    return out


def flip_rotate_u8(images: torch.Tensor, flips=None, angles: Optional[Sequence[float]] = None) -> torch.Tensor:
    """RandomHorizontalFlip then RandomRotation (``Image.rotate(angle, NEAREST, expand=False)``, fill 0) with per-image
    flip flags / angles in degrees (unet_flip_rotate_u8)."""
    images = _check_u8(images, "flip_rotate_u8")
    n, h, w, c = images.shape
    dev = images.device
    fl = None if flips is None else torch.as_tensor(flips).to(device=dev, dtype=torch.uint8).contiguous()
    mats = None
    if angles is not None:
        mats = torch.tensor([rotation_matrix_fixed(w, h, float(a)) for a in angles], dtype=torch.int32).to(dev)
    out = torch.empty_like(images)
    L.check(L.lib().unet_flip_rotate_u8(_ptr(images), n, h, w, c, _ptr(fl), _ptr(mats), _ptr(out), _stream()),
            "unet_flip_rotate_u8")
    return out


def jitter_table(orders, brightness, contrast, saturation, hue) -> np.ndarray:
    """struct unet_jitter_desc[n] from torchvision-style parameters: ``orders[n]`` = permutation of (0 brightness, 1
    contrast, 2 saturation, 3 hue; -1 skips a slot), three enhancement factors and the hue factor in [-0.5, 0.5]
    (``np.uint8(hue_factor * 255)``: truncated toward zero, wrapped to a byte -- torchvision's adjust_hue)."""
    n = len(orders)
    rec = np.zeros(n, dtype=JITTER_DTYPE)
    for i in range(n):
        rec[i]["order"] = np.asarray(orders[i], dtype=np.int32)
        rec[i]["brightness"], rec[i]["contrast"], rec[i]["saturation"] = brightness[i], contrast[i], saturation[i]
        rec[i]["hue_shift"] = int(float(hue[i]) * 255) & 255
    return rec


def color_jitter_normalize_u8(images: torch.Tensor, jitter: Optional[np.ndarray] = None, mean=MEAN, std=STD) -> torch.Tensor:
    """ColorJitter (optional: ``jitter`` = jitter_table(...)) + ToTensor + Normalize: uint8 [N, H, W, 3] -> fp32 NCHW."""
    images = _check_u8(images, "color_jitter_normalize_u8")
    n, h, w, c = images.shape
    if c != 3:
        raise ValueError("color_jitter_normalize_u8 expects RGB images")
    dev = images.device
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
    lib = L.lib()
    desc = ws = None
    if jitter is not None:
        if jitter.dtype != JITTER_DTYPE or jitter.shape != (n,):
            raise ValueError("jitter: expected jitter_table(...) of the batch size")
        desc = torch.from_numpy(jitter.view(np.uint8).reshape(n, -1).copy()).to(dev)
        ws = torch.empty(lib.unet_color_jitter_workspace(n), dtype=torch.uint8, device=dev)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s_ = (C.c_float * 3)(*[float(v) for v in std])
    L.check(lib.unet_color_jitter_normalize_u8(_ptr(images), n, h, w, _ptr(desc), m, s_, _ptr(out), _ptr(ws),
                                               0 if ws is None else ws.numel(), _stream()), "unet_color_jitter_normalize_u8")
    return out


def _resize_any(images, out_h, out_w, device, nearest=False):
    """A stacked [N, H, W, C] tensor or a list of differently sized [H, W, C] tensors -> [N, out_h, out_w, C] on the device."""
    fn = resize_nearest_u8 if nearest else resize_bilinear_u8
    if isinstance(images, torch.Tensor):
        return fn(images.to(device, non_blocking=True), out_h, out_w)
    items = [t if t.dim() == 3 else t.unsqueeze(-1) for t in images]
    out = torch.empty((len(items), out_h, out_w, items[0].shape[-1]), dtype=torch.uint8, device=device)
    groups = {}
    for i, t in enumerate(items):
        groups.setdefault(tuple(t.shape), []).append(i)
    for idx in groups.values():
        batch = torch.stack([items[i] for i in idx]).to(device, non_blocking=True)
        out[torch.as_tensor(idx, device=device)] = fn(batch, out_h, out_w)
    return out


class DeviceTransform:
    """``get_transforms(image_size, is_train)[0]`` of the reference (src/dataset.py:130-146; ``degrees=5`` gives
    src/kolektorsdd_dataset.py:133-150) for batches of decoded uint8 RGB images, on the GPU.

    ``__call__(images)`` -> normalised fp32 NCHW; ``images`` is a stacked uint8 [N, H, W, 3] tensor or a list of
    [H, W, 3] tensors of any sizes (host or device).  ``draw(n)`` makes the per-sample random parameters (the same
    distributions and per-sample order as torchvision: flip, angle, then ColorJitter's permutation and four factors);
    pass them back as ``params`` to replay a batch (tests do)."""

    def __init__(self, size, train: bool, degrees=10.0, brightness=0.1, contrast=0.1, saturation=0.1, hue=0.05,
                 flip_p=0.5, seed: Optional[int] = None, mean=MEAN, std=STD):
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.train = bool(train)
        self.degrees, self.flip_p = float(degrees), float(flip_p)
        self.ranges = ((max(0.0, 1 - brightness), 1 + brightness), (max(0.0, 1 - contrast), 1 + contrast),
                       (max(0.0, 1 - saturation), 1 + saturation), (-hue, hue))
        self.mean, self.std = tuple(mean), tuple(std)
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def draw(self, n: int) -> dict:
        flips, angles, orders, fac = [], [], [], [[], [], [], []]
        for _ in range(n):
            flips.append(bool(torch.rand(1, generator=self.gen) < self.flip_p))
            angles.append(float(torch.empty(1).uniform_(-self.degrees, self.degrees, generator=self.gen)))
            orders.append(torch.randperm(4, generator=self.gen).tolist())
            for k, (lo, hi) in enumerate(self.ranges):
                fac[k].append(float(torch.empty(1).uniform_(lo, hi, generator=self.gen)))
        return {"flips": flips, "angles": angles, "orders": orders, "brightness": fac[0], "contrast": fac[1],
                "saturation": fac[2], "hue": fac[3]}

    def __call__(self, images, params: Optional[dict] = None, device="cuda") -> torch.Tensor:
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        x = _resize_any(images, self.size[0], self.size[1], device)
        jitter = None
        if self.train:
            p = params if params is not None else self.draw(x.shape[0])
            x = flip_rotate_u8(x, p["flips"], p["angles"])
            jitter = jitter_table(p["orders"], p["brightness"], p["contrast"], p["saturation"], p["hue"])
        return color_jitter_normalize_u8(x, jitter, self.mean, self.std)

    def masks(self, masks, device="cuda") -> torch.Tensor:
        """``get_transforms(...)[1]`` (src/dataset.py:148-151): Resize (bilinear) + ToTensor of the {0, 1} uint8 masks
        -> fp32 [N, 1, H, W] (values k/255: the reference's mask quirk, reproduced by feeding the same data)."""
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        m = _resize_any(masks, self.size[0], self.size[1], device)
        return (m.permute(0, 3, 1, 2).float() / 255.0).contiguous()
