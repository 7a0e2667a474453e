"""torchvision-free MVTec-AD reader with the reference's batch contract (host-side I/O).

Mirrors /root/reference/src/dataset.py: directory layout and label rules (:38-89), the batch dict keys
`image, mask, label, anomaly_type, image_path` (:121-127), the image transform Resize -> RandomHorizontalFlip ->
RandomRotation(10) -> ColorJitter(0.1, 0.1, 0.1, 0.05) -> ToTensor -> Normalize (:134-146) and the reference's mask quirk
-- masks are binarised to {0,1} uint8 and then scaled by 1/255 (:101-103,:149-152), so targets are {0, 1/255}.
torchvision is absent from this environment; on PIL images it delegates every one of those transforms to Pillow, so the
HOST path below calls the same Pillow functions (Image.resize / transpose / rotate, ImageEnhance, the HSV round trip).
With ``device_preprocess=True`` (what train.py / test.py select on a GPU) the workers only DECODE: samples carry the raw
uint8 image and mask, and the whole transform runs on the GPU, bit-exact to Pillow (augment.DeviceTransform,
csrc/augment.hip; train_utils._batches applies it per batch and restores the `image` / `mask` keys).
`write_synthetic_mvtec` creates a small MVTec-layout tree of PNGs for CLI plumbing tests; `--synthetic` in train.py uses it.
"""
from __future__ import annotations

import glob
import os
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance
from torch.utils.data import DataLoader, Dataset

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


def _adjust_hue(img: Image.Image, hue_factor: float) -> Image.Image:
    """torchvision's adjust_hue on a PIL image: H channel of the HSV image shifted with uint8 wrap-around."""
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h += np.array(int(hue_factor * 255) & 255).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def _color_jitter(img: Image.Image, rng: random.Random, brightness=0.1, contrast=0.1, saturation=0.1, hue=0.05):
    """transforms.ColorJitter on a PIL image: a random order of the four adjustments, factors uniform in
    [1 - x, 1 + x] (hue: [-hue, hue]) -- src/dataset.py:138."""
    order = [0, 1, 2, 3]
    rng.shuffle(order)
    b, c = rng.uniform(1 - brightness, 1 + brightness), rng.uniform(1 - contrast, 1 + contrast)
    s_, h = rng.uniform(1 - saturation, 1 + saturation), rng.uniform(-hue, hue)
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(b)
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(c)
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(s_)
        else:
            img = _adjust_hue(img, h)
    return img


_host_rng = random.Random()


def _image_tensor(img: Image.Image, size, train: bool, degrees=10.0) -> torch.Tensor:
    """The host image transform (src/dataset.py:134-146) on Pillow, in the reference's order."""
    img = img.resize((size[1], size[0]), Image.BILINEAR)
    if train:
        if _host_rng.random() < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        img = img.rotate(_host_rng.uniform(-degrees, degrees), resample=Image.NEAREST)
        img = _color_jitter(img, _host_rng)
    a = np.array(img, dtype=np.uint8).astype(np.float32).transpose(2, 0, 1) / 255.0
    return torch.from_numpy(np.ascontiguousarray((a - MEAN) / STD))


def _mask_tensor(mask: Image.Image, size) -> torch.Tensor:
    mask = mask.resize((size[1], size[0]), Image.BILINEAR)
    return torch.from_numpy(np.asarray(mask, dtype=np.float32)[None] / 255.0)   # ToTensor() on a {0,1} uint8 image


def collate_raw(samples):
    """Batch of ``device_preprocess`` samples: raw images / masks stay a list when their sizes differ."""
    out = {}
    for k in samples[0]:
        vals = [s_[k] for s_ in samples]
        if k in ("image_raw", "mask_raw"):
            out[k] = torch.stack(vals) if len({tuple(v.shape) for v in vals}) == 1 else vals
        elif isinstance(vals[0], torch.Tensor):
            out[k] = torch.stack(vals)
        elif isinstance(vals[0], (int, np.integer)):
            out[k] = torch.tensor(vals)
        else:
            out[k] = vals
    return out


class MVTecDataset(Dataset):
    def __init__(self, root_dir, category, split="train", image_size=256, is_train=True, device_preprocess=False, seed=None):
        """``device_preprocess``: samples carry ``image_raw`` (decoded uint8 HWC at its native size) and ``mask_raw``
        (uint8 HW1 in {0, 1}) instead of ``image`` / ``mask``; ``self.device_transform`` (augment.DeviceTransform) turns a
        batch of them into the reference's tensors on the GPU (train_utils._batches does)."""
        self.device_preprocess = bool(device_preprocess)
        self.size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        self.split, self.is_train = split, is_train
        self.device_transform = None
        if self.device_preprocess:
            from .augment import DeviceTransform
            self.device_transform = DeviceTransform(self.size, train=(split == "train"), degrees=10.0, seed=seed)
        self.image_paths, self.mask_paths, self.labels, self.anomaly_types = [], [], [], []
        cat = os.path.join(root_dir, category)
        if split == "train":
            self._add(sorted(glob.glob(os.path.join(cat, "train", "good", "*.png"))), 0, "good", None)
        else:
            test_dir, gt_dir = os.path.join(cat, "test"), os.path.join(cat, "ground_truth")
            self._add(sorted(glob.glob(os.path.join(test_dir, "good", "*.png"))), 0, "good", None)
            if not is_train and os.path.isdir(test_dir):
                for kind in sorted(os.listdir(test_dir)):
                    d = os.path.join(test_dir, kind)
                    if kind != "good" and os.path.isdir(d):
                        self._add(sorted(glob.glob(os.path.join(d, "*.png"))), 1, kind, os.path.join(gt_dir, kind))

    def _add(self, images, label, kind, mask_dir):
        for p in images:
            m = None
            if mask_dir is not None:
                cand = os.path.join(mask_dir, os.path.basename(p).replace(".png", "_mask.png"))
                m = cand if os.path.exists(cand) else None
            self.image_paths.append(p); self.mask_paths.append(m)
            self.labels.append(label); self.anomaly_types.append(kind)

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, i):
        img = Image.open(self.image_paths[i]).convert("RGB")
        if self.mask_paths[i]:
            m = (np.asarray(Image.open(self.mask_paths[i]).convert("L")) > 0).astype(np.uint8)
        else:
            m = np.zeros((img.size[1], img.size[0]), dtype=np.uint8)
        out = {"label": self.labels[i], "anomaly_type": self.anomaly_types[i], "image_path": self.image_paths[i]}
        if self.device_preprocess:
            out["image_raw"] = torch.from_numpy(np.array(img, dtype=np.uint8))
            out["mask_raw"] = torch.from_numpy(m[:, :, None].copy())
        else:
            out["image"] = _image_tensor(img, self.size, self.split == "train")
            out["mask"] = _mask_tensor(Image.fromarray(m), self.size)
        return out


class ShardSampler(torch.utils.data.Sampler):
    """Per-rank shard of a dataset for data-parallel training (rank r trains on its own images).

    Every epoch all ranks draw the SAME permutation (seed + epoch), pad it by wrap-around to a multiple of the world
    size -- so every rank runs the same number of steps and no collective is left waiting -- and rank r takes indices
    r, r + world, r + 2*world, ...: disjoint shards whose union is the dataset.  Call ``set_epoch`` before each epoch."""

    def __init__(self, n, rank=0, world=1, shuffle=True, seed=0):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world of {world}")
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = int(n), rank, world, shuffle, seed, 0
        self.per_rank = (self.n + world - 1) // world

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        total = self.per_rank * self.world
        while len(order) < total:
            order += order[:total - len(order)]
        return order[self.rank:total:self.world]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.per_rank


def get_dataloaders(root_dir, category, batch_size=16, image_size=256, num_workers=4, rank=0, world=1, seed=0,
                    device_preprocess=False):
    """(train_loader, test_loader) like the reference's (src/dataset.py:157-199).  With ``world > 1`` the train loader
    draws from this rank's ShardSampler shard (``loader.sampler.set_epoch(e)`` reshuffles); the test loader is whole.
    ``device_preprocess`` (default off: batches then carry the reference's ``image`` / ``mask`` keys): workers decode only,
    ``train_utils._batches`` (used by train_epoch / validate_epoch / test.py) runs the transform on the GPU."""
    train = MVTecDataset(root_dir, category, "train", image_size, is_train=True, device_preprocess=device_preprocess,
                         seed=seed * 7919 + rank)
    test = MVTecDataset(root_dir, category, "test", image_size, is_train=False, device_preprocess=device_preprocess)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=torch.cuda.is_available())
    if device_preprocess:
        kw["collate_fn"] = collate_raw
    if world > 1:
        sampler = ShardSampler(len(train), rank, world, shuffle=True, seed=seed)
        return DataLoader(train, sampler=sampler, **kw), DataLoader(test, shuffle=False, **kw)
    return DataLoader(train, shuffle=True, **kw), DataLoader(test, shuffle=False, **kw)


def get_available_categories(root_dir):
    if not os.path.isdir(root_dir):
        return []
    return sorted(d for d in os.listdir(root_dir)
                  if not d.startswith(".") and os.path.isdir(os.path.join(root_dir, d, "train"))
                  and os.path.isdir(os.path.join(root_dir, d, "test")))


def write_synthetic_mvtec(root_dir, category="bottle", n_train=8, n_good=4, n_bad=4, size=64, seed=0):
    """A tiny MVTec-layout tree of random PNGs (train/good, test/good, test/broken, ground_truth/broken)."""
    rng = np.random.default_rng(seed)
    cat = os.path.join(root_dir, category)

    def put(sub, n, with_mask=False):
        d = os.path.join(cat, *sub)
        os.makedirs(d, exist_ok=True)
        for i in range(n):
            img = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
            Image.fromarray(img).save(os.path.join(d, f"{i:03d}.png"))
            if with_mask:
                md = os.path.join(cat, "ground_truth", sub[-1])
                os.makedirs(md, exist_ok=True)
                m = np.zeros((size, size), dtype=np.uint8)
                y, x = rng.integers(0, size // 2, 2)
                m[y:y + size // 4, x:x + size // 4] = 255
                Image.fromarray(m).save(os.path.join(md, f"{i:03d}_mask.png"))

    put(("train", "good"), n_train)
    put(("test", "good"), n_good)
    put(("test", "broken"), n_bad, with_mask=True)
    return root_dir
