"""torchvision-free MVTec-AD reader with the reference's batch contract (host-side I/O).

Mirrors /root/reference/src/dataset.py: directory layout and label rules (:38-89), the batch dict keys
`image, mask, label, anomaly_type, image_path` (:121-127), resize + ImageNet normalisation (:134-146) and
the reference's mask quirk -- masks are binarised to {0,1} uint8 and then scaled by 1/255 (:101-103,:149-152),
so targets are {0, 1/255}.  torchvision is absent from this environment, so resize / rotation are PIL (ColorJitter is
not reproduced; flip and +-10 degree rotation are); flip + ToTensor + Normalize run on the GPU when the loaders are
built with ``device_preprocess`` (the default where a GPU is present): one unet_preprocess_u8 launch per batch.  `write_synthetic_mvtec` creates a small
MVTec-layout tree of PNGs for CLI plumbing tests; `--synthetic` in train.py uses it.
"""
from __future__ import annotations

import glob
import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32).reshape(3, 1, 1)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32).reshape(3, 1, 1)


def _image_u8(img: Image.Image, size, train: bool):
    """Host half of the image transform: resize (+ the train-time flip decision and +-10 degree rotation) -> uint8 HWC
    and the flip flag.  Flip, ToTensor and Normalize happen in _normalise (host) or in ONE unet_preprocess_u8 launch per
    batch (device, train_utils._batches): the workers then ship 1 byte per sample instead of 4."""
    img = img.resize((size[1], size[0]), Image.BILINEAR)
    flip = False
    if train:
        flip = random.random() < 0.5
        if flip:                        # (the rotation of the reference follows the flip: rotate the flipped image,
            img = img.transpose(Image.FLIP_LEFT_RIGHT)      #  then flip back so that the deferred flip reproduces it)
        img = img.rotate(random.uniform(-10.0, 10.0), resample=Image.NEAREST)
        if flip:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
    return np.ascontiguousarray(np.asarray(img, dtype=np.uint8)), flip


def _normalise(u8: np.ndarray, flip: bool) -> torch.Tensor:
    if flip:
        u8 = u8[:, ::-1]
    a = u8.astype(np.float32).transpose(2, 0, 1) / 255.0
    return torch.from_numpy(np.ascontiguousarray((a - MEAN) / STD))


def _image_tensor(img: Image.Image, size, train: bool) -> torch.Tensor:
    return _normalise(*_image_u8(img, size, train))


def _mask_tensor(mask: Image.Image, size) -> torch.Tensor:
    mask = mask.resize((size[1], size[0]), Image.BILINEAR)
    return torch.from_numpy(np.asarray(mask, dtype=np.float32)[None] / 255.0)   # ToTensor() on a {0,1} uint8 image


class MVTecDataset(Dataset):
    def __init__(self, root_dir, category, split="train", image_size=256, is_train=True, device_preprocess=False):
        """``device_preprocess``: samples carry ``image_u8`` (uint8 HWC) + ``flip`` instead of the normalised ``image``;
        ``train_utils._batches`` turns a batch of them into the same fp32 NCHW tensor with one unet_preprocess_u8
        launch (bit-identical to the host arithmetic)."""
        self.device_preprocess = bool(device_preprocess)
        self.size = (image_size, image_size) if isinstance(image_size, int) else tuple(image_size)
        self.split, self.is_train = split, is_train
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
        train_aug = self.split == "train"
        u8, flip = _image_u8(img, self.size, train_aug)
        out = {"mask": _mask_tensor(Image.fromarray(m), self.size), "label": self.labels[i],
               "anomaly_type": self.anomaly_types[i], "image_path": self.image_paths[i]}
        if self.device_preprocess:
            out["image_u8"], out["flip"] = torch.from_numpy(u8), int(flip)
        else:
            out["image"] = _normalise(u8, flip)
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
                    device_preprocess=None):
    """(train_loader, test_loader) like the reference's (src/dataset.py:157-199).  With ``world > 1`` the train loader
    draws from this rank's ShardSampler shard (``loader.sampler.set_epoch(e)`` reshuffles); the test loader is whole."""
    if device_preprocess is None:          # on the GPU box: flip + ToTensor + Normalize on the device
        device_preprocess = torch.cuda.is_available()
    train = MVTecDataset(root_dir, category, "train", image_size, is_train=True, device_preprocess=device_preprocess)
    test = MVTecDataset(root_dir, category, "test", image_size, is_train=False, device_preprocess=device_preprocess)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=torch.cuda.is_available())
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
