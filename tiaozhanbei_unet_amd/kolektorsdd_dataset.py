"""torchvision-free KolektorSDD reader with the reference's contract (/root/reference/src/kolektorsdd_dataset.py).

Same sample discovery (`kos*/PartN.jpg` + `PartN_label.bmp`, :55-68), the same deterministic 70/15/15 split (sorted
list, `random.seed(42)` shuffle, :70-90), masks clamped to {0,1,2} (:108-111) and resized with NEAREST (:121-124),
`(image, mask, img_path)` samples (:126) and `get_kolektorsdd_dataloaders(...) -> (train, val, test, num_classes)`
(:157-223).  torchvision is absent from this environment: the host path resizes with PIL (what torchvision's Resize
calls on PIL images) and normalises; with ``raw=True`` the workers only DECODE and `GpuPreprocess` runs the whole
transform of :133-155 on the GPU (augment.py / csrc/augment.hip, bit-exact to Pillow): bilinear resize, horizontal flip,
RandomRotation(5), ColorJitter(0.1, 0.1, 0.1, 0.05), ToTensor, Normalize, and the NEAREST mask resize.
With ``world > 1`` the train loader draws from this rank's ``dataset.ShardSampler`` shard."""
from __future__ import annotations

import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from .dataset import MEAN, STD, ShardSampler

CLASS_NAMES = ["background", "defect_type_1", "defect_type_2"]


def list_samples(root_dir):
    """[(image path, mask path)] of every kos*/X.jpg that has an X_label.bmp, sorted (reference :47-68)."""
    if not os.path.exists(root_dir):
        raise ValueError(f"Dataset root directory not found: {root_dir}")
    out = []
    for folder in sorted(os.listdir(root_dir)):
        path = os.path.join(root_dir, folder)
        if os.path.isdir(path) and folder.startswith("kos"):
            for name in os.listdir(path):
                if name.endswith(".jpg"):
                    mask = os.path.join(path, name.replace(".jpg", "_label.bmp"))
                    if os.path.exists(mask):
                        out.append((os.path.join(path, name), mask))
    out.sort()
    return out


def split_samples(samples, split, train_split=0.7, val_split=0.15):
    """The reference's split: indices from the SORTED list length, then a `random.seed(42)` shuffle (:70-90)."""
    samples = list(samples)
    total = len(samples)
    train_end = int(total * train_split)
    val_end = int(total * (train_split + val_split))
    rng = random.Random(42)              # == random.seed(42); random.shuffle(...) without touching the global state
    rng.shuffle(samples)
    if split == "train":
        return samples[:train_end]
    if split == "val":
        return samples[train_end:val_end]
    if split == "test":
        return samples[val_end:]
    raise ValueError(f"Invalid split: {split}. Must be 'train', 'val', or 'test'")


class KolektorSDDDataset(Dataset):
    """``raw=True``: samples are the DECODED uint8 image [H, W, 3] and clamped mask [H, W, 1] at their native size (the
    parts differ: ~1240-1270 x 500) for ``GpuPreprocess``; otherwise normalised CHW fp32 tensors and resized long masks
    like the reference's eval transform (Resize + ToTensor + Normalize, :146-155)."""

    def __init__(self, root_dir, split="train", image_size=(1024, 512), train_split=0.7, val_split=0.15, raw=False):
        self.root_dir, self.split, self.image_size, self.raw = root_dir, split, tuple(image_size), raw
        self.class_names, self.num_classes = list(CLASS_NAMES), 3
        pairs = split_samples(list_samples(root_dir), split, train_split, val_split)
        self.image_paths = [p for p, _ in pairs]
        self.mask_paths = [m for _, m in pairs]

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        h, w = self.image_size
        img = Image.open(self.image_paths[idx]).convert("RGB")
        mask = np.clip(np.array(Image.open(self.mask_paths[idx]).convert("L")), 0, 2).astype(np.uint8)
        if self.raw:
            return torch.from_numpy(np.array(img, dtype=np.uint8)), torch.from_numpy(mask[:, :, None].copy()), self.image_paths[idx]
        img = img.resize((w, h), Image.BILINEAR)
        mask = Image.fromarray(mask, mode="L").resize((w, h), Image.NEAREST)
        mask = torch.from_numpy(np.array(mask)).long()
        a = np.array(img, dtype=np.uint8).astype(np.float32).transpose(2, 0, 1) / 255.0
        return torch.from_numpy((a - MEAN) / STD), mask, self.image_paths[idx]


def collate_raw(samples):
    """(images, masks, paths) of ``raw`` samples: stacked when the parts have one size, lists otherwise."""
    imgs, masks, paths = zip(*samples)
    same = len({tuple(t.shape) for t in imgs}) == 1
    return (torch.stack(imgs) if same else list(imgs)), (torch.stack(masks) if same else list(masks)), list(paths)


class GpuPreprocess:
    """get_kolektorsdd_transforms (reference :133-155) on the GPU for batches of ``raw`` samples: images Resize (bilinear)
    [-> RandomHorizontalFlip -> RandomRotation(5) -> ColorJitter when ``train``] -> ToTensor -> Normalize; masks Resize
    (NEAREST) -> long.  The reference applies the random flip / rotation to the image ONLY (its target_transform has
    none, :151-155), which misaligns image and mask; ``sync_mask=True`` (default) applies the same flip and rotation
    (nearest, fill 0 = background) to the mask, ``sync_mask=False`` reproduces the reference."""

    def __init__(self, image_size=(1024, 512), train=False, flip_p=0.5, degrees=5.0, seed=0, sync_mask=True):
        from .augment import DeviceTransform
        self.train, self.sync_mask = bool(train), bool(sync_mask)
        self.tf = DeviceTransform(tuple(image_size), train=train, degrees=degrees, flip_p=flip_p, seed=seed)

    def __call__(self, images_u8, masks=None, device="cuda"):
        from . import augment as A
        n = len(images_u8)
        params = self.tf.draw(n) if self.train else None
        x = self.tf(images_u8, params, device=device)
        if masks is None:
            return x
        m = A._resize_any(masks, self.tf.size[0], self.tf.size[1], x.device, nearest=True)
        if self.train and self.sync_mask:
            m = A.flip_rotate_u8(m, params["flips"], params["angles"])
        return x, m[..., 0].long()


def get_kolektorsdd_dataloaders(root_dir, batch_size=16, image_size=(1024, 512), num_workers=4, train_split=0.7,
                                val_split=0.15, rank=0, world=1, seed=0, raw=False):
    sets = [KolektorSDDDataset(root_dir, s, image_size, train_split, val_split, raw) for s in ("train", "val", "test")]
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=torch.cuda.is_available())
    if raw:
        kw["collate_fn"] = collate_raw
    if world > 1:
        train = DataLoader(sets[0], sampler=ShardSampler(len(sets[0]), rank, world, True, seed), **kw)
    else:
        train = DataLoader(sets[0], shuffle=True, **kw)
    return train, DataLoader(sets[1], shuffle=False, **kw), DataLoader(sets[2], shuffle=False, **kw), sets[0].num_classes


def write_synthetic_kolektorsdd(root_dir, n_folders=5, per_folder=4, size=(160, 64), seed=0):
    """A tiny KolektorSDD-layout tree (kosXX/PartN.jpg + PartN_label.bmp) for plumbing tests."""
    rng = np.random.default_rng(seed)
    for f in range(n_folders):
        d = os.path.join(root_dir, f"kos{f + 1:02d}")
        os.makedirs(d, exist_ok=True)
        for i in range(per_folder):
            img = rng.integers(0, 256, (size[0], size[1], 3), dtype=np.uint8)
            Image.fromarray(img).save(os.path.join(d, f"Part{i}.jpg"), quality=95)
            m = np.zeros(size, dtype=np.uint8)
            if (f + i) % 2:
                y, x = int(rng.integers(0, size[0] // 2)), int(rng.integers(0, size[1] // 2))
                m[y:y + size[0] // 4, x:x + size[1] // 4] = int(rng.integers(1, 5))       # values > 2 get clamped
            Image.fromarray(m, mode="L").save(os.path.join(d, f"Part{i}_label.bmp"))
    return root_dir
