"""Fixtures for the image-transform path: outputs of Pillow itself -- the library the reference's
torchvision transforms run on for PIL inputs (/root/reference/src/dataset.py:130-154, src/kolektorsdd_dataset.py:133-155)
-- on seeded inputs, written to tests/golden/aug_*.npz.  Each stage is called exactly the way torchvision's PIL backend
calls it (torchvision/transforms/_functional_pil.py: resize -> Image.resize, rotate -> Image.rotate, adjust_brightness /
contrast / saturation -> ImageEnhance.*.enhance, adjust_hue -> convert("HSV") + uint8 add + convert back; to_tensor ->
/255, normalize -> sub/div in fp32).  torchvision is not installable in this image; Pillow 12.2.0 is.

Run in the build container: python tools/make_goldens_aug.py
"""
import os
import sys

import numpy as np
import PIL
from PIL import Image, ImageEnhance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def rnd_image(seed, h, w, c=3, smooth=False):
    g = np.random.default_rng(seed)
    a = g.integers(0, 256, (h, w, c), dtype=np.uint8)
    if smooth:            # low-frequency content + noise: exercises small differences instead of saturating clips
        yy, xx = np.mgrid[0:h, 0:w]
        base = 128 + 80 * np.sin(yy / 7.0)[..., None] * np.cos(xx[..., None] / 5.0 + np.arange(c))
        a = np.clip(base + g.normal(0, 20, (h, w, c)), 0, 255).astype(np.uint8)
    return a


def pil(a):
    return Image.fromarray(a if a.shape[2] == 3 else a[:, :, 0])


def arr(im):
    a = np.asarray(im)
    return a if a.ndim == 3 else a[:, :, None]


def adjust_hue(img, hue_factor):
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(int(hue_factor * 255) & 255).astype(np.uint8)     # == np.uint8(hue_factor * 255) incl. negatives
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def jitter(img, order, b, c, s, h):
    for op in order:
        if op == 0:
            img = ImageEnhance.Brightness(img).enhance(b)
        elif op == 1:
            img = ImageEnhance.Contrast(img).enhance(c)
        elif op == 2:
            img = ImageEnhance.Color(img).enhance(s)
        elif op == 3:
            img = adjust_hue(img, h)
    return img


def to_tensor_normalize(img):
    a = np.asarray(img, dtype=np.uint8).astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    m = np.asarray([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
    s = np.asarray([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)
    return (a - m) / s


RESIZE_CASES = [(97, 131, 3, 64, 48), (61, 40, 3, 64, 96), (120, 90, 1, 32, 32), (37, 50, 3, 20, 50), (50, 37, 3, 50, 16),
                (230, 170, 3, 32, 24)]
ROTATE_CASES = [(64, 48, [-10.0, -3.7, 0.0, 5.0, 9.99, 0.013]), (33, 57, [7.25, -9.5])]
JITTER_CASES = [([0, 1, 2, 3], 0.93, 1.07, 0.91, 0.031), ([3, 2, 1, 0], 1.1, 0.9, 1.1, -0.05), ([1, 3, 0, 2], 1.0, 1.04, 0.97, 0.0),
                ([2, 0, 3, 1], 0.9001, 1.0999, 1.0, 0.0499), ([1, 0, 2, 3], 1.05, 0.95, 1.05, -0.0123)]
NEAREST_CASES = [(127, 50, 141, 52), (90, 64, 32, 32), (37, 53, 101, 7)]


def main():
    os.makedirs(OUT, exist_ok=True)
    z = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, c, oh, ow) in enumerate(RESIZE_CASES):
        a = rnd_image(100 + i, h, w, c, smooth=i % 2 == 1)
        z[f"resize{i}_in"] = a
        z[f"resize{i}_out"] = arr(pil(a).resize((ow, oh), Image.BILINEAR))
    for i, (h, w, angles) in enumerate(ROTATE_CASES):
        a = rnd_image(200 + i, h, w)
        z[f"rotate{i}_in"] = a
        z[f"rotate{i}_angles"] = np.asarray(angles, np.float64)
        z[f"rotate{i}_out"] = np.stack([arr(pil(a).rotate(ang, resample=Image.NEAREST)) for ang in angles])
        z[f"rotate{i}_flip_out"] = np.stack([arr(pil(a).transpose(Image.FLIP_LEFT_RIGHT).rotate(ang, resample=Image.NEAREST))
                                            for ang in angles])
    a = rnd_image(300, 48, 40, smooth=True)
    z["jitter_in"] = a
    z["jitter_params"] = np.asarray([[*o, b, c, s, h] for (o, b, c, s, h) in JITTER_CASES], np.float64)
    z["jitter_out"] = np.stack([arr(jitter(pil(a), o, b, c, s, h)) for (o, b, c, s, h) in JITTER_CASES])
    z["jitter_norm_out"] = np.stack([to_tensor_normalize(jitter(pil(a), o, b, c, s, h)) for (o, b, c, s, h) in JITTER_CASES])
    a2 = rnd_image(301, 48, 40)                  # uniform noise: every blend clips somewhere
    z["jitter2_in"] = a2
    z["jitter2_out"] = np.stack([arr(jitter(pil(a2), o, b, c, s, h)) for (o, b, c, s, h) in JITTER_CASES])
    cols = rnd_image(302, 64, 64)
    z["hsv_in"] = cols
    z["hsv_out"] = arr(pil(cols).convert("HSV"))
    z["hsv_back"] = arr(Image.fromarray(cols, "HSV").convert("RGB"))
    for i, (h, w, oh, ow) in enumerate(NEAREST_CASES):
        m = np.random.default_rng(400 + i).integers(0, 3, (h, w, 1), dtype=np.uint8)
        z[f"nearest{i}_in"] = m
        z[f"nearest{i}_out"] = arr(pil(m).resize((ow, oh), Image.NEAREST))
    # the whole training transform of src/dataset.py:134-141 on two images of different sizes
    full = []
    for i, (h, w) in enumerate([(90, 70), (75, 110)]):
        a = rnd_image(500 + i, h, w, smooth=True)
        z[f"full{i}_in"] = a
        o, b, c, s, hh = JITTER_CASES[i]
        flip, ang = bool(i), (-6.5, 8.25)[i]
        im = pil(a).resize((32, 40), Image.BILINEAR)         # Resize((40, 32)): PIL takes (width, height)
        if flip:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)
        im = im.rotate(ang, resample=Image.NEAREST)
        full.append(to_tensor_normalize(jitter(im, o, b, c, s, hh)))
        z[f"full{i}_params"] = np.asarray([flip, ang, *o, b, c, s, hh], np.float64)
    z["full_out"] = np.stack(full)
    # the mask transform (src/dataset.py:148-151): Resize (bilinear, L image with values {0, 1}) + ToTensor
    m = (np.random.default_rng(600).random((90, 70, 1)) < 0.2).astype(np.uint8)
    z["mask_in"] = m
    z["mask_out"] = (np.asarray(pil(m).resize((32, 40), Image.BILINEAR), np.uint8).astype(np.float32) / np.float32(255.0))[None]
    path = os.path.join(OUT, "aug_pil.npz")
    np.savez_compressed(path, **z)
    print(path, os.path.getsize(path), "bytes, Pillow", PIL.__version__)


if __name__ == "__main__":
    sys.exit(main())
