"""GPU parity tests of the image-transform path (csrc/augment.hip through tiaozhanbei_unet_amd/augment.py):
bit-exact against the fixtures Pillow produced (tests/golden/aug_pil.npz) and against oracle/pil_oracle.py at the
loaders' real sizes (MVTec 900 x 900 / 1024 x 1024 -> 256 x 256, KolektorSDD ~1270 x 500 -> 1408 x 512)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import pil_oracle as PO

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from make_goldens_aug import JITTER_CASES, NEAREST_CASES, RESIZE_CASES, ROTATE_CASES  # noqa: E402  (case tables only)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def g():
    return {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in load_golden("aug_pil").items()}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_resize_bilinear_bit_exact_against_pil_fixtures(g):
    from tiaozhanbei_unet_amd import augment as A
    for i, (h, w, c, oh, ow) in enumerate(RESIZE_CASES):
        got = A.resize_bilinear_u8(dev(g[f"resize{i}_in"][None]), oh, ow)[0].cpu().numpy()
        assert np.array_equal(got, g[f"resize{i}_out"]), (i, h, w, oh, ow)


def test_resize_nearest_bit_exact_against_pil_fixtures(g):
    from tiaozhanbei_unet_amd import augment as A
    for i, (h, w, oh, ow) in enumerate(NEAREST_CASES):
        got = A.resize_nearest_u8(dev(g[f"nearest{i}_in"][None]), oh, ow)[0].cpu().numpy()
        assert np.array_equal(got, g[f"nearest{i}_out"]), i


def test_flip_and_rotation_bit_exact_against_pil_fixtures(g):
    from tiaozhanbei_unet_amd import augment as A
    for i, (h, w, angles) in enumerate(ROTATE_CASES):
        n = len(angles)
        batch = dev(np.repeat(g[f"rotate{i}_in"][None], n, 0))
        got = A.flip_rotate_u8(batch, None, angles).cpu().numpy()
        assert np.array_equal(got, g[f"rotate{i}_out"]), i
        got = A.flip_rotate_u8(batch, [True] * n, angles).cpu().numpy()
        assert np.array_equal(got, g[f"rotate{i}_flip_out"]), (i, "flip")
        # a flip without rotation is the mirror image
        got = A.flip_rotate_u8(batch[:1], [True], None).cpu().numpy()[0]
        assert np.array_equal(got, g[f"rotate{i}_in"][:, ::-1])


def test_color_jitter_bit_exact_against_pil_fixtures(g):
    from tiaozhanbei_unet_amd import augment as A
    n = len(JITTER_CASES)
    tab = A.jitter_table([c[0] for c in JITTER_CASES], *[[c[k] for c in JITTER_CASES] for k in (1, 2, 3, 4)])
    got = A.color_jitter_normalize_u8(dev(np.repeat(g["jitter_in"][None], n, 0)), tab).cpu().numpy()
    assert np.array_equal(got, g["jitter_norm_out"])
    # uint8 results (read through an identity normalisation: u8 / 255 * 255 is not exact, so compare as the oracle does)
    for name in ("jitter", "jitter2"):
        got = A.color_jitter_normalize_u8(dev(np.repeat(g[f"{name}_in"][None], n, 0)), tab).cpu().numpy()
        want = np.stack([PO.to_tensor_normalize(g[f"{name}_out"][j]) for j in range(n)])
        assert np.array_equal(got, want), name
    # no jitter: ToTensor + Normalize only (the eval transform)
    got = A.color_jitter_normalize_u8(dev(g["jitter_in"][None])).cpu().numpy()[0]
    assert np.array_equal(got, PO.to_tensor_normalize(g["jitter_in"]))


def test_whole_training_transform_bit_exact_against_pil_fixtures(g):
    """src/dataset.py:134-141 end to end on two images of different sizes in one call (list input)."""
    from tiaozhanbei_unet_amd import augment as A
    tf = A.DeviceTransform((40, 32), train=True)
    ps = [g[f"full{i}_params"] for i in range(2)]
    params = {"flips": [bool(p[0]) for p in ps], "angles": [float(p[1]) for p in ps],
              "orders": [[int(v) for v in p[2:6]] for p in ps], "brightness": [float(p[6]) for p in ps],
              "contrast": [float(p[7]) for p in ps], "saturation": [float(p[8]) for p in ps], "hue": [float(p[9]) for p in ps]}
    got = tf([torch.from_numpy(g[f"full{i}_in"]) for i in range(2)], params, device=DEV).cpu().numpy()
    assert np.array_equal(got, g["full_out"])
    m = tf.masks([torch.from_numpy(g["mask_in"])], device=DEV).cpu().numpy()
    assert np.array_equal(m[0], g["mask_out"])
    # the eval transform: Resize + ToTensor + Normalize
    ev = A.DeviceTransform((40, 32), train=False)(torch.from_numpy(g["full0_in"][None]), device=DEV).cpu().numpy()[0]
    assert np.array_equal(ev, PO.to_tensor_normalize(PO.resize_bilinear(g["full0_in"], 40, 32)))


def test_hue_over_every_rgb_colour_against_the_oracle():
    """adjust_hue's RGB -> HSV -> (h + shift) -> RGB chain on all 2^24 colours, two shifts (one negative)."""
    from tiaozhanbei_unet_amd import augment as A
    v = np.arange(1 << 24, dtype=np.uint32)
    cols = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8).reshape(2, 2048, 4096, 3)
    for hue in (0.031, -0.05):
        tab = A.jitter_table([[3, -1, -1, -1]] * 2, [1.0] * 2, [1.0] * 2, [1.0] * 2, [hue] * 2)
        got = A.color_jitter_normalize_u8(dev(cols), tab, mean=(0, 0, 0), std=(1, 1, 1)).cpu().numpy()
        want = PO.hue(cols.reshape(-1, 3), hue).astype(np.float32).reshape(2, 2048, 4096, 3).transpose(0, 3, 1, 2) / np.float32(255.0)
        assert np.array_equal(got, want), hue


@pytest.mark.parametrize("src,dst", [((900, 900), (256, 256)), ((1024, 1024), (256, 256)), ((1270, 500), (1408, 512))],
                         ids=["mvtec900", "mvtec1024", "kolektor"])
def test_loader_sizes_against_the_oracle(src, dst):
    """The loaders' real geometry: a batch of 2, resize + flip + rotation + jitter + normalise, drawn parameters."""
    from tiaozhanbei_unet_amd import augment as A
    rng = np.random.default_rng(src[0] + dst[1])
    imgs = rng.integers(0, 256, (2,) + src + (3,), dtype=np.uint8)
    tf = A.DeviceTransform(dst, train=True, degrees=10 if dst[0] == 256 else 5, seed=7)
    p = tf.draw(2)
    got = tf(torch.from_numpy(imgs), p, device=DEV).cpu().numpy()
    for i in range(2):
        want = PO.train_transform(imgs[i], dst[0], dst[1], p["flips"][i], p["angles"][i], p["orders"][i], p["brightness"][i],
                                  p["contrast"][i], p["saturation"][i], p["hue"][i])
        assert np.array_equal(got[i], want), i
    masks = (rng.random((2,) + src + (1,)) < 0.1).astype(np.uint8) * 2
    near = A.resize_nearest_u8(dev(masks), dst[0], dst[1]).cpu().numpy()
    for i in range(2):
        assert np.array_equal(near[i], PO.resize_nearest(masks[i], dst[0], dst[1]))


def test_kolektor_gpu_preprocess_matches_the_host_loader_and_the_oracle(tmp_path):
    """kolektorsdd_dataset.GpuPreprocess on raw samples (src/kolektorsdd_dataset.py:133-155): the eval transform equals
    the host loader's tensors bit for bit (Pillow on the same files); the training transform equals the oracle replayed
    with the drawn parameters, the mask following the image's flip / rotation (sync_mask) or not (the reference's form)."""
    from tiaozhanbei_unet_amd import kolektorsdd_dataset as K
    root = K.write_synthetic_kolektorsdd(str(tmp_path / "kol"), n_folders=5, per_folder=4, size=(160, 64))
    host = K.KolektorSDDDataset(root, "val", image_size=(96, 32))
    tr, va, te, _ = K.get_kolektorsdd_dataloaders(root, batch_size=3, image_size=(96, 32), num_workers=0, raw=True)
    xs, ms, paths = next(iter(va))
    x, m = K.GpuPreprocess((96, 32), train=False)(xs, ms, device=DEV)
    for j in range(3):
        hx, hm, hp = host[j]
        assert hp == paths[j] and torch.equal(x[j].cpu(), hx) and torch.equal(m[j].cpu(), hm)
    xs, ms, _ = next(iter(tr))
    for sync in (True, False):
        pre = K.GpuPreprocess((96, 32), train=True, seed=3, sync_mask=sync)
        x, m = pre(xs, ms, device=DEV)
        pre.tf.gen.manual_seed(3)
        p = pre.tf.draw(3)
        for j in range(3):
            want = PO.train_transform(xs[j].numpy(), 96, 32, p["flips"][j], p["angles"][j], p["orders"][j], p["brightness"][j],
                                      p["contrast"][j], p["saturation"][j], p["hue"][j])
            assert np.array_equal(x[j].cpu().numpy(), want), (sync, j)
            wm = PO.resize_nearest(ms[j].numpy(), 96, 32)
            if sync:
                wm = PO.rotate_nearest(PO.hflip(wm) if p["flips"][j] else wm, p["angles"][j])
            assert np.array_equal(m[j].cpu().numpy(), wm[..., 0]), (sync, j, "mask")
