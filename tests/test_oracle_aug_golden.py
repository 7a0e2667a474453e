"""CPU tests of the image-transform path: oracle/pil_oracle.py against the fixtures Pillow itself produced
(tests/golden/aug_pil.npz, tools/make_goldens_aug.py), bit for bit; and the HOST functions of libunet_hip.so that build
PIL's per-axis tables (no GPU needed to call them) against the oracle's tables."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from oracle import pil_oracle as PO

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from make_goldens_aug import JITTER_CASES, NEAREST_CASES, RESIZE_CASES, ROTATE_CASES  # noqa: E402  (case tables only)


@pytest.fixture(scope="module")
def g():
    return {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in load_golden("aug_pil").items()}


def test_resize_bilinear_matches_pil(g):
    for i, (h, w, c, oh, ow) in enumerate(RESIZE_CASES):
        got = PO.resize_bilinear(g[f"resize{i}_in"], oh, ow)
        assert np.array_equal(got, g[f"resize{i}_out"]), (i, h, w, oh, ow)


def test_resize_nearest_matches_pil(g):
    for i, (h, w, oh, ow) in enumerate(NEAREST_CASES):
        assert np.array_equal(PO.resize_nearest(g[f"nearest{i}_in"], oh, ow), g[f"nearest{i}_out"]), i


def test_rotation_and_flip_match_pil(g):
    for i, (h, w, angles) in enumerate(ROTATE_CASES):
        a = g[f"rotate{i}_in"]
        for j, ang in enumerate(angles):
            assert np.array_equal(PO.rotate_nearest(a, ang), g[f"rotate{i}_out"][j]), (i, ang)
            assert np.array_equal(PO.rotate_nearest(PO.hflip(a), ang), g[f"rotate{i}_flip_out"][j]), (i, ang, "flip")


def test_colour_conversions_match_pil(g):
    assert np.array_equal(PO.rgb2hsv(g["hsv_in"]), g["hsv_out"])
    assert np.array_equal(PO.hsv2rgb(g["hsv_in"]), g["hsv_back"])


def test_color_jitter_matches_pil(g):
    for name in ("jitter", "jitter2"):
        a = g[f"{name}_in"]
        for j, (o, b, c, s, h) in enumerate(JITTER_CASES):
            assert np.array_equal(PO.color_jitter(a, o, b, c, s, h), g[f"{name}_out"][j]), (name, j)
    for j, (o, b, c, s, h) in enumerate(JITTER_CASES):
        got = PO.to_tensor_normalize(PO.color_jitter(g["jitter_in"], o, b, c, s, h))
        assert np.array_equal(got, g["jitter_norm_out"][j]), j


def test_whole_training_transform_matches_pil(g):
    for i in range(2):
        p = g[f"full{i}_params"]
        flip, ang, order = bool(p[0]), float(p[1]), [int(v) for v in p[2:6]]
        got = PO.train_transform(g[f"full{i}_in"], 40, 32, flip, ang, order, *[float(v) for v in p[6:10]])
        assert np.array_equal(got, g["full_out"][i]), i
    m = PO.resize_bilinear(g["mask_in"], 40, 32).astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    assert np.array_equal(m, g["mask_out"])


# ------------------------------------------------------------------ the library's host-side table builders
def _lib():
    from tiaozhanbei_unet_amd import _lib as L
    return L.lib()


@pytest.mark.parametrize("in_size,out_size", [(900, 256), (1024, 256), (500, 512), (1270, 1408), (64, 128), (37, 16), (50, 50),
                                              (7, 300), (3000, 17)])
def test_library_resample_tables_equal_the_oracles(in_size, out_size):
    lib = _lib()
    bounds, kk = PO.bilinear_coeffs(in_size, out_size)
    ksize = lib.unet_resize_bilinear_ksize(in_size, out_size)
    assert ksize == kk.shape[1]
    b2 = np.zeros((out_size, 2), np.int32)
    k2 = np.zeros((out_size, ksize), np.int32)
    assert lib.unet_resize_bilinear_coeffs(in_size, out_size, b2.ctypes.data_as(C.c_void_p), k2.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(b2, bounds) and np.array_equal(k2, kk)
    idx = np.zeros(out_size, np.int32)
    assert lib.unet_resize_nearest_index(in_size, out_size, idx.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(idx, PO.nearest_index(in_size, out_size))


def test_rotation_matrix_fixed_point_equals_the_oracles():
    import math
    from tiaozhanbei_unet_amd.augment import rotation_matrix_fixed
    for (w, h, ang) in [(256, 256, 3.3), (512, 1408, -9.99), (48, 64, 0.0), (33, 57, 180.0), (100, 100, 359.5)]:
        m = PO.rotation_matrix(w, h, ang)
        fix = lambda v: int(math.floor(v * 65536.0 + 0.5))    # noqa: E731
        want = [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]
        assert rotation_matrix_fixed(w, h, ang) == want


def test_device_transform_draws_follow_the_reference_distributions():
    """augment.DeviceTransform.draw: the per-sample parameters of src/dataset.py:136-138 (flip p = 0.5, angle U(-10, 10),
    ColorJitter: a permutation of the four operations, brightness / contrast / saturation U(0.9, 1.1), hue U(-0.05, 0.05));
    seeded draws repeat, the Kolektor variant narrows the angle (src/kolektorsdd_dataset.py:139)."""
    from tiaozhanbei_unet_amd.augment import DeviceTransform, jitter_table, JITTER_DTYPE
    tf = DeviceTransform(256, train=True, seed=5)
    p = tf.draw(400)
    assert 0.35 < sum(p["flips"]) / 400 < 0.65
    assert all(-10.0 <= a <= 10.0 for a in p["angles"]) and max(p["angles"]) > 8 and min(p["angles"]) < -8
    assert all(sorted(o) == [0, 1, 2, 3] for o in p["orders"]) and len({tuple(o) for o in p["orders"]}) == 24
    for k in ("brightness", "contrast", "saturation"):
        assert all(0.9 <= v <= 1.1 for v in p[k]) and max(p[k]) > 1.08 and min(p[k]) < 0.92
    assert all(-0.05 <= v <= 0.05 for v in p["hue"])
    again = DeviceTransform(256, train=True, seed=5).draw(400)
    assert again == p
    k = DeviceTransform((1408, 512), train=True, degrees=5, seed=1).draw(200)
    assert all(-5.0 <= a <= 5.0 for a in k["angles"])
    tab = jitter_table(p["orders"][:3], p["brightness"][:3], p["contrast"][:3], p["saturation"][:3], [0.031, -0.05, 0.0])
    assert tab.dtype == JITTER_DTYPE and tab.itemsize == 32 and tab["hue_shift"].tolist() == [7, 244, 0]
