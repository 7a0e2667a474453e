"""CPU-side checks (no GPU): the C-ABI library builds/loads and exports every symbol of
include/unet_hip.h, the drop-in modules keep the reference's state_dict layout and default init,
and the product refuses to compute without a GPU (no fallback)."""
import ctypes
import json
import os
import re
import sys

import pytest
import torch

from conftest import GOLDEN, ROOT


@pytest.fixture(scope="module")
def libpath():
    from tiaozhanbei_unet_amd import _lib
    return _lib.build()


def test_library_exports_every_declared_symbol(libpath):
    from tiaozhanbei_unet_amd import _lib
    header = open(os.path.join(ROOT, "include", "unet_hip.h")).read()
    declared = set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", header))
    handle = ctypes.CDLL(libpath)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/unet_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle.unet_abi_version.restype = ctypes.c_int32
    assert handle.unet_abi_version() == 1


def test_state_dict_layout_matches_reference():
    import tiaozhanbei_unet_amd as P
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        pinned = json.load(f)
    cases = {"unet_3_1": P.UNet(3, 1), "unet_3_4": P.UNet(3, 4), "unet_3_1_bilinear": P.UNet(3, 1, True),
             "anomaly_unet_3": P.AnomalyUNet(3), "anomaly_unet_3_bilinear": P.AnomalyUNet(3, True)}
    for name, m in cases.items():
        got = [[k, list(v.shape)] for k, v in m.state_dict().items()]
        assert got == pinned[name]["keys"], name
        assert sum(p.numel() for p in m.parameters()) == pinned[name]["n_params"]
        assert len(list(m.parameters())) == pinned[name]["n_param_tensors"]
    m = P.AnomalyUNet(3)
    assert m.n_channels == 3 and m.bilinear is False and P.UNet(3, 4).n_classes == 4


def test_default_init_follows_torch_seed():
    """Same registration order as the reference => same default init under torch.manual_seed."""
    import tiaozhanbei_unet_amd as P
    torch.manual_seed(7)
    a = P.DoubleConv(3, 64)
    torch.manual_seed(7)
    w0 = torch.nn.Conv2d(3, 64, 3, padding=1, bias=False).weight
    torch.nn.BatchNorm2d(64)
    w1 = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).weight
    assert torch.equal(a.double_conv[0].weight, w0) and torch.equal(a.double_conv[3].weight, w1)


def test_no_cpu_fallback():
    import tiaozhanbei_unet_amd as P
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.AnomalyUNet(3)(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.CombinedLoss()(torch.rand(1, 3, 4, 4), torch.rand(1, 1, 4, 4), torch.rand(1, 3, 4, 4), torch.zeros(1, 1, 4, 4))


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "tiaozhanbei_unet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", ""), f"{fn} mentions the oracle package"


def test_cli_flags_match_reference_contract():
    """Flag names/defaults of src/train.py:38-97 and src/test.py:26-61 (pinned here as data)."""
    from tiaozhanbei_unet_amd import test as test_cli
    from tiaozhanbei_unet_amd import train as train_cli
    a = train_cli.parse_args([])
    ref = dict(data_root="../datasets/mvtec_anomaly_detection", category="bottle", image_size=256,
               model="anomaly_unet", bilinear=False, epochs=100, batch_size=16, learning_rate=1e-3,
               weight_decay=1e-4, optimizer="adam", scheduler="cosine", recon_weight=1.0, seg_weight=1.0,
               use_ssim=False, num_workers=4, device="auto", seed=42, save_dir="../outputs", save_freq=10,
               resume=None, val_freq=5, debug=False, debug_samples=20)
    for k, v in ref.items():
        assert getattr(a, k) == v, k
    t = test_cli.parse_args(["--checkpoint", "x.pth"])
    assert t.pixel_thresholds == [0.3, 0.5, 0.7] and t.output_dir == "../test_results" and t.threshold is None


def test_mvtec_reader_contract(tmp_path):
    from tiaozhanbei_unet_amd.dataset import get_available_categories, get_dataloaders, write_synthetic_mvtec
    root = write_synthetic_mvtec(str(tmp_path), "bottle", n_train=3, n_good=2, n_bad=2, size=40)
    assert get_available_categories(root) == ["bottle"]
    train, test = get_dataloaders(root, "bottle", batch_size=2, image_size=32, num_workers=0, device_preprocess=False)
    b = next(iter(train))
    assert set(b) == {"image", "mask", "label", "anomaly_type", "image_path"}
    assert b["image"].shape == (2, 3, 32, 32) and b["mask"].shape == (2, 1, 32, 32)
    assert float(b["mask"].max()) == 0.0 and int(b["label"].sum()) == 0            # train split = good only
    labels, mx = [], 0.0
    for b in test:
        labels += b["label"].tolist(); mx = max(mx, float(b["mask"].max()))
    assert sorted(labels) == [0, 0, 1, 1]
    assert 0 < mx <= 1.0 / 255.0 + 1e-9, "masks are {0,1} uint8 scaled by 1/255 (reference quirk)"
    # device_preprocess form (what train.py / test.py select): workers only decode -- samples carry the raw uint8 image
    # and {0, 1} mask at their native size; collate_raw stacks equal sizes and keeps ragged ones as a list
    from tiaozhanbei_unet_amd import dataset as D
    for split, is_train in (("train", True), ("test", False)):
        ds = D.MVTecDataset(root, "bottle", split, 32, is_train, device_preprocess=True)
        assert ds.device_transform is not None and ds.device_transform.train == (split == "train")
        d = ds[len(ds) - 1]
        assert set(d) == {"image_raw", "mask_raw", "label", "anomaly_type", "image_path"}
        assert d["image_raw"].dtype == torch.uint8 and tuple(d["image_raw"].shape) == (40, 40, 3)
        assert d["mask_raw"].dtype == torch.uint8 and tuple(d["mask_raw"].shape) == (40, 40, 1)
        assert int(d["mask_raw"].max()) == (1 if split == "test" else 0)
        b = D.collate_raw([ds[0], ds[1]])
        assert tuple(b["image_raw"].shape) == (2, 40, 40, 3) and b["label"].tolist() == [ds[0]["label"], ds[1]["label"]]
        ragged = dict(ds[0]); ragged["image_raw"] = ragged["image_raw"][:30]
        assert isinstance(D.collate_raw([ragged, ds[1]])["image_raw"], list)
    # the host transform draws flip / rotation / colour jitter for the training split only
    ev = D.MVTecDataset(root, "bottle", "test", 32, False)
    assert torch.equal(ev[0]["image"], ev[0]["image"])
    tr = D.MVTecDataset(root, "bottle", "train", 32, True)
    assert not torch.equal(tr[0]["image"], tr[0]["image"])


def test_shard_sampler_gives_disjoint_equal_shards():
    """dataset.ShardSampler (data-parallel CLI): same permutation on every rank, disjoint strided shards whose union is
    the dataset, equal length on every rank (wrap-around padding), reshuffled by set_epoch."""
    from tiaozhanbei_unet_amd.dataset import ShardSampler
    for n, world in ((10, 2), (11, 4), (3, 8), (64, 8)):
        shards = []
        for r in range(world):
            s = ShardSampler(n, r, world, shuffle=True, seed=7)
            s.set_epoch(3)
            idx = list(iter(s))
            assert len(idx) == len(s) == (n + world - 1) // world
            shards.append(idx)
        flat = [i for sh in shards for i in sh]
        assert set(flat) == set(range(n))
        assert len(flat) - len(set(flat)) == len(shards[0]) * world - n       # only the padding repeats
        s0 = ShardSampler(n, 0, world, shuffle=True, seed=7)
        s0.set_epoch(4)
        if n > world:
            assert list(iter(s0)) != shards[0]
    assert list(iter(ShardSampler(5, 1, 2, shuffle=False))) == [1, 3, 0]


def test_kolektorsdd_reader_contract(tmp_path):
    """kolektorsdd_dataset.KolektorSDDDataset against the reference's rules (src/kolektorsdd_dataset.py:47-126):
    sorted discovery of kos*/X.jpg + X_label.bmp pairs, 70/15/15 split sizes from the sorted list then the seed-42
    shuffle (identical to `random.seed(42); random.shuffle(...)`), masks clamped to {0,1,2} and resized NEAREST,
    (image, mask, path) samples, 3 classes.  (torchvision is not importable here; on PIL images it delegates the resize
    to Pillow, which the host path calls and the oracle restates -- tests/test_oracle_aug_golden.py pins that.)"""
    import random
    import numpy as np
    import torch
    from tiaozhanbei_unet_amd import kolektorsdd_dataset as K
    root = K.write_synthetic_kolektorsdd(str(tmp_path / "kol"), n_folders=5, per_folder=4, size=(160, 64))
    pairs = K.list_samples(root)
    assert len(pairs) == 20 and pairs == sorted(pairs) and all(m.endswith("_label.bmp") for _, m in pairs)
    ref = list(pairs)
    random.seed(42)
    random.shuffle(ref)                                       # what the reference does (:79-80)
    got = {s: K.split_samples(pairs, s) for s in ("train", "val", "test")}
    assert got["train"] == ref[:14] and got["val"] == ref[14:17] and got["test"] == ref[17:]
    ds = K.KolektorSDDDataset(root, "train", image_size=(96, 32))
    img, mask, path = ds[0]
    assert img.shape == (3, 96, 32) and img.dtype == torch.float32 and mask.shape == (96, 32) and mask.dtype == torch.int64
    assert int(mask.max()) <= 2 and int(mask.min()) >= 0 and os.path.exists(path) and ds.num_classes == 3
    # raw samples (for GpuPreprocess): the decoded image / clamped mask at their NATIVE size; the host sample is Pillow's
    # resize of exactly those bytes (pinned against the oracle's restatement of Pillow's resampling arithmetic)
    from oracle import pil_oracle as PO
    raw = K.KolektorSDDDataset(root, "train", image_size=(96, 32), raw=True)
    u8, mask2, _ = raw[0]
    assert u8.dtype == torch.uint8 and u8.shape == (160, 64, 3) and mask2.shape == (160, 64, 1) and int(mask2.max()) <= 2
    want = torch.from_numpy(PO.to_tensor_normalize(PO.resize_bilinear(u8.numpy(), 96, 32)))
    assert torch.equal(img, want)
    assert torch.equal(mask, torch.from_numpy(PO.resize_nearest(mask2.numpy(), 96, 32)[..., 0]).long())
    xs, ms, ps = K.collate_raw([raw[0], raw[1]])
    assert tuple(xs.shape) == (2, 160, 64, 3) and tuple(ms.shape) == (2, 160, 64, 1) and len(ps) == 2
    tr, va, te, ncls = K.get_kolektorsdd_dataloaders(root, batch_size=4, image_size=(96, 32), num_workers=0)
    assert ncls == 3 and len(tr.dataset) == 14 and len(va.dataset) == 3 and len(te.dataset) == 3
    xb, mb, pb = next(iter(va))
    assert xb.shape == (3, 3, 96, 32) and mb.shape == (3, 96, 32) and len(pb) == 3
    tr2, _, _, _ = K.get_kolektorsdd_dataloaders(root, batch_size=2, image_size=(96, 32), num_workers=0, rank=1, world=2)
    assert len(tr2.sampler) == 7


def test_inline_asm_dpp_reductions_keep_their_wait_states():
    """The statistics epilogues reduce over DPP rows with inline-asm v_add_f32_dpp (csrc/igemm.hip, first.hip: row16_sum_n).
    hipcc pads nothing inside an asm statement, so the helper's step-major order must leave two instructions between the
    VALU write of a value and the DPP read of it: compile both sources to assembly (gfx950 cross-compile, no GPU) and scan
    every DPP add (tools/check_dpp_hazards.py).  The same scan covers the inline-asm y loads of the fused BatchNorm-backward
    epilogues: nothing may touch a load's destination VGPRs before its hand-counted s_waitcnt."""
    import shutil
    import subprocess
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dpp_hazards.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "igemm.hip: " in r.stdout and " 0 hazards" in r.stdout
    assert "inline-asm y loads, 0 used before their hand-counted wait" in r.stdout, r.stdout
    # ... and the kernels whose vmcnt waits are hand-counted over an exact DMA / load / store sequence do not spill (a scratch
    # access would be one more operation in that sequence)
    assert "igemm.hip: 16 kernels with hand-counted vmcnt waits, 0 with scratch traffic" in r.stdout, r.stdout
    assert "wgrad.hip: 2 kernels with hand-counted vmcnt waits, 0 with scratch traffic" in r.stdout, r.stdout


def test_fused_adam_takes_a_torch_adam_state_dict():
    """torch's load_state_dict replaces param_groups with the saved ones -- a torch.optim.Adam / AdamW checkpoint
    (what the reference writes, src/utils.py:39-44) has no `decoupled` key and may carry device `step` scalars.  The host
    side of the fix is checkable without a GPU: the key is derived from torch's `decoupled_weight_decay`, steps become
    host fp32 scalars."""
    import torch
    from tiaozhanbei_unet_amd.optim import FusedAdam
    for Opt, dec in ((torch.optim.Adam, False), (torch.optim.AdamW, True)):
        ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
        ref = Opt(ps, lr=1e-3, weight_decay=1e-4)
        for p in ps:
            p.grad = torch.randn_like(p)
        ref.step()
        ref.step()
        fused = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3, decoupled=not dec)
        fused.load_state_dict(ref.state_dict())
        assert fused.param_groups[0]["decoupled"] is dec
        assert fused.param_groups[0]["weight_decay"] == 1e-4
        for st in fused.state.values():
            assert st["step"].dtype == torch.float32 and st["step"].device.type == "cpu" and float(st["step"]) == 2.0
        # and its own state_dict round-trips (the key survives)
        again = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
        again.load_state_dict(fused.state_dict())
        assert again.param_groups[0]["decoupled"] is dec
