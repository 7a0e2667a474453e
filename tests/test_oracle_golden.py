"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz,
made by tools/make_goldens.py from /root/reference in the build container)."""
import json
import os

import pytest
import torch

from oracle import unet_oracle as O
from oracle import weights as W
from conftest import GOLDEN, load_golden


def close(a, b, atol, rtol=0.0, what=""):
    err = (a.double() - b.double()).abs()
    tol = atol + rtol * b.double().abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max()):.3e} (atol {atol}, rtol {rtol})"


def test_state_dict_key_layout():
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        pinned = json.load(f)
    cases = {"unet_3_1": ("unet", 3, 1, False), "unet_3_4": ("unet", 3, 4, False),
             "unet_3_1_bilinear": ("unet", 3, 1, True),
             "anomaly_unet_3": ("anomaly_unet", 3, 1, False),
             "anomaly_unet_3_bilinear": ("anomaly_unet", 3, 1, True)}
    for name, args in cases.items():
        spec = W.state_spec(*args)
        assert [[k, list(v)] for k, v in spec.items()] == pinned[name]["keys"]
    assert pinned["anomaly_unet_3"]["n_params"] == 43228228
    assert pinned["unet_3_1"]["n_params"] == 31037633
    assert len(pinned["anomaly_unet_3"]["keys"]) == 176


BLOCKS = [
    ("block_dc_3_64", ("double_conv", 3, 64), [("dc_3_64:x", (2, 3, 16, 16))]),
    ("block_dc_64_64", ("double_conv", 64, 64), [("dc_64_64:x", (2, 64, 12, 20))]),
    ("block_dc_128_64_mid64", ("double_conv", 128, 64, 64), [("dc_128_64_mid64:x", (1, 128, 9, 7))]),
    ("block_down_64_128", ("down", 64, 128), [("down_64_128:x", (2, 64, 13, 10))]),
    ("block_up_128_64", ("up", 128, 64, False),
     [("up_128_64:x1", (1, 128, 8, 8)), ("up_128_64:x2", (1, 64, 17, 19))]),
    ("block_up_128_64_even", ("up", 128, 64, False),
     [("up_128_64_even:x1", (2, 128, 8, 16)), ("up_128_64_even:x2", (2, 64, 16, 32))]),
    ("block_up_128_64_bilinear", ("up", 128, 64, True),
     [("up_128_64_bilinear:x1", (1, 64, 8, 8)), ("up_128_64_bilinear:x2", (1, 64, 17, 19))]),
]


def oracle_block(spec_args, state, xs, training, new_stats=None):
    kind = spec_args[0]
    st = {("m." + k): v for k, v in state.items()}
    if kind == "double_conv":
        return O.double_conv(st, "m", xs[0], training, new_stats)
    if kind == "down":
        return O.down(st, "m", xs[0], training, new_stats)
    if kind == "up":
        return O.up(st, "m", xs[0], xs[1], training, spec_args[3], new_stats)
    raise ValueError(kind)


@pytest.mark.parametrize("tag,spec_args,inputs", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_blocks_fwd_bwd(tag, spec_args, inputs):
    g = load_golden(tag)
    state = W.make_state(W.block_spec(*spec_args), 0)
    work = {k: (v.clone().requires_grad_(True) if O.is_trainable(k) else v) for k, v in state.items()}
    xs = [W.make_input(n, s).requires_grad_(True) for n, s in inputs]
    new_stats = {}
    y = oracle_block(spec_args, work, xs, True, new_stats)
    close(y, g["y"], 2e-5, 1e-5, "y")
    y.backward(W.make_input(tag + ":gy", tuple(y.shape)))
    for i, x in enumerate(xs):
        close(x.grad, g[f"dx{i}"], 5e-5, 1e-4, f"dx{i}")
    for k in state:
        if O.is_trainable(k):
            ref = g["grad:" + k]
            close(work[k].grad, ref, 1e-4 * max(1.0, float(ref.abs().max())), 1e-4, "grad:" + k)
    for k, v in new_stats.items():
        close(v.to(torch.float32), g["buf:" + k.removeprefix("m.")].to(torch.float32), 1e-5, 1e-5, k)
    st2 = dict(state)
    st2.update({k.removeprefix("m."): v for k, v in new_stats.items()})
    with torch.no_grad():
        ye = oracle_block(spec_args, st2, [x.detach() for x in xs], False)
    close(ye, g["y_eval"], 2e-5, 1e-5, "y_eval")


@pytest.mark.parametrize("co", [1, 3, 4])
def test_outconv(co):
    g = load_golden(f"block_outconv_64_{co}")
    state = W.make_state(W.block_spec("outconv", 64, co), 0)
    st = {"m." + k: v.clone().requires_grad_(True) for k, v in state.items()}
    x = W.make_input(f"outc_{co}:x", (2, 64, 9, 11)).requires_grad_(True)
    logits = O.out_conv(st, "m", x)
    close(logits, g["logits"], 1e-5, 1e-5)
    prob = torch.sigmoid(logits)
    prob.backward(W.make_input(f"outc_{co}:gy", tuple(prob.shape)))
    close(x.grad, g["dx"], 1e-5, 1e-4)
    close(st["m.conv.weight"].grad, g["grad:conv.weight"], 1e-4, 1e-4)
    close(st["m.conv.bias"].grad, g["grad:conv.bias"], 1e-4, 1e-4)
    assert torch.equal(logits.argmax(1).to(torch.uint8), g["argmax"])


MODELS = [("unet_3_1", ("unet", 3, 1, False)), ("unet_3_4", ("unet", 3, 4, False)),
          ("anomaly_unet_3", ("anomaly_unet", 3, 1, False)),
          ("anomaly_unet_3_bilinear", ("anomaly_unet", 3, 1, True))]
SIZES = {"s32": (2, 3, 32, 32), "s48x80": (1, 3, 48, 80), "s36x52": (1, 3, 36, 52)}


@pytest.mark.parametrize("name,spec_args", MODELS, ids=[m[0] for m in MODELS])
@pytest.mark.parametrize("sz", list(SIZES))
def test_full_models(name, spec_args, sz):
    if "bilinear" in name and sz != "s36x52":
        pytest.skip("not generated")
    g = load_golden(f"model_{name}_{sz}")
    state = W.make_state(W.state_spec(*spec_args), 0)
    x = W.make_input(f"model:{sz}", SIZES[sz])
    fwd = O.unet_forward if spec_args[0] == "unet" else O.anomaly_unet_forward
    for training, pre in ((True, "train"), (False, "eval")):
        new_stats = {}
        with torch.no_grad():
            out = fwd(state, x, training, spec_args[3], new_stats)
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            close(o, g[f"{pre}_out{i}"], 1e-4, 1e-4, f"{pre}_out{i}")
            # the reference's own fp32-vs-fp64 noise bounds the tolerance used above
            noise = float((g[f"{pre}_out{i}"].double() - g[f"{pre}_out{i}_f64"]).abs().max())
            assert noise < 1e-4
        if training:
            close(new_stats["inc.double_conv.1.running_mean"], g["inc_bn0_running_mean"], 1e-6, 1e-5)
            close(new_stats["inc.double_conv.1.running_var"], g["inc_bn0_running_var"], 1e-6, 1e-5)
            close(new_stats["down4.maxpool_conv.1.double_conv.4.running_var"],
                  g["down4_bn1_running_var"], 1e-5, 1e-4)
            assert int(new_stats["inc.double_conv.1.num_batches_tracked"]) == int(g["nbt"]) == 1
        if name == "unet_3_4":
            agree = (outs[0].argmax(1).to(torch.uint8) == g[f"{pre}_argmax"]).float().mean()
            assert float(agree) == 1.0


LOSS_TAGS = [f"loss_combined_{t}_{rw}_{sw}" for t in ("binary", "over255", "zeros")
             for rw, sw in ((1.0, 1.0), (0.3, 2.5))]


@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_combined_loss(tag):
    g = load_golden(tag)
    rw, sw = map(float, tag.split("_")[-2:])
    shape = (2, 3, 12, 10)
    recon = W.make_input("loss:recon", shape, kind="uniform").requires_grad_(True)
    image = W.make_input("loss:image", shape)
    amap = W.make_input("loss:amap", (2, 1, 12, 10), kind="uniform")
    amap.view(-1)[:6] = torch.tensor([0.0, 1.0, 1e-30, 1 - 1e-7, 0.5, 1e-45])
    amap.requires_grad_(True)
    d = O.combined_loss(recon, amap, image, g["mask"], rw, sw)
    close(d["total_loss"], g["total"], 1e-6, 1e-6)
    close(d["recon_loss"], g["recon"], 1e-6, 1e-6)
    close(d["seg_loss"], g["seg"], 1e-6, 1e-6)
    d["total_loss"].backward()
    close(recon.grad, g["d_recon"], 1e-8, 1e-5)
    # the reference's gradient is finite even at p = 0 / 1 / denormal (ATen clamps the
    # BCE-backward denominator at 1e-12): e.g. p=1,t=0 gives 0.25e12/240 = 1.04e9.
    assert bool(torch.isfinite(g["d_amap"]).all())
    close(amap.grad, g["d_amap"], 1e-7, 1e-4)


@pytest.mark.parametrize("c,hw", [(3, (40, 36)), (1, (20, 50)), (3, (64, 64))])
def test_ssim(c, hw):
    g = load_golden(f"ssim_c{c}_{hw[0]}x{hw[1]}")
    a = W.make_input(f"ssim:a{c}", (2, c) + hw, kind="uniform").requires_grad_(True)
    b = W.make_input(f"ssim:b{c}", (2, c) + hw).requires_grad_(True)
    v = O.ssim_loss(a, b)
    close(v, g["value"], 1e-6, 1e-6)
    v.backward()
    close(a.grad, g["d_img1"], 1e-7, 1e-4)
    close(b.grad, g["d_img2"], 1e-7, 1e-4)
    close(O.ssim_loss(a.detach(), a.detach()), g["same"], 1e-6)
    close(O.gaussian_window()[None, None].expand(c, 1, -1, -1), g["window"], 1e-9)


def test_adam_trajectory():
    """3 steps of train_epoch's body on AnomalyUNet 4x3x32x32 (train_utils.py:117-133)."""
    g = load_golden("trajectory_anomaly_unet_3")
    state = dict(W.make_state(W.state_spec("anomaly_unet", 3, 1, False), 0))
    image = W.make_input("traj:image", (4, 3, 32, 32))
    mask = W.make_input("traj:mask", (4, 1, 32, 32), kind="bernoulli")
    opt_state = {}
    losses = []
    torch.set_num_threads(8)
    for _ in range(3):
        state, l = O.train_step(state, opt_state, image, mask)
        losses.append([l["total_loss"], l["recon_loss"], l["seg_loss"]])
    close(torch.tensor(losses), g["losses"], 2e-4, 2e-4, "loss trajectory")
    for k in ("outc_seg.conv.weight", "outc_recon.conv.bias", "inc.double_conv.1.running_mean"):
        close(state[k], g["final:" + k], 5e-4, 1e-3, k)


# ------------------------------------------------------------------ round-2 fixtures (tools/make_goldens_r2.py)
def _val_batches():
    out = []
    for i, n in enumerate((2, 2, 1)):
        out.append({"image": W.make_input(f"val:image{i}", (n, 3, 32, 32)), "mask": torch.zeros(n, 1, 32, 32),
                    "label": torch.zeros(n, dtype=torch.long)})
    return out


def test_validate_epoch_all_normal_branch():
    """validate_epoch (reference train_utils.py:155-260) on a seeded all-normal loader: weighted losses, score maps."""
    g = load_golden("validate_epoch_all_normal")
    state = W.make_state(W.state_spec("anomaly_unet", 3, 1, False), 0)
    r = O.validate_pass(state, _val_batches())
    for k in ("total_loss", "recon_loss", "seg_loss"):
        assert abs(r[k] - float(g[k])) < 2e-6 * max(1.0, abs(float(g[k]))), k
    close(r["scores"], g["scores"], 1e-6, 1e-5, what="score maps")
    close(r["masks_pred"], g["masks_pred"], 1e-6, what="predicted masks")
    assert tuple(g["scores"].shape) == (5, 32, 32) and list(g["labels"]) == [0] * 5


@pytest.mark.parametrize("c,hw", [(3, (40, 36)), (1, (20, 50))])
def test_ssim_per_image(c, hw):
    """SSIMLoss(size_average=False) (reference train_utils.py:84-87): one loss per image."""
    g = load_golden(f"ssim_noavg_c{c}_{hw[0]}x{hw[1]}")
    a = W.make_input(f"ssim:a{c}", (2, c) + hw, kind="uniform").requires_grad_(True)
    b = W.make_input(f"ssim:b{c}", (2, c) + hw).requires_grad_(True)
    v = O.ssim_loss(a, b, size_average=False)
    assert tuple(v.shape) == (2,)
    close(v, g["value"], 2e-6, what="per-image ssim loss")
    (v * g["gy"]).sum().backward()
    close(a.grad, g["d_img1"], 1e-7, 1e-4, "d_img1")
    close(b.grad, g["d_img2"], 1e-7, 1e-4, "d_img2")


@pytest.mark.parametrize("sz,shape", [("s32", (2, 3, 32, 32)), ("s36x52", (1, 3, 36, 52))])
def test_segmentation_unet_forward(sz, shape):
    """SegmentationUNet (reference model.py:111-153): eval forward and train forward with dropout 0; argmax exact."""
    g = load_golden(f"model_segunet_3_4_{sz}")
    state = W.make_state(W.state_spec("unet", 3, 4, False), 0)
    x = W.make_input(f"model:{sz}", shape)
    with torch.no_grad():
        ev = O.segmentation_unet_forward(state, x, training=False)
        tr = O.segmentation_unet_forward(dict(state), x, training=True, new_stats={})
    close(ev, g["eval_out"], 2e-5, what="eval logits")
    close(tr, g["train_out_nodrop"], 2e-5, what="train logits")
    assert torch.equal(ev.argmax(1).to(torch.uint8), g["eval_argmax"])
    assert torch.equal(tr.argmax(1).to(torch.uint8), g["train_argmax_nodrop"])


def test_unet_seg_only_training_step():
    """BASELINE configs[1]: UNet(3,1) trained on focal(sigmoid(logits)) alone -- gradients of step 1 and the 3-step
    Adam loss trajectory of the reference."""
    g = load_golden("train_unet_segonly")
    state = W.make_state(W.state_spec("unet", 3, 1, False), 0)
    image = W.make_input("segonly:image", (2, 3, 32, 32))
    mask = W.make_input("segonly:mask", (2, 1, 32, 32), kind="bernoulli")
    work = {k: (v.clone().requires_grad_(True) if O.is_trainable(k) else v) for k, v in state.items()}
    amap = torch.sigmoid(O.unet_forward(work, image, True))
    close(amap, g["amap"], 2e-6, what="probabilities")
    loss = O.focal_loss(amap, mask)
    loss.backward()
    assert abs(float(loss) - float(g["losses"][0, 2])) < 1e-6
    for k in ("inc.double_conv.0.weight", "up4.conv.double_conv.3.weight", "outc.conv.weight", "outc.conv.bias"):
        ref = g["grad:" + k]
        rel = float((work[k].grad.double() - ref.double()).norm() / ref.double().norm())
        assert rel < 2e-3, f"{k}: {rel:.3e}"
    opt, losses = {}, []
    st = dict(state)
    for _ in range(3):
        st, d = O.train_step(st, opt, image, mask, model="unet", recon_weight=0.0)
        losses.append(d["total_loss"])
    assert float((torch.tensor(losses) - g["losses"][:, 0].float()).abs().max()) < 2e-4, losses
