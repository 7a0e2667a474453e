#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Imports /root/reference/src/model.py and src/train_utils.py (read-only, no bytecode
written; `seaborn` -- absent from the image, used only by plotting helpers in
src/utils.py -- is stubbed), feeds them the key-seeded weights / inputs of
``oracle/weights.py`` and stores the reference's outputs.  Nothing from the
reference's source text is stored: the fixtures are inputs-by-seed + expected outputs.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_goldens.py

The fixtures travel to the GPU box; /root/reference does not.
"""
import json
import os
import sys
import types
from collections import OrderedDict

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))

import model as ref_model            # noqa: E402  (the reference)
import train_utils as ref_tu         # noqa: E402

from oracle import weights as W      # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    print(f"  wrote {name}.npz  ({sum(a.nbytes for a in conv.values()) / 1e3:.1f} kB raw)")


def load_into(module, state):
    sd = module.state_dict()
    assert list(sd.keys()) == list(state.keys()), "key order / names differ from the oracle's spec"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(state[k].shape), (k, sd[k].shape, state[k].shape)
    module.load_state_dict(state)


# ------------------------------------------------------------ key layout pins
def gen_specs():
    out = {}
    cases = {
        "unet_3_1": (ref_model.UNet(3, 1), ("unet", 3, 1, False)),
        "unet_3_4": (ref_model.UNet(3, 4), ("unet", 3, 4, False)),
        "unet_3_1_bilinear": (ref_model.UNet(3, 1, True), ("unet", 3, 1, True)),
        "anomaly_unet_3": (ref_model.AnomalyUNet(3), ("anomaly_unet", 3, 1, False)),
        "anomaly_unet_3_bilinear": (ref_model.AnomalyUNet(3, True), ("anomaly_unet", 3, 1, True)),
    }
    for name, (m, args) in cases.items():
        sd = m.state_dict()
        spec = W.state_spec(*args)
        assert list(spec.keys()) == list(sd.keys()), name
        assert all(tuple(sd[k].shape) == tuple(spec[k]) for k in sd), name
        out[name] = {
            "keys": [[k, list(v.shape)] for k, v in sd.items()],
            "n_params": int(sum(p.numel() for p in m.parameters())),
            "n_param_tensors": len(list(m.parameters())),
            "n_buffer_elems": int(sum(b.numel() for b in m.buffers())),
        }
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("  wrote state_dict_keys.json", {k: v["n_params"] for k, v in out.items()})


# ------------------------------------------------------------ G1: blocks
def run_block(tag, module, spec_args, inputs, seed=0):
    """fwd (train) + bwd with a seeded upstream gradient, then an eval fwd."""
    spec = W.block_spec(*spec_args)
    state = W.make_state(spec, seed)
    load_into(module, state)
    module.train()
    xs = [t.clone().requires_grad_(True) for t in inputs]
    y = module(*xs)
    gy = W.make_input(tag + ":gy", tuple(y.shape), seed)
    y.backward(gy)
    rec = OrderedDict(y=y)
    for i, x in enumerate(xs):
        rec[f"dx{i}"] = x.grad
    for k, p in module.named_parameters():
        rec["grad:" + k] = p.grad
    for k, b in module.named_buffers():
        rec["buf:" + k] = b
    module.eval()
    with torch.no_grad():
        rec["y_eval"] = module(*[t.clone() for t in inputs])   # uses the UPDATED running stats
    save(tag, **rec)


def gen_blocks():
    mk = W.make_input
    run_block("block_dc_3_64", ref_model.DoubleConv(3, 64), ("double_conv", 3, 64),
              [mk("dc_3_64:x", (2, 3, 16, 16))])
    run_block("block_dc_64_64", ref_model.DoubleConv(64, 64), ("double_conv", 64, 64),
              [mk("dc_64_64:x", (2, 64, 12, 20))])
    run_block("block_dc_128_64_mid64", ref_model.DoubleConv(128, 64, 64), ("double_conv", 128, 64, 64),
              [mk("dc_128_64_mid64:x", (1, 128, 9, 7))])
    run_block("block_down_64_128", ref_model.Down(64, 128), ("down", 64, 128),
              [mk("down_64_128:x", (2, 64, 13, 10))])
    run_block("block_up_128_64", ref_model.Up(128, 64, False), ("up", 128, 64, False),
              [mk("up_128_64:x1", (1, 128, 8, 8)), mk("up_128_64:x2", (1, 64, 17, 19))])
    run_block("block_up_128_64_even", ref_model.Up(128, 64, False), ("up", 128, 64, False),
              [mk("up_128_64_even:x1", (2, 128, 8, 16)), mk("up_128_64_even:x2", (2, 64, 16, 32))])
    run_block("block_up_128_64_bilinear", ref_model.Up(128, 64, True), ("up", 128, 64, True),
              [mk("up_128_64_bilinear:x1", (1, 64, 8, 8)), mk("up_128_64_bilinear:x2", (1, 64, 17, 19))])
    for co in (1, 3, 4):
        m = ref_model.OutConv(64, co)
        spec = W.block_spec("outconv", 64, co)
        load_into(m, W.make_state(spec, 0))
        x = mk(f"outc_{co}:x", (2, 64, 9, 11)).requires_grad_(True)
        logits = m(x)
        prob = torch.sigmoid(logits)
        g = mk(f"outc_{co}:gy", tuple(prob.shape))
        prob.backward(g)
        save(f"block_outconv_64_{co}", logits=logits, prob=prob, dx=x.grad,
             **{"grad:" + k: p.grad for k, p in m.named_parameters()},
             argmax=logits.argmax(1).to(torch.uint8))


# ------------------------------------------------------------ G2 / G6: full models
SIZES = {"s32": (2, 3, 32, 32), "s48x80": (1, 3, 48, 80), "s36x52": (1, 3, 36, 52)}


def gen_models():
    cfgs = [
        ("unet_3_1", lambda: ref_model.UNet(3, 1), ("unet", 3, 1, False)),
        ("unet_3_4", lambda: ref_model.UNet(3, 4), ("unet", 3, 4, False)),
        ("anomaly_unet_3", lambda: ref_model.AnomalyUNet(3), ("anomaly_unet", 3, 1, False)),
        ("anomaly_unet_3_bilinear", lambda: ref_model.AnomalyUNet(3, True), ("anomaly_unet", 3, 1, True)),
    ]
    for name, ctor, spec_args in cfgs:
        state = W.make_state(W.state_spec(*spec_args), 0)
        for sz, shape in SIZES.items():
            if "bilinear" in name and sz != "s36x52":
                continue
            x = W.make_input(f"model:{sz}", shape)
            rec = OrderedDict()
            for dt in (torch.float32, torch.float64):
                m = ctor()
                load_into(m, state)
                m = m.to(dt)
                sfx = "" if dt == torch.float32 else "_f64"
                m.train()
                with torch.no_grad():
                    out = m(x.to(dt))
                outs = out if isinstance(out, tuple) else (out,)
                for i, o in enumerate(outs):
                    rec[f"train_out{i}{sfx}"] = o
                if dt == torch.float32:
                    sd = m.state_dict()
                    rec["inc_bn0_running_mean"] = sd["inc.double_conv.1.running_mean"]
                    rec["inc_bn0_running_var"] = sd["inc.double_conv.1.running_var"]
                    rec["down4_bn1_running_var"] = sd["down4.maxpool_conv.1.double_conv.4.running_var"]
                    rec["nbt"] = sd["inc.double_conv.1.num_batches_tracked"]
                    rec["running_checksum"] = np.array(
                        [float(sum(v.double().sum() for k, v in sd.items() if "running" in k))])
                m2 = ctor()
                load_into(m2, state)
                m2 = m2.to(dt).eval()
                with torch.no_grad():
                    out = m2(x.to(dt))
                outs = out if isinstance(out, tuple) else (out,)
                for i, o in enumerate(outs):
                    rec[f"eval_out{i}{sfx}"] = o
                if name == "unet_3_4" and dt == torch.float32:
                    rec["eval_argmax"] = outs[0].argmax(1).to(torch.uint8)
                    rec["train_argmax"] = rec["train_out0"].argmax(1).to(torch.uint8)
            rec["weight_checksum"] = np.array(
                [float(sum(v.double().abs().sum() for v in state.values()))])
            save(f"model_{name}_{sz}", **rec)


# ------------------------------------------------------------ G3: CombinedLoss
def gen_losses():
    mk = W.make_input
    shape = (2, 3, 12, 10)
    recon = mk("loss:recon", shape, kind="uniform")
    image = mk("loss:image", shape)                      # ImageNet-normalised target
    amap = mk("loss:amap", (2, 1, 12, 10), kind="uniform")
    amap.view(-1)[:6] = torch.tensor([0.0, 1.0, 1e-30, 1 - 1e-7, 0.5, 1e-45])   # log clamp edges
    for tag, mask in (("binary", mk("loss:mask", (2, 1, 12, 10), kind="bernoulli")),
                      ("over255", mk("loss:mask", (2, 1, 12, 10), kind="bernoulli") / 255.0),
                      ("zeros", torch.zeros(2, 1, 12, 10))):
        for rw, sw in ((1.0, 1.0), (0.3, 2.5)):
            r = recon.clone().requires_grad_(True)
            a = amap.clone().requires_grad_(True)
            crit = ref_tu.CombinedLoss(recon_weight=rw, seg_weight=sw)
            d = crit(r, a, image, mask)
            d["total_loss"].backward()
            save(f"loss_combined_{tag}_{rw}_{sw}", total=d["total_loss"], recon=d["recon_loss"],
                 seg=d["seg_loss"], d_recon=r.grad, d_amap=a.grad, mask=mask)


# ------------------------------------------------------------ G4: SSIM
def gen_ssim():
    mk = W.make_input
    for c, hw in ((3, (40, 36)), (1, (20, 50)), (3, (64, 64))):
        a = mk(f"ssim:a{c}", (2, c) + hw, kind="uniform").requires_grad_(True)
        b = mk(f"ssim:b{c}", (2, c) + hw).requires_grad_(True)
        crit = ref_tu.SSIMLoss()
        v = crit(a, b)
        v.backward()
        same = ref_tu.SSIMLoss()(a.detach(), a.detach())
        save(f"ssim_c{c}_{hw[0]}x{hw[1]}", value=v, d_img1=a.grad, d_img2=b.grad, same=same,
             window=crit.window)


# ------------------------------------------------------------ G5: Adam trajectory
def gen_trajectory():
    for name, ctor, spec_args in (
            ("anomaly_unet_3", lambda: ref_model.AnomalyUNet(3), ("anomaly_unet", 3, 1, False)),):
        state = W.make_state(W.state_spec(*spec_args), 0)
        m = ctor()
        load_into(m, state)
        opt = ref_tu.get_optimizer(m, "adam", 1e-3, 1e-4)
        crit = ref_tu.CombinedLoss()
        image = W.make_input("traj:image", (4, 3, 32, 32))
        mask = W.make_input("traj:mask", (4, 1, 32, 32), kind="bernoulli")
        losses, rec = [], OrderedDict()
        m.train()
        for step in range(3):
            recon, amap = m(image)
            d = crit(recon, amap, image, mask)
            opt.zero_grad()
            d["total_loss"].backward()
            if step == 0:
                for k, p in m.named_parameters():
                    rec["gnorm:" + k] = p.grad.double().norm()
                    if p.numel() <= 4096:
                        rec["grad:" + k] = p.grad.clone()
                rec["grad:inc.double_conv.0.weight"] = m.inc.double_conv[0].weight.grad.clone()
            opt.step()
            losses.append([float(d["total_loss"]), float(d["recon_loss"]), float(d["seg_loss"])])
        sd = m.state_dict()
        rec["losses"] = np.array(losses)
        for k in ("inc.double_conv.0.weight", "outc_seg.conv.weight", "outc_recon.conv.bias",
                  "inc.double_conv.1.running_mean", "up4_seg.conv.double_conv.4.weight"):
            rec["final:" + k] = sd[k]
        rec["final_checksum"] = np.array([float(sum(v.double().abs().sum() for k, v in sd.items()))])
        save(f"trajectory_{name}", **rec)
        print("   losses", losses)


if __name__ == "__main__":
    torch.manual_seed(0)
    print("state_dict key layout"); gen_specs()
    print("blocks"); gen_blocks()
    print("losses"); gen_losses()
    print("ssim"); gen_ssim()
    print("full models"); gen_models()
    print("trajectory"); gen_trajectory()
