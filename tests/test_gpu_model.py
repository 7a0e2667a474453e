"""GPU parity tests, module level: the drop-in nn.Modules (HIP path) against the goldens produced by
the reference itself (tests/golden, tools/make_goldens.py) and against the CPU oracle.

Bars (north_star): fp32 forward within 1e-3 of the reference, argmax mask indices bit-exact; gradients
with norm-based bounds (the reference's own fp32-vs-fp64 gradient noise reaches 7.8e-3
L2-relative on the full net); bf16 is the throughput mode with its own looser, documented bound."""
import os

import pytest
import torch

from conftest import load_golden
from oracle import unet_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxabs(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def build(block, spec_args, precision):
    import tiaozhanbei_unet_amd as P
    kind = spec_args[0]
    if kind == "double_conv":
        m = P.DoubleConv(*spec_args[1:], precision=precision)
    elif kind == "down":
        m = P.Down(*spec_args[1:], precision=precision)
    elif kind == "up":
        m = P.Up(*spec_args[1:], precision=precision)
    else:
        raise ValueError(kind)
    state = W.make_state(W.block_spec(*spec_args), 0)
    assert list(m.state_dict().keys()) == list(state.keys()), "state_dict key layout differs from the reference's"
    m.load_state_dict(state)
    return m.to(DEV)


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


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tag,spec_args,inputs", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_blocks_against_reference_goldens(tag, spec_args, inputs, precision):
    g = load_golden(tag)
    m = build(tag, spec_args, precision)
    m.train()
    xs = [W.make_input(n, s).to(DEV).requires_grad_(True) for n, s in inputs]
    y = m(*xs)
    assert tuple(y.shape) == tuple(g["y"].shape)
    # bf16: activations/gradients are rounded to 8 bits at every layer boundary and ReLU gates flip, so
    # the input gradient (two BN backward passes deep) carries the loosest bound
    fwd_tol, grad_tol, stat_tol = (1e-4, 2e-4, 1e-4) if precision == "fp32" else (6e-2, 0.12, 2e-2)
    dx_tol = grad_tol
    assert maxabs(y, g["y"]) < fwd_tol * max(1.0, float(g["y"].abs().max())), f"fwd {maxabs(y, g['y']):.3e}"
    y.backward(W.make_input(tag + ":gy", tuple(y.shape)).to(DEV))
    for i, x in enumerate(xs):
        assert l2rel(x.grad, g[f"dx{i}"]) < dx_tol, f"dx{i} l2rel {l2rel(x.grad, g[f'dx{i}']):.3e}"
    for k, prm in m.named_parameters():
        assert prm.grad is not None, k
        assert l2rel(prm.grad, g["grad:" + k]) < grad_tol, f"grad:{k} l2rel {l2rel(prm.grad, g['grad:' + k]):.3e}"
    for k, b in m.named_buffers():
        ref = g["buf:" + k]
        if k.endswith("num_batches_tracked"):
            assert int(b) == int(ref) == 1
        else:
            assert maxabs(b, ref) < stat_tol * max(1.0, float(ref.abs().max())), k
    m.eval()
    with torch.no_grad():
        ye = m(*[x.detach() for x in xs])
    assert maxabs(ye, g["y_eval"]) < fwd_tol * max(1.0, float(g["y_eval"].abs().max()))


MODELS = [("unet_3_1", ("unet", 3, 1, False)), ("unet_3_4", ("unet", 3, 4, False)),
          ("anomaly_unet_3", ("anomaly_unet", 3, 1, False)),
          ("anomaly_unet_3_bilinear", ("anomaly_unet", 3, 1, True))]
SIZES = {"s32": (2, 3, 32, 32), "s48x80": (1, 3, 48, 80), "s36x52": (1, 3, 36, 52)}


def make_model(spec_args, precision):
    import tiaozhanbei_unet_amd as P
    kind, nch, ncls, bil = spec_args
    m = P.UNet(nch, ncls, bil, precision=precision) if kind == "unet" else P.AnomalyUNet(nch, bil, precision=precision)
    state = W.make_state(W.state_spec(*spec_args), 0)
    assert list(m.state_dict().keys()) == list(state.keys())
    m.load_state_dict(state)
    return m.to(DEV), state


@pytest.mark.parametrize("name,spec_args", MODELS, ids=[m[0] for m in MODELS])
@pytest.mark.parametrize("sz", list(SIZES))
def test_full_model_forward_fp32_within_1e_3(name, spec_args, sz):
    """The north-star bar: forward within 1e-3 (fp32) of reference model.py on identical inputs,
    argmax mask indices bit-exact; odd sizes (36x52) exercise the centre-pad path."""
    if "bilinear" in name and sz != "s36x52":
        pytest.skip("no golden")
    g = load_golden(f"model_{name}_{sz}")
    m, _ = make_model(spec_args, "fp32")
    x = W.make_input(f"model:{sz}", SIZES[sz]).to(DEV)
    for mode in ("train", "eval"):
        m.train(mode == "train")
        m.load_state_dict(W.make_state(W.state_spec(*spec_args), 0))   # golden eval used the pristine stats
        with torch.no_grad():
            out = m(x)
        outs = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(outs):
            assert o.dtype == torch.float32 and tuple(o.shape) == tuple(g[f"{mode}_out{i}"].shape)
            err = maxabs(o, g[f"{mode}_out{i}"])
            assert err < 1e-3, f"{mode}_out{i}: max abs err {err:.3e}"
            assert err < 2e-4, f"{mode}_out{i}: fp32 path looser than expected ({err:.3e})"
        if name == "unet_3_4":
            assert torch.equal(outs[0].argmax(1).to(torch.uint8).cpu(), g[f"{mode}_argmax"]), "argmax indices differ"
        if mode == "train":
            sd = m.state_dict()
            assert maxabs(sd["inc.double_conv.1.running_mean"], g["inc_bn0_running_mean"]) < 1e-5
            assert maxabs(sd["inc.double_conv.1.running_var"], g["inc_bn0_running_var"]) < 1e-5
            assert maxabs(sd["down4.maxpool_conv.1.double_conv.4.running_var"], g["down4_bn1_running_var"]) < 1e-4
            assert int(sd["inc.double_conv.1.num_batches_tracked"]) == 1
    if isinstance(out, tuple):
        thr_ref = (g["eval_out1"] > 0.5)
        assert torch.equal((outs[1].cpu() > 0.5), thr_ref), "thresholded anomaly mask differs"


@pytest.mark.parametrize("sz", ["s32", "s36x52"])
def test_full_model_forward_bf16_documented_bound(sz):
    """bf16 throughput mode: SURVEY section 4 measured 1.2e-3..9e-3 for a bf16 run of the reference itself."""
    g = load_golden(f"model_anomaly_unet_3_{sz}")
    m, _ = make_model(("anomaly_unet", 3, 1, False), "bf16")
    x = W.make_input(f"model:{sz}", SIZES[sz]).to(DEV)
    m.eval()
    with torch.no_grad():
        recon, amap = m(x)
    agree = float(((amap.cpu() > 0.5) == (g["eval_out1"] > 0.5)).float().mean())
    print(f"[bf16 forward {sz}] max|recon err| {maxabs(recon, g['eval_out0']):.3e}  max|amap err| "
          f"{maxabs(amap, g['eval_out1']):.3e}  mask agreement {agree:.4f}")
    # SURVEY section 4's suggested bound (2e-2 abs, >= 98 % thresholded-mask agreement)
    assert maxabs(recon, g["eval_out0"]) < 2e-2 and maxabs(amap, g["eval_out1"]) < 2e-2
    assert agree >= 0.98


def test_training_trajectory_matches_reference():
    """3 Adam steps of train_epoch's body (train_utils.py:117-133) on AnomalyUNet 4x3x32x32, fp32."""
    import tiaozhanbei_unet_amd as P
    g = load_golden("trajectory_anomaly_unet_3")
    m, _ = make_model(("anomaly_unet", 3, 1, False), "fp32")
    opt = P.get_optimizer(m, "adam", 1e-3, 1e-4)
    crit = P.CombinedLoss()
    image = W.make_input("traj:image", (4, 3, 32, 32)).to(DEV)
    mask = W.make_input("traj:mask", (4, 1, 32, 32), kind="bernoulli").to(DEV)
    m.train()
    losses = []
    for step in range(3):
        recon, amap = m(image)
        d = crit(recon, amap, image, mask)
        opt.zero_grad()
        d["total_loss"].backward()
        if step == 0:
            worst = 0.0
            for k, prm in m.named_parameters():
                ref = g["gnorm:" + k]
                got = float(prm.grad.double().norm())
                assert abs(got - float(ref)) <= 2e-2 * float(ref) + 1e-7, f"grad norm {k}: {got} vs {float(ref)}"
                if "grad:" + k in g:
                    worst = max(worst, l2rel(prm.grad, g["grad:" + k]))
            assert worst < 2e-2, f"worst per-tensor gradient L2-rel error {worst:.3e}"
        opt.step()
        losses.append([float(d["total_loss"]), float(d["recon_loss"]), float(d["seg_loss"])])
    got, want = torch.tensor(losses), g["losses"].float()
    assert float((got - want).abs().max()) < 2e-3, f"loss trajectory {got.tolist()} vs {want.tolist()}"
    sd = m.state_dict()
    for k in ("outc_seg.conv.weight", "outc_recon.conv.bias", "inc.double_conv.1.running_mean"):
        assert maxabs(sd[k], g["final:" + k]) < 2e-3, k


def test_train_epoch_contract_and_determinism():
    """train_epoch returns the reference's keys; two identical runs are bitwise equal (ordered reductions)."""
    import tiaozhanbei_unet_amd as P
    image = W.make_input("ep:image", (4, 3, 32, 32))
    mask = W.make_input("ep:mask", (4, 1, 32, 32), kind="bernoulli")
    loader = [{"image": image[:2], "mask": mask[:2]}, {"image": image[2:], "mask": mask[2:]}]
    results, finals = [], []
    for _ in range(2):
        m, _ = make_model(("anomaly_unet", 3, 1, False), "bf16")
        opt = P.get_optimizer(m)
        out = P.train_epoch(m, loader, P.CombinedLoss(), opt, torch.device(DEV), 0)
        assert set(out) == {"total_loss", "recon_loss", "seg_loss"}
        assert abs(out["total_loss"] - (out["recon_loss"] + out["seg_loss"])) < 1e-5
        results.append(out)
        finals.append(m.state_dict()["down4.maxpool_conv.1.double_conv.3.weight"].clone())
    assert results[0] == results[1]
    assert torch.equal(finals[0], finals[1]), "training step is not bitwise reproducible"


def test_double_conv_at_benchmark_size_against_torch():
    """DoubleConv(64, 64) -- src/model.py:13-20 -- at the benchmark's own frame (8 x 64 x 256 x 256, bf16 mode: the
    weight-stationary 64-channel kernels with in-kernel statistics, BN apply, their backward) against plain torch on
    the CPU with the same bf16 storage points (replaces a range/moments check: VERDICT r2 weak #3)."""
    import tiaozhanbei_unet_amd as P
    torch.manual_seed(0)
    dc = P.DoubleConv(64, 64, precision="bf16").to(DEV).train()
    x = (torch.randn(8, 64, 256, 256) * 0.7).bfloat16().float()
    g = torch.randn(8, 64, 256, 256).bfloat16().float()
    xd = x.to(DEV).requires_grad_(True)
    y = dc(xd)
    y.backward(g.to(DEV).to(y.dtype))
    torch.cuda.synchronize()
    state = {"b." + k: v.detach().cpu().clone() for k, v in dc.state_dict().items()}
    work = {k: (v.requires_grad_(True) if O.is_trainable(k) else v) for k, v in state.items()}
    xr = x.clone().requires_grad_(True)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with O.bf16_storage():
        yr = O.double_conv(work, "b", xr, True)
        yr.backward(g)
    # (one bf16 ulp of the largest activations, ~5 sigma after BatchNorm, is 2^-5)
    assert maxabs(y, yr) <= 1.2e-2 * max(1.0, float(yr.abs().max())), (maxabs(y, yr), float(yr.abs().max()))
    assert l2rel(xd.grad, xr.grad) < 3e-2, l2rel(xd.grad, xr.grad)
    for k in ("double_conv.0.weight", "double_conv.3.weight", "double_conv.1.weight", "double_conv.4.bias"):
        got = dict(dc.named_parameters())[k].grad
        assert l2rel(got, work["b." + k].grad) < 3e-2, (k, l2rel(got, work["b." + k].grad))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("two_streams", [False, True])
def test_skip_gradient_fan_in_matches_autograd_sums(precision, two_streams):
    """ops.GradSink (the consumers of a skip tensor add their data gradients into ONE buffer inside their
    kernels) against plain autograd accumulation (one tensor per consumer + add kernels): same parameter
    gradients, with the decoders on one and on two HIP streams."""
    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd import ops
    model, _ = make_model(("anomaly_unet", 3, 1, False), precision)
    model.train()
    model.two_streams = two_streams
    x = W.make_input("fanin:x", (2, 3, 32, 48)).to(DEV)
    tgt = W.make_input("fanin:t", (2, 3, 32, 48)).to(DEV).sigmoid()
    grads = {}
    try:
        for share in (True, False):
            ops.SHARE_SKIP_GRADS = share
            model.zero_grad(set_to_none=True)
            r, a = model(x)
            ((r - tgt) ** 2).mean().add(a.mean()).backward()
            torch.cuda.synchronize()
            grads[share] = {k: v.grad.clone() for k, v in model.named_parameters()}
    finally:
        ops.SHARE_SKIP_GRADS = True
    worst = max(l2rel(grads[True][k], grads[False][k]) for k in grads[True])
    # same addition order (the autograd engine's), same roundings: equal up to bf16 ties in the fp32 sums
    assert worst <= (1e-6 if precision == "fp32" else 2e-2), worst


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_inference_with_folded_batchnorm_matches_the_unfused_path(precision):
    """model.eval() under no_grad runs one kernel per conv layer (BatchNorm folded into weights + epilogue bias/ReLU,
    SURVEY 8f-4); with autograd recording (or UNET_FOLD_BN=0) the separate apply pass runs.  Same results."""
    from tiaozhanbei_unet_amd import ops
    model, _ = make_model(("anomaly_unet", 3, 1, False), precision)
    model.eval()
    x = W.make_input("fold:x", (2, 3, 48, 32)).to(DEV)
    with torch.no_grad():
        r1, a1 = model(x)
    try:
        ops.FOLD_EVAL_BN = False
        with torch.no_grad():
            r0, a0 = model(x)
    finally:
        ops.FOLD_EVAL_BN = True
    tol = 2e-5 if precision == "fp32" else 3e-2
    assert maxabs(r1, r0) <= tol and maxabs(a1, a0) <= tol, (maxabs(r1, r0), maxabs(a1, a0))
    if precision == "fp32":
        ref_r, ref_a = O.anomaly_unet_forward({k: v.cpu() for k, v in model.state_dict().items()}, x.cpu(), training=False)[:2]
        assert maxabs(r1, ref_r) <= 1e-3 and maxabs(a1, ref_a) <= 1e-3
        assert torch.equal(a1.cpu() > 0.5, ref_a > 0.5)


@pytest.mark.parametrize("model", ["anomaly_unet", "unet"])
def test_cli_train_then_test_roundtrip(tmp_path, model):
    """BASELINE configs[0] as named (128x128, bs=4, 2 epochs) through the train/test CLIs on an MVTec-layout toy tree:
    output tree, args.json, checkpoint dict keys and training_results.json keys of the reference."""
    import json
    import os
    from tiaozhanbei_unet_amd import test as test_cli
    from tiaozhanbei_unet_amd import train as train_cli
    from tiaozhanbei_unet_amd.dataset import write_synthetic_mvtec
    root = write_synthetic_mvtec(str(tmp_path / "data"), "bottle", n_train=8, n_good=3, n_bad=3, size=128)
    exp = train_cli.main(["--data_root", root, "--category", "bottle", "--model", model, "--epochs", "2",
                          "--batch_size", "4", "--image_size", "128", "--num_workers", "0", "--val_freq", "1",
                          "--save_freq", "1", "--save_dir", str(tmp_path / "out")])
    for d in ("checkpoints", "results", "visualizations", "logs"):
        assert os.path.isdir(os.path.join(exp, d))
    res = json.load(open(os.path.join(exp, "results", "training_results.json")))
    assert set(res) == {"train_losses", "val_losses", "best_val_loss", "total_epochs", "total_params", "args"}
    assert len(res["train_losses"]) == 2 and all(l == l for l in res["train_losses"])
    assert res["total_params"] == (43228228 if model == "anomaly_unet" else 31037633)
    ck = os.path.join(exp, "checkpoints", "checkpoint_epoch_1.pth")
    sd = torch.load(ck, map_location="cpu", weights_only=True)
    assert set(sd) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    assert "inc.double_conv.0.weight" in sd["model_state_dict"]
    out = test_cli.main(["--data_root", root, "--category", "bottle", "--model", model, "--checkpoint", ck,
                         "--batch_size", "4", "--image_size", "128", "--num_workers", "0",
                         "--output_dir", str(tmp_path / "test_out")])
    tm = json.load(open(os.path.join(out, "test_metrics.json")))
    assert {"image_metrics", "pixel_metrics", "threshold", "args"} <= set(tm)


def test_bench_two_ranks_rehearsal(tmp_path):
    """The multi-rank path of bench.py (torch.distributed.run, DataParallel buckets, barrier/max timing, one JSON
    line from rank 0) rehearsed with 2 ranks stacked on this one card over gloo (RCCL needs one GPU per rank;
    the driver runs the real N=2/4/8 case)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, UNET_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "2", "--size", "64"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 4
    assert rec["scaling"] == "weak" and rec["value"] > 0 and "cpu_baseline" not in rec


def test_nonsquare_config_shapes():
    """BASELINE configs[3]/[4] shapes.  A scaled-down 176x64 non-square crop (H != W, both divisible by 16) and the
    512x512 frame of configs[3] are checked against the CPU oracle in fp32; bf16 at 512x512 against the fp32 mode."""
    import tiaozhanbei_unet_amd as P
    state = W.make_state(W.state_spec("anomaly_unet", 3, 1, False), 0)
    x = W.make_input("ns:x", (1, 3, 176, 64))
    m, _ = make_model(("anomaly_unet", 3, 1, False), "fp32")
    m.train()
    with torch.no_grad():
        r, a = m(x.to(DEV))
        r_ref, a_ref = O.anomaly_unet_forward(state, x, True)
    assert maxabs(r, r_ref) < 1e-3 and maxabs(a, a_ref) < 1e-3
    # configs[3] geometry (3 x 512 x 512), N = 1: fp32 train-mode forward against the oracle; then the bf16 mode against
    # the (oracle-checked) fp32 mode of the same library on the same input, forward and loss within the documented bf16
    # bounds, and one gradient per kernel family (a range / finiteness check stood here before: VERDICT r2 weak #3).
    # The 1408 x 512 crop (configs[4]) has its oracle test in test_gpu_round2.py.
    x5 = W.make_input("ns:x512", (1, 3, 512, 512))
    mask5 = W.make_input("ns:m512", (1, 1, 512, 512), kind="bernoulli")
    outs = {}
    for precision in ("fp32", "bf16"):
        m, _ = make_model(("anomaly_unet", 3, 1, False), precision)
        m.train()
        recon, amap = m(x5.to(DEV))
        loss = P.CombinedLoss()(recon, amap, x5.to(DEV), mask5.to(DEV))["total_loss"]
        loss.backward()
        outs[precision] = (recon.detach().float().cpu(), amap.detach().float().cpu(), float(loss),
                           {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()
                            if k.startswith(("outc_", "up4_recon.conv.double_conv.3", "up4_seg.conv.double_conv.3"))})
        del m, recon, amap, loss
        torch.cuda.empty_cache()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    with torch.no_grad():
        r_ref, a_ref = O.anomaly_unet_forward(state, x5, True)
    assert maxabs(outs["fp32"][0], r_ref) < 1e-3 and maxabs(outs["fp32"][1], a_ref) < 1e-3
    assert maxabs(outs["bf16"][0], outs["fp32"][0]) < 3e-2 and maxabs(outs["bf16"][1], outs["fp32"][1]) < 3e-2
    assert abs(outs["bf16"][2] - outs["fp32"][2]) < 2e-3
    for k, gref in outs["fp32"][3].items():
        assert l2rel(outs["bf16"][3][k], gref) < 8e-2, (k, l2rel(outs["bf16"][3][k], gref))
