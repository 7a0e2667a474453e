"""GPU parity tests added in round 4 (VERDICT r3 "next" item 1): oracle-checked TRAINING steps at the non-headline
geometries of BASELINE.json (configs[3] 512 x 512, configs[4] 1408 x 512), and layer-isolated bf16 checks of every
encoder / decoder block at the benchmark's own shapes, so that the deep bf16 backward kernels are pinned by something
tighter than the model-level bound."""
import os

import pytest
import torch

from oracle import unet_oracle as O
from oracle import weights as W
from test_gpu_model import DEV, l2rel, make_model, maxabs

pytestmark = pytest.mark.gpu


def _host_threads():
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


# ------------------------------------------------------------------ training steps at configs[3] / configs[4] geometry
@pytest.mark.parametrize("hw", [(1408, 512), (512, 512)], ids=["kolektor_1408x512", "synthetic_512x512"])
def test_training_step_at_large_geometries_against_oracle(hw):
    """AnomalyUNet (src/model.py:161-210), N = 1, 3 x H x W, train mode, fp32: forward + CombinedLoss
    (src/train_utils.py:13-44) + backward against oracle.anomaly_unet_forward / combined_loss -- both outputs within
    1e-3, the loss within 1e-4, EVERY parameter gradient L2-relative within 3e-2 (the bound of the 256 x 256 step,
    tests/test_gpu_round2.py; fp32 summation-order noise amplified through 26 BatchNorm layers at N = 1).  The
    1408 x 512 frame is non-square and 88 x 32 at the bottleneck: persistent-kernel work lists with more pixel tiles
    than CUs in one image, the two-view decoders and the weight-gradient split at K = 720 896 pixels."""
    import tiaozhanbei_unet_amd as P
    h, w = hw
    m, state = make_model(("anomaly_unet", 3, 1, False), "fp32")
    m.train()
    image = W.make_input(f"r4:image{h}x{w}", (1, 3, h, w))
    mask = W.make_input(f"r4:mask{h}x{w}", (1, 1, h, w), kind="bernoulli")
    recon, amap = m(image.to(DEV))
    d = P.CombinedLoss()(recon, amap, image.to(DEV), mask.to(DEV))
    d["total_loss"].backward()
    torch.cuda.synchronize()
    _host_threads()
    work = {k: (v.clone().requires_grad_(True) if O.is_trainable(k) else v) for k, v in state.items()}
    r_ref, a_ref = O.anomaly_unet_forward(work, image, True)
    l_ref = O.combined_loss(r_ref, a_ref, image, mask)
    l_ref["total_loss"].backward()
    assert maxabs(recon, r_ref) < 1e-3 and maxabs(amap, a_ref) < 1e-3, (maxabs(recon, r_ref), maxabs(amap, a_ref))
    for k in ("total_loss", "recon_loss", "seg_loss"):
        assert abs(float(d[k].detach()) - float(l_ref[k].detach())) < 1e-4, (k, float(d[k].detach()), float(l_ref[k].detach()))
    clear = (a_ref - 0.5).abs() > 2e-4
    assert torch.equal((amap.cpu() > 0.5)[clear], (a_ref > 0.5)[clear])
    errs = {k: l2rel(p.grad, work[k].grad) for k, p in m.named_parameters()}
    assert len(errs) == sum(1 for k in state if O.is_trainable(k))
    worst = max(errs, key=errs.get)
    median = sorted(errs.values())[len(errs) // 2]
    print(f"[{h}x{w}] worst gradient L2-rel {errs[worst]:.3e} ({worst}), median {median:.3e}")
    assert errs[worst] < 3e-2, f"{worst}: L2-relative gradient error {errs[worst]:.3e}; median {median:.3e}"
    assert median < 5e-3, f"median L2-relative gradient error {median:.3e}"


# ------------------------------------------------------------------ every block at its benchmark shape, bf16, N = 8
# (kind, channels in, channels out, input frame of the block) -- SURVEY 2.4 K1: each distinct (Cin, Cout, H) of the
# AnomalyUNet step at 256 x 256.  Down blocks take the level above (pool inside), Up blocks take the level below
# (transposed convolution inside) plus the skip: the two-source convolutions.
BENCH_BLOCKS = [
    ("down", 64, 128, 256), ("down", 128, 256, 128), ("down", 256, 512, 64), ("down", 512, 1024, 32),
    ("up", 1024, 512, 16), ("up", 512, 256, 32), ("up", 256, 128, 64), ("up", 128, 64, 128),
]


@pytest.mark.parametrize("kind,cin,cout,size", BENCH_BLOCKS, ids=[f"{b[0]}_{b[1]}_{b[2]}_at{b[3]}" for b in BENCH_BLOCKS])
def test_blocks_at_benchmark_shapes_bf16_against_bf16_storage_oracle(kind, cin, cout, size):
    """One encoder / decoder block (src/model.py:29-37 / :43-66 with DoubleConv :13-20) in bf16 mode at N = 8 and the
    benchmark's frame, seeded upstream gradient, against the CPU oracle with the same bf16 STORAGE points
    (oracle.bf16_storage): output within one bf16 ulp of its largest value, dx and EVERY parameter gradient within
    3e-2 L2-relative.  A sign or indexing error in one deep bf16 kernel moves its gradient by O(1) and cannot hide
    here the way it could inside the 0.6 model-level bound."""
    import tiaozhanbei_unet_amd as P
    n = 8
    tag = f"r4:{kind}_{cin}_{cout}"
    if kind == "down":
        m = P.Down(cin, cout, precision="bf16")
        state = W.make_state(W.block_spec("down", cin, cout), 0)
        shapes = [(n, cin, size, size)]
    else:
        m = P.Up(cin, cout, False, precision="bf16")
        state = W.make_state(W.block_spec("up", cin, cout, False), 0)
        shapes = [(n, cin, size, size), (n, cin // 2, 2 * size, 2 * size)]
    assert list(m.state_dict().keys()) == list(state.keys())
    m.load_state_dict(state)
    m = m.to(DEV).train()
    # post-ReLU-like inputs already on the bf16 grid (what the neighbouring layers hand over)
    xs = [W.make_input(f"{tag}:x{i}", s).clamp_min(-0.5).bfloat16().float() for i, s in enumerate(shapes)]
    xd = [x.to(DEV).requires_grad_(True) for x in xs]
    y = m(*xd)
    gy = W.make_input(f"{tag}:gy", tuple(y.shape)).bfloat16().float()
    y.backward(gy.to(DEV).to(y.dtype))
    torch.cuda.synchronize()
    _host_threads()
    work = {"b." + k: (v.clone().requires_grad_(True) if O.is_trainable(k) else v.clone()) for k, v in state.items()}
    xr = [x.clone().requires_grad_(True) for x in xs]
    with O.bf16_storage():
        yr = O.down(work, "b", xr[0], True) if kind == "down" else O.up(work, "b", xr[0], xr[1], True, False)
        yr.backward(gy)
    assert maxabs(y, yr) <= 1.2e-2 * max(1.0, float(yr.abs().max())), (maxabs(y, yr), float(yr.abs().max()))
    report = {}
    for i in range(len(xs)):
        report[f"dx{i}"] = l2rel(xd[i].grad, xr[i].grad)
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        report[k] = l2rel(p.grad, work["b." + k].grad)
    worst = max(report, key=report.get)
    print(f"[{tag} @{size}] worst {worst} = {report[worst]:.3e}")
    assert report[worst] < 3e-2, ", ".join(f"{k}={v:.3e}" for k, v in sorted(report.items(), key=lambda kv: -kv[1])[:5])


# ------------------------------------------------------------------ BatchNorm-backward apply folded into the image layer's wgrad
@pytest.mark.parametrize("shape", [(4, 3, 48, 64), (2, 3, 256, 256), (3, 1, 40, 32)], ids=["48x64", "256x256", "grey_40x32"])
def test_first_layer_wgrad_with_folded_bn_backward_is_bit_identical(shape):
    """inc.double_conv.0..2 (src/model.py:14-16) in bf16 mode: the image layer's dy = A*dz + B*y + K has one consumer, its
    weight gradient, so unet_conv3x3_first_wgrad_bn forms it on the operand instead of a standalone pass.  Same values,
    same rounding: every gradient of the block is bit-identical to the unfused path (UNET_FUSE_FIRST_BN=0), and both agree
    with the oracle."""
    import tiaozhanbei_unet_amd as P
    from tiaozhanbei_unet_amd import ops
    n, c, h, w = shape
    state = W.make_state(W.block_spec("double_conv", c, 64), 0)
    x = W.make_input(f"r4:first{h}x{w}", shape)
    gy = W.make_input(f"r4:first_gy{h}x{w}", (n, 64, h, w)).bfloat16().float()
    grads = {}
    for fused in (True, False):
        ops.FUSE_FIRST_BN_BWD = fused
        try:
            m = P.DoubleConv(c, 64, precision="bf16")
            m.load_state_dict(state)
            m = m.to(DEV).train()
            assert ops.first_layer_ok(x.to(DEV), m.double_conv[0], torch.bfloat16)
            y = m(x.to(DEV))
            y.backward(gy.to(DEV).to(y.dtype))
            torch.cuda.synchronize()
            grads[fused] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        finally:
            ops.FUSE_FIRST_BN_BWD = True
    for k in grads[True]:
        assert torch.equal(grads[True][k], grads[False][k]), k
    _host_threads()
    work = {"b." + k: (v.clone().requires_grad_(True) if O.is_trainable(k) else v.clone()) for k, v in state.items()}
    with O.bf16_storage():
        yr = O.double_conv(work, "b", O._qw(x), True)
        yr.backward(gy)
    for k, g in grads[True].items():
        assert l2rel(g, work["b." + k].grad) < 3e-2, (k, l2rel(g, work["b." + k].grad))
