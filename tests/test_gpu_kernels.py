"""GPU parity tests, kernel level: each C-ABI entry point of libunet_hip.so against the CPU oracle /
plain fp32 torch on the same seeded inputs.  fp32 kernels are held to tight bounds (the parity mode);
bf16 kernels are compared on bf16-rounded inputs with a bound set by bf16 output rounding (2^-8)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from oracle import unet_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
IDS = ["fp32", "bf16"]


@pytest.fixture(scope="module")
def hip():
    from tiaozhanbei_unet_amd import _lib, ops
    return _lib, ops


def dev():
    return torch.device("cuda:0")


def rnd(name, shape, kind="normal"):
    return W.make_input("k:" + name, shape, kind=kind)


def q(t, dtype):
    """value the kernel actually sees (bf16 rounding of inputs)"""
    return t.detach().to(dtype).float().clone()


def nhwc(t, dtype):
    return t.to(dev()).to(dtype).contiguous(memory_format=torch.channels_last)


def tol(dtype, ref, f32=2e-5, bf=1.2e-2):
    return (f32 if dtype == torch.float32 else bf) * max(1.0, float(ref.abs().max()))


def check(out, ref, dtype, what, f32=2e-5, bf=1.2e-2):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    err = float((out - ref).abs().max())
    bound = tol(dtype, ref, f32, bf)
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e} (|ref|max {float(ref.abs().max()):.3e})"


def views(L, items):
    arr = L.View2()
    for i, it in enumerate(items):
        arr[i] = L.View(None, 0, 0, 0, 0, 0) if it is None else L.View(it[0].data_ptr(), it[0].shape[1],
                                                                        it[0].shape[2], it[0].shape[3], it[1], it[2])
    return arr


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    return C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------ conv3x3 forward / dgrad / wgrad
CONV_CASES = [  # n, cin, cout, h, w
    (2, 64, 64, 16, 16), (1, 64, 128, 9, 21), (2, 128, 64, 8, 16), (1, 256, 128, 5, 3), (1, 64, 64, 40, 33),
    # 16-aligned frames with >= 128 input channels: the persistent LDS-DMA kernels (128- and 64-channel blocks,
    # several work items per block, dgrad with 128 / 256 rows)
    (2, 128, 128, 16, 32), (1, 256, 64, 32, 16), (3, 128, 64, 48, 32), (1, 512, 256, 16, 16)]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3x3_fwd_dgrad_wgrad(hip, dtype, case):
    L, ops = hip
    n, ci, co, h, w = case
    x = rnd(f"cx{case}", (n, ci, h, w))
    wt = rnd(f"cw{case}", (co, ci, 3, 3)) * (1.0 / (3 * ci ** 0.5))
    gy = rnd(f"cg{case}", (n, co, h, w))
    xq, wq, gq = (q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), q(gy, dtype))
    ref = F.conv2d(xq, wq, padding=1)
    ref.backward(gq)
    dt = ops._DT[dtype]
    xd, gd = nhwc(x, dtype), nhwc(gy, dtype)
    wd = wt.to(dev())
    # forward
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wd, L.PACK_CONV_FWD, co, ci, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(xd, 0, 0), None]), p(wp), co, views(L, [(y, 0, 0), None]),
                                 co, 0, L.K_CONV_FWD, st()), "conv fwd")
    check(y, ref, dtype, "conv3x3 fwd")
    # data gradient (same kernel, flipped weights)
    dx = ops._nhwc_empty(n, ci, h, w, dtype, dev())
    wpd = ops.pack_weight(wd, L.PACK_CONV_DGRAD, ci, co, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), ci, views(L, [(dx, 0, 0), None]),
                                 ci, 0, L.K_CONV_DGRAD, st()), "conv dgrad")
    check(dx, xq.grad, dtype, "conv3x3 dgrad")
    # accumulate flag: dst += result
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), ci, views(L, [(dx, 0, 0), None]),
                                 ci, 1, L.K_CONV_DGRAD, st()), "conv dgrad acc")
    check(dx, 2 * xq.grad, dtype, "conv3x3 dgrad accumulate", bf=2.5e-2)
    # weight gradient
    dw = torch.empty(co, ci, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_wgrad(dt, n, h, w, views(L, [(xd, 0, 0), None]), p(gd), co, p(dw), ci, p(ws), need,
                                       st()), "conv wgrad")
    check(dw, wq.grad, dtype, "conv3x3 wgrad", f32=5e-5, bf=5e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_conv3x3_image_layer_padded_channels(hip, dtype):
    """3 image channels zero-padded to 64 by unet_nchw_to_nhwc; weight grad returns only the 3 real ones."""
    L, ops = hip
    n, ci, co, h, w = 2, 3, 64, 12, 20
    x, wt, gy = rnd("ix", (n, ci, h, w)), rnd("iw", (co, ci, 3, 3)) * 0.2, rnd("ig", (n, co, h, w))
    xq, wq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True)
    ref = F.conv2d(xq, wq, padding=1)
    ref.backward(q(gy, dtype))
    xd = ops.PackInput.apply(x.to(dev()), dtype)
    assert xd.shape == (n, 64, h, w)
    back = torch.empty(n, ci, h, w, device=dev())
    L.check(L.lib().unet_nhwc_to_nchw(p(xd), p(back), n, ci, h, w, 64, ops._DT[dtype], st()), "unpack")
    check(back, q(x, dtype), dtype, "pack/unpack round trip", f32=0, bf=0)
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wt.to(dev()), L.PACK_CONV_FWD, co, 64, dtype)
    L.check(L.lib().unet_conv3x3(ops._DT[dtype], n, h, w, views(L, [(xd, 0, 0), None]), p(wp), co,
                                 views(L, [(y, 0, 0), None]), co, 0, 0, st()), "conv fwd")
    check(y, ref, dtype, "image-layer conv")
    gd = nhwc(gy, dtype)
    dw = torch.empty(co, ci, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_wgrad_workspace(n, h, w, 64, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_wgrad(ops._DT[dtype], n, h, w, views(L, [(xd, 0, 0), None]), p(gd), co, p(dw), ci,
                                       p(ws), need, st()), "wgrad")
    check(dw, wq.grad, dtype, "image-layer wgrad", f32=5e-5, bf=5e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 64, 64, 16, 16), (1, 64, 128, 9, 21), (2, 128, 128, 16, 32), (1, 256, 64, 32, 16),
                                  (1, 128, 64, 5, 7)], ids=str)
def test_conv3x3_folded_batchnorm_inference(hip, dtype, case):
    """unet_pack_conv_weight_folded + unet_conv3x3_bias_relu (inference: BatchNorm(eval) folded into the layer, shift and
    ReLU in the conv epilogue) against relu(batch_norm_eval(conv(x))) -- every conv kernel family (weight-stationary,
    persistent LDS-DMA with 128- and 64-channel blocks, register-staged fallback)."""
    L, ops = hip
    n, ci, co, h, w = case
    x = rnd(f"bx{case}", (n, ci, h, w))
    wt = rnd(f"bw{case}", (co, ci, 3, 3)) * (1.0 / (3 * ci ** 0.5))
    gamma, beta = rnd(f"bg{case}", (co,)) * 0.5 + 1.0, rnd(f"bb{case}", (co,)) * 0.3
    rm, rv = rnd(f"bm{case}", (co,)) * 0.2, rnd(f"bv{case}", (co,), kind="uniform") + 0.5
    scale = gamma / torch.sqrt(rv + 1e-5)
    shift = beta - rm * scale
    ref = torch.relu(F.conv2d(q(x, dtype), q(wt * scale[:, None, None, None], dtype), padding=1) + shift[None, :, None, None])
    dt = ops._DT[dtype]
    ss = torch.empty(2, co, device=dev())
    gd, bd, md, vd, wd, xd = (t.to(dev()) for t in (gamma, beta, rm, rv, wt, x))      # keep the device copies alive
    xn = nhwc(xd, dtype)
    L.check(L.lib().unet_bn_eval_coeffs(co, p(gd), p(bd), p(md), p(vd), C.c_float(1e-5), p(ss[0]), p(ss[1]), st()), "coeffs")
    wq = torch.empty(9 * co * ci, dtype=dtype, device=dev())
    L.check(L.lib().unet_pack_conv_weight_folded(p(wd), p(ss[0]), p(wq), co, ci, co, ci, dt, st()), "fold")
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    L.check(L.lib().unet_conv3x3_bias_relu(dt, n, h, w, views(L, [(xn, 0, 0), None]), p(wq), co, p(y), p(ss[1]), 1, st()),
            "conv bias relu")
    check(y, ref, dtype, "folded conv + BN(eval) + ReLU", f32=5e-5, bf=2e-2)
    assert float(y.float().min()) >= 0.0


@pytest.mark.parametrize("case", [(2, 3, 12, 32), (1, 1, 16, 16), (3, 3, 33, 48), (2, 2, 5, 64)], ids=str)
def test_conv3x3_first_layer_kernels(hip, case):
    """unet_conv3x3_first_stats / _first_wgrad (the image layer without the 64-channel padded copy, bf16 mode)
    against F.conv2d on the bf16-rounded operands: output, fused BatchNorm partial sums, weight gradient."""
    L, ops = hip
    n, ci, h, w = case
    co, dtype = 64, torch.bfloat16
    assert L.lib().unet_conv3x3_first_supported(ci, co, h, w) == 1
    assert L.lib().unet_conv3x3_first_supported(4, co, h, w) == 0 and L.lib().unet_conv3x3_first_supported(ci, co, h, 24) == 0
    x, wt, gy = rnd(f"fx{case}", (n, ci, h, w)), rnd(f"fw{case}", (co, ci, 3, 3)) * 0.3, rnd(f"fg{case}", (n, co, h, w))
    xq, wq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True)
    ref = F.conv2d(xq, wq, padding=1)
    ref.backward(q(gy, dtype))
    xd, wd = x.to(dev()).contiguous(), wt.to(dev()).contiguous()
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    cap = L.lib().unet_conv3x3_stats_max_parts(n, h, w)
    part = torch.zeros(cap * 2 * co, device=dev())
    nparts = C.c_int32(0)
    L.check(L.lib().unet_conv3x3_first_stats(n, h, w, p(xd), ci, p(wd), p(y), p(part), C.byref(nparts), st()), "first fwd")
    check(y, ref, dtype, "first-layer conv fwd")
    assert 0 < nparts.value <= cap
    sums = part[:nparts.value * 2 * co].view(nparts.value, 2, co).double().sum(0).cpu()
    yf = y.float().cpu().double()
    assert torch.allclose(sums[0], yf.sum((0, 2, 3)), rtol=1e-4, atol=1e-2), "fused sum(y)"
    assert torch.allclose(sums[1], (yf * yf).sum((0, 2, 3)), rtol=1e-4, atol=1e-2), "fused sum(y^2)"
    # no statistics requested (eval mode): same output
    y2 = ops._nhwc_empty(n, co, h, w, dtype, dev())
    L.check(L.lib().unet_conv3x3_first_stats(n, h, w, p(xd), ci, p(wd), p(y2), None, None, st()), "first fwd eval")
    assert torch.equal(y2, y)
    gd = nhwc(gy, dtype)
    dw = torch.empty(co, ci, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_first_wgrad_workspace(n, h, w)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_first_wgrad(n, h, w, p(xd), ci, p(gd), p(dw), p(ws), need, st()), "first wgrad")
    check(dw, wq.grad, dtype, "first-layer wgrad", bf=5e-3)
    dw2 = torch.empty_like(dw)
    L.check(L.lib().unet_conv3x3_first_wgrad(n, h, w, p(xd), ci, p(gd), p(dw2), p(ws), need, st()), "first wgrad again")
    assert torch.equal(dw, dw2), "ordered reductions: bitwise reproducible"


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("geom", [(1, 64, 64, 64, 17, 19, 16, 16), (2, 64, 64, 64, 32, 16, 16, 16),
                                  (1, 128, 128, 128, 16, 32, 16, 32),
                                  # dense 16-aligned frames, 64 gradient channels: the two-destination (and gradient
                                  # fan-in) forms of conv3_ws16_kernel, several tiles per block
                                  (2, 64, 64, 64, 32, 48, 32, 48), (9, 64, 64, 64, 64, 64, 64, 64),
                                  (24, 64, 64, 64, 64, 64, 64, 64)], ids=str)      # 768 tiles x 2 channel groups: six per block
def test_conv3x3_concat_and_centre_pad_views(hip, dtype, geom):
    """cat([x2, pad(x1)]) (model.py:57-65) expressed as two source views, and the matching
    two-destination data gradient (odd frame: register-staged kernels; 16-aligned frames: LDS-DMA kernels)."""
    L, ops = hip
    n, c0, c1, co, h, w, h1, w1 = geom
    oy, ox = (h - h1) // 2, (w - w1) // 2
    x2, x1 = rnd("vx2", (n, c0, h, w)), rnd("vx1", (n, c1, h1, w1))
    wt = rnd("vw", (co, c0 + c1, 3, 3)) * 0.05
    gy = rnd("vg", (n, co, h, w))
    x2q, x1q = q(x2, dtype).requires_grad_(True), q(x1, dtype).requires_grad_(True)
    wq = q(wt, dtype).requires_grad_(True)
    x1p = F.pad(x1q, [ox, w - w1 - ox, oy, h - h1 - oy])
    ref = F.conv2d(torch.cat([x2q, x1p], 1), wq, padding=1)
    ref.backward(q(gy, dtype))
    dt = ops._DT[dtype]
    x2d, x1d, gd = nhwc(x2, dtype), nhwc(x1, dtype), nhwc(gy, dtype)
    src = views(L, [(x2d, 0, 0), (x1d, oy, ox)])
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wt.to(dev()), L.PACK_CONV_FWD, co, c0 + c1, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, src, p(wp), co, views(L, [(y, 0, 0), None]), co, 0, 0, st()), "fwd")
    check(y, ref, dtype, "two-source conv")
    d2 = ops._nhwc_empty(n, c0, h, w, dtype, dev())
    d1 = ops._nhwc_empty(n, c1, h1, w1, dtype, dev())
    wpd = ops.pack_weight(wt.to(dev()), L.PACK_CONV_DGRAD, c0 + c1, co, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), c0 + c1,
                                 views(L, [(d2, 0, 0), (d1, oy, ox)]), c0, 0, 1, st()), "dgrad")
    check(d2, x2q.grad, dtype, "skip gradient")
    check(d1, x1q.grad, dtype, "up-sampled gradient (cropped by the pad)")
    # accumulate is a bit mask over the two destination views: 1 = only the skip half adds into its buffer
    # (the in-place gradient fan-in of ops.GradSink), 2 = only the second view
    for mask, k2, k1 in ((1, 2, 1), (2, 1, 2), (3, 2, 3)):
        L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), c0 + c1,
                                     views(L, [(d2, 0, 0), (d1, oy, ox)]), c0, mask, 1, st()), "dgrad acc")
        check(d2, k2 * x2q.grad, dtype, f"skip gradient after accumulate={mask}", f32=2e-4, bf=4e-2)
        check(d1, k1 * x1q.grad, dtype, f"up gradient after accumulate={mask}", f32=2e-4, bf=4e-2)
    dw = torch.empty(co, c0 + c1, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_wgrad_workspace(n, h, w, c0 + c1, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_wgrad(dt, n, h, w, src, p(gd), co, p(dw), c0 + c1, p(ws), need, st()), "wgrad")
    check(dw, wq.grad, dtype, "two-source wgrad", f32=5e-5, bf=5e-3)


# ------------------------------------------------------------------ transposed conv
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 128, 64, 8, 16), (1, 256, 128, 5, 3), (1, 128, 64, 9, 20),
                                  # widths that are multiples of 32: the streaming weight-gradient kernels
                                  (2, 128, 64, 8, 32), (1, 256, 128, 4, 64), (3, 128, 64, 3, 32), (2, 256, 128, 16, 32),
                                  # the deep levels (bf16: convt_gemm_kernel, 256 x 256 tiles; ragged and several pixel tiles)
                                  (2, 512, 256, 8, 8), (3, 1024, 512, 5, 7), (4, 512, 256, 16, 16),
                                  # ... and their streaming weight gradient (width 16: a tile = two image rows; width 32)
                                  (2, 1024, 512, 16, 16), (1, 512, 256, 4, 32)], ids=str)
def test_convt2x2(hip, dtype, case):
    L, ops = hip
    n, ci, co, h, w = case
    x, wt, b = rnd(f"tx{case}", (n, ci, h, w)), rnd(f"tw{case}", (ci, co, 2, 2)) * 0.1, rnd(f"tb{case}", (co,))
    gy = rnd(f"tg{case}", (n, co, 2 * h, 2 * w))
    xq, wq, bq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv_transpose2d(xq, wq, bq, stride=2)
    check(O.conv_transpose2x2(xq, wq, bq), ref, torch.float32, "oracle convT restatement")
    ref.backward(q(gy, dtype))
    xd = nhwc(x, dtype).requires_grad_(True)
    wd, bd = wt.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    y = ops.ConvT2x2.apply(xd, wd, bd)
    check(y, ref, dtype, "convT fwd")
    y.backward(nhwc(gy, dtype))
    check(xd.grad, xq.grad, dtype, "convT dgrad")
    check(wd.grad, wq.grad, dtype, "convT wgrad", f32=5e-5, bf=5e-3)
    check(bd.grad, bq.grad, dtype, "convT bias grad", f32=5e-5, bf=5e-3)


# ------------------------------------------------------------------ BatchNorm + ReLU
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 64, 12, 20), (3, 128, 7, 5), (1, 64, 64, 64)], ids=str)
def test_bn_relu_fwd_bwd(hip, dtype, case):
    L, ops = hip
    n, c, h, w = case
    y = rnd(f"by{case}", (n, c, h, w)) * 1.7 + 0.3
    ga, be = rnd(f"bg{case}", (c,), "uniform") + 0.5, rnd(f"bb{case}", (c,)) * 0.3
    rm, rv = rnd(f"brm{case}", (c,)) * 0.1, rnd(f"brv{case}", (c,), "uniform") + 0.5
    da = rnd(f"bd{case}", (n, c, h, w))
    yq = q(y, dtype).requires_grad_(True)
    gq, bq = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
    state = {"b.weight": gq, "b.bias": bq, "b.running_mean": rm, "b.running_var": rv,
             "b.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    new = {}
    ref = torch.clamp_min(O.batch_norm(state, "b", yq, True, new), 0)
    ref.backward(q(da, dtype))
    dt, pixels = ops._DT[dtype], n * h * w
    yd, dad = nhwc(y, dtype), nhwc(da, dtype)
    gd, bd, rmd, rvd = ga.to(dev()), be.to(dev()), rm.to(dev()).clone(), rv.to(dev()).clone()
    coef = torch.empty(4, c, device=dev())
    need = L.lib().unet_bn_workspace(pixels, c)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_bn_train_stats(dt, p(yd), pixels, c, p(gd), p(bd), p(rmd), p(rvd), 0.1, 1e-5, p(coef[0]),
                                        p(coef[1]), p(coef[2]), p(coef[3]), p(ws), need, st()), "bn stats")
    a = torch.empty_like(yd)
    L.check(L.lib().unet_bn_relu_apply(dt, p(yd), pixels, c, p(coef[2]), p(coef[3]), p(a), st()), "bn apply")
    check(coef[0], yq.detach().mean((0, 2, 3)), torch.float32, "batch mean", f32=1e-5)
    check(rmd, new["b.running_mean"], torch.float32, "running_mean", f32=1e-5)
    check(rvd, new["b.running_var"], torch.float32, "running_var (unbiased)", f32=1e-5)
    check(a, ref, dtype, "bn+relu fwd", f32=1e-5)
    dy = torch.empty_like(yd)
    dgb = torch.empty(2, c, device=dev())
    L.check(L.lib().unet_bn_relu_bwd(dt, p(dad), p(yd), pixels, c, p(gd), p(coef[0]), p(coef[1]), p(coef[2]),
                                     p(coef[3]), p(dgb[0]), p(dgb[1]), p(dy), p(ws), need, st()), "bn bwd")
    check(dgb[0], gq.grad, torch.float32, "dgamma", f32=2e-5)
    check(dgb[1], bq.grad, torch.float32, "dbeta", f32=2e-5)
    check(dy, yq.grad, dtype, "bn+relu dgrad", f32=2e-5)
    # eval coefficients
    L.check(L.lib().unet_bn_eval_coeffs(c, p(gd), p(bd), p(rmd), p(rvd), 1e-5, p(coef[2]), p(coef[3]), st()), "eval")
    L.check(L.lib().unet_bn_relu_apply(dt, p(yd), pixels, c, p(coef[2]), p(coef[3]), p(a), st()), "bn apply")
    st2 = dict(state); st2.update(new)
    check(a, torch.clamp_min(O.batch_norm(st2, "b", yq.detach(), False), 0), dtype, "bn eval", f32=1e-5)


# ------------------------------------------------------------------ max-pool / bilinear
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 64, 13, 10), (1, 128, 8, 8), (1, 64, 5, 7)], ids=str)
def test_maxpool(hip, dtype, case):
    L, ops = hip
    n, c, h, w = case
    x = rnd(f"px{case}", (n, c, h, w))
    x[:, :, 0:2, 0:2] = 1.0                       # tie: the gradient must go to the FIRST maximum
    x[:, :, 2, 2], x[:, :, 2, 3], x[:, :, 3, 2], x[:, :, 3, 3] = 1.0, 2.0, 2.0, 1.0
    xq = q(x, dtype).requires_grad_(True)
    ref = F.max_pool2d(xq, 2)
    gy = rnd(f"pg{case}", tuple(ref.shape))
    ref.backward(q(gy, dtype))
    xd = nhwc(x, dtype).requires_grad_(True)
    y = ops.MaxPool2.apply(xd)
    check(y, ref, dtype, "maxpool fwd", f32=0, bf=0)
    y.backward(nhwc(gy, dtype))
    check(xd.grad, xq.grad, dtype, "maxpool bwd (first-max ties, floor)", f32=0, bf=0)
    # accumulate: dx += routed gradient (third consumer of a skip tensor)
    base = rnd(f"pb{case}", (n, c, h, w))
    dx = nhwc(base, dtype).clone(memory_format=torch.channels_last)
    L.check(L.lib().unet_maxpool2_bwd(ops._DT[dtype], p(xd.detach()), p(nhwc(gy, dtype)), n, h, w, c, p(dx), 1, st()),
            "maxpool bwd accumulate")
    check(dx, q(base, dtype) + xq.grad, dtype, "maxpool bwd accumulate", f32=1e-6, bf=2e-2)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(1, 64, 8, 8), (2, 64, 5, 3), (1, 128, 1, 4)], ids=str)
def test_bilinear2x(hip, dtype, case):
    L, ops = hip
    n, c, h, w = case
    x = rnd(f"ux{case}", (n, c, h, w))
    xq = q(x, dtype).requires_grad_(True)
    ref = F.interpolate(xq, scale_factor=2, mode="bilinear", align_corners=True)
    check(O.upsample_bilinear2x(xq), ref, torch.float32, "oracle bilinear restatement", f32=1e-5)
    gy = rnd(f"ug{case}", tuple(ref.shape))
    ref.backward(q(gy, dtype))
    xd = nhwc(x, dtype).requires_grad_(True)
    y = ops.Bilinear2x.apply(xd)
    check(y, ref, dtype, "bilinear fwd", f32=1e-5)
    y.backward(nhwc(gy, dtype))
    check(xd.grad, xq.grad, dtype, "bilinear bwd", f32=1e-5, bf=2e-2)


# ------------------------------------------------------------------ 1x1 head
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("co", [1, 3, 4])
@pytest.mark.parametrize("sigmoid", [False, True])
def test_head(hip, dtype, co, sigmoid):
    L, ops = hip
    g = load_golden(f"block_outconv_64_{co}")
    state = W.make_state(W.block_spec("outconv", 64, co), 0)
    x = W.make_input(f"outc_{co}:x", (2, 64, 9, 11))
    gy = W.make_input(f"outc_{co}:gy", (2, co, 9, 11))
    xq = q(x, dtype).requires_grad_(True)
    wq, bq = state["conv.weight"].clone().requires_grad_(True), state["conv.bias"].clone().requires_grad_(True)
    ref = F.conv2d(xq, wq, bq)
    if sigmoid:
        ref = torch.sigmoid(ref)
    ref.backward(gy)
    xd = nhwc(x, dtype).requires_grad_(True)
    wd, bd = state["conv.weight"].to(dev()).requires_grad_(True), state["conv.bias"].to(dev()).requires_grad_(True)
    out = ops.Head.apply(xd, wd, bd, sigmoid)
    assert out.dtype == torch.float32 and out.is_contiguous()
    check(out, ref, torch.float32, "head fwd", f32=1e-5)
    if dtype == torch.float32:
        check(out, g["prob"] if sigmoid else g["logits"], torch.float32, "head fwd vs reference golden", f32=1e-5)
        if not sigmoid:
            assert torch.equal(out.argmax(1).to(torch.uint8).cpu(), g["argmax"]), "argmax mask indices differ"
    out.backward(gy.to(dev()))
    check(xd.grad, xq.grad, dtype, "head dx", f32=1e-5)
    check(wd.grad, wq.grad, torch.float32, "head dW", f32=2e-5)
    check(bd.grad, bq.grad, torch.float32, "head db", f32=2e-5)


# ------------------------------------------------------------------ losses vs the reference's goldens
LOSS_TAGS = [f"loss_combined_{t}_{rw}_{sw}" for t in ("binary", "over255", "zeros")
             for rw, sw in ((1.0, 1.0), (0.3, 2.5))]


@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_combined_loss_golden(hip, tag):
    from tiaozhanbei_unet_amd import CombinedLoss
    g = load_golden(tag)
    rw, sw = map(float, tag.split("_")[-2:])
    shape = (2, 3, 12, 10)
    recon = W.make_input("loss:recon", shape, kind="uniform").to(dev()).requires_grad_(True)
    image = W.make_input("loss:image", shape).to(dev())
    amap = W.make_input("loss:amap", (2, 1, 12, 10), kind="uniform")
    amap.view(-1)[:6] = torch.tensor([0.0, 1.0, 1e-30, 1 - 1e-7, 0.5, 1e-45])
    amap = amap.to(dev()).requires_grad_(True)
    d = CombinedLoss(recon_weight=rw, seg_weight=sw)(recon, amap, image, g["mask"].to(dev()))
    assert set(d) == {"total_loss", "recon_loss", "seg_loss"} and d["total_loss"].dim() == 0
    for k, gk in (("total_loss", "total"), ("recon_loss", "recon"), ("seg_loss", "seg")):
        assert abs(float(d[k]) - float(g[gk])) <= 2e-6 + 2e-6 * abs(float(g[gk])), (k, float(d[k]), float(g[gk]))
    d["total_loss"].backward()
    err = (recon.grad.cpu() - g["d_recon"]).abs().max()
    assert float(err) < 1e-8 + 1e-5 * float(g["d_recon"].abs().max())
    ga, gr = amap.grad.cpu(), g["d_amap"]
    assert bool(torch.isfinite(ga).all())
    rel = ((ga - gr).abs() / (gr.abs() + 1e-7)).max()
    assert float(rel) < 2e-3, f"focal gradient rel err {float(rel):.3e}"


@pytest.mark.parametrize("c,hw", [(3, (40, 36)), (1, (20, 50)), (3, (64, 64))])
def test_ssim_golden(hip, c, hw):
    from tiaozhanbei_unet_amd import SSIMLoss
    g = load_golden(f"ssim_c{c}_{hw[0]}x{hw[1]}")
    a = W.make_input(f"ssim:a{c}", (2, c) + hw, kind="uniform").to(dev()).requires_grad_(True)
    b = W.make_input(f"ssim:b{c}", (2, c) + hw).to(dev()).requires_grad_(True)
    v = SSIMLoss()(a, b)
    assert abs(float(v) - float(g["value"])) < 5e-6
    v.backward()
    for got, want, nm in ((a.grad, g["d_img1"], "d_img1"), (b.grad, g["d_img2"], "d_img2")):
        err = float((got.cpu() - want).abs().max())
        assert err < 1e-7 + 2e-4 * float(want.abs().max()), f"{nm}: {err:.3e}"
    assert abs(float(SSIMLoss()(a.detach(), a.detach()))) < 5e-6


# ------------------------------------------------------------------ fused Adam
def test_adam_matches_torch(hip):
    L, ops = hip
    n = 4 * 1000
    p0, g1, g2 = rnd("ap", (n,)), rnd("ag1", (n,)), rnd("ag2", (n,))
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=1e-4)
    pd = p0.to(dev()).clone()
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step, g in enumerate((g1, g2), 1):
        ref.grad = g.clone()
        opt.step()
        ops.adam_step_(pd, g.to(dev()), m, v, step, 1e-3, 0.9, 0.999, 1e-8, 1e-4)
    check(pd, ref.detach(), torch.float32, "adam", f32=1e-6)
    one = torch.ones(4, device=dev())
    ops.adam_step_(one, torch.zeros(4, device=dev()), torch.zeros(4, device=dev()), torch.zeros(4, device=dev()),
                   1, 1e-3, 0.9, 0.999, 1e-8, 1e-4)
    assert abs(float(one[0]) - 0.99900007) < 1e-6      # L2-coupled decay


def test_cpu_tensors_are_refused(hip):
    from tiaozhanbei_unet_amd import DoubleConv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DoubleConv(64, 64)(torch.zeros(1, 64, 8, 8))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_batched_weight_pack_matches_single_packs(hip, dtype):
    """unet_pack_weights_batched (LDS-tiled, one launch) == unet_pack_weight for every layout, incl. zero padding."""
    L, ops = hip
    ws = [(rnd("pw0", (64, 3, 3, 3)), L.PACK_CONV_FWD, 64, 64), (rnd("pw0", (64, 3, 3, 3)), L.PACK_CONV_DGRAD, 64, 64),
          (rnd("pw1", (128, 64, 3, 3)), L.PACK_CONV_FWD, 128, 64), (rnd("pw1", (128, 64, 3, 3)), L.PACK_CONV_DGRAD, 64, 128),
          (rnd("pw2", (128, 64, 2, 2)), L.PACK_CONVT_FWD, 64, 128), (rnd("pw2", (128, 64, 2, 2)), L.PACK_CONVT_DGRAD, 128, 64),
          # bf16: forward + data-gradient layouts of one weight written from ONE tile read (paired path), several tiles per
          # block column; a forward layout with no sibling behind it and a padded one take the single-layout path
          (rnd("pw3", (256, 160, 3, 3)), L.PACK_CONV_FWD, 256, 160), (rnd("pw3", (256, 160, 3, 3)), L.PACK_CONV_DGRAD, 160, 256),
          (rnd("pw5", (64, 96, 3, 3)), L.PACK_CONV_FWD, 64, 128), (rnd("pw5", (64, 96, 3, 3)), L.PACK_CONV_DGRAD, 128, 64),
          (rnd("pw4", (64, 64, 3, 3)), L.PACK_CONV_FWD, 64, 64)]
    cache = ops.PackCache(dtype)
    dev_ws, by_shape = {}, {}
    for w, mode, rows, k in ws:                 # the layouts of one weight share ONE device tensor (as in a model)
        wd = by_shape.setdefault((tuple(w.shape), float(w.flatten()[0])), w.to(dev()))
        dev_ws[id(w)] = wd
        cache.add(wd, mode, rows, k)
    cache.refresh(force=True)
    for w, mode, rows, k in ws:
        wd = dev_ws[id(w)]
        got = cache.get(wd, mode, rows, k)
        assert got is not None
        want = ops.pack_weight(wd, mode, rows, k, dtype)
        assert torch.equal(got.float().cpu(), want.float().cpu()), f"mode {mode}"


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 64, 64, 24, 40), (2, 128, 128, 16, 24), (1, 128, 64, 9, 20), (3, 256, 128, 8, 8),
                                  (1, 64, 128, 9, 21), (4, 64, 64, 128, 144),
                                  # 640 / 1152 tiles of conv3_ws16_kernel on <= 256 blocks: three and five tiles per block -- the
                                  # steady state of its 3-slot ring, the hand-counted waits and the deferred stores
                                  (40, 64, 64, 64, 64), (18, 64, 128, 128, 64),
                                  # a ragged frame of the same class: conv3_ws_kernel (per-lane geometry), three tiles per block
                                  (6, 64, 64, 200, 136)],
                         ids=str)
def test_conv3x3_fused_bn_statistics(hip, dtype, case):
    """unet_conv3x3_stats: the conv epilogue's wavefront-reduced partial sums (weight-stationary kernel, 16x16x32
    kernel) and the streaming fallback all add up to the per-channel sum / sum of squares of the STORED y."""
    L, ops = hip
    n, ci, co, h, w = case
    x = rnd(f"sx{case}", (n, ci, h, w))
    wt = rnd(f"sw{case}", (co, ci, 3, 3)) * (1.0 / (3 * ci ** 0.5))
    xd = nhwc(x, dtype)
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wt.to(dev()), L.PACK_CONV_FWD, co, ci, dtype)
    cap = L.lib().unet_conv3x3_stats_max_parts(n, h, w)
    part = torch.full((cap, 2, co), float("nan"), device=dev())
    nparts = C.c_int32(0)
    L.check(L.lib().unet_conv3x3_stats(ops._DT[dtype], n, h, w, views(L, [(xd, 0, 0), None]), p(wp), co, p(y), p(part),
                                       C.byref(nparts), st()), "conv+stats")
    assert 0 < nparts.value <= cap
    ref = F.conv2d(q(x, dtype), q(wt, dtype), padding=1)
    check(y, ref, dtype, "conv output")
    ys = y.float()
    sums = part[:nparts.value].double().sum(0).cpu()
    assert bool(torch.isfinite(sums).all())
    want_s, want_q = ys.double().sum((0, 2, 3)).cpu(), (ys.double() ** 2).sum((0, 2, 3)).cpu()
    assert float((sums[0] - want_s).abs().max()) < 1e-3 * max(1.0, float(want_s.abs().max()))
    assert float((sums[1] - want_q).abs().max()) < 1e-3 * max(1.0, float(want_q.abs().max()))


# ------------------------------------------------------------------ BatchNorm fusions (round 2)
def _bn_coefs(y, gamma, beta, eps=1e-5):
    """mean, istd, scale, shift of training-mode BatchNorm2d over an NCHW fp32 tensor (fp64 arithmetic)."""
    yd = y.double()
    mean = yd.mean((0, 2, 3))
    var = yd.var((0, 2, 3), unbiased=False)
    istd = 1.0 / torch.sqrt(var + eps)
    scale = gamma.double() * istd
    shift = beta.double() - mean * scale
    return mean.float(), istd.float(), scale.float(), shift.float()


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("co,sigmoid", [(1, True), (3, True), (4, False)])
def test_head_fused_with_batchnorm_relu(hip, dtype, co, sigmoid):
    """unet_head_bnrelu_fwd / _bwd + unet_bn_bwd_premasked (OutConv reading the RAW conv output: BatchNorm + ReLU on
    load, ReLU mask + BatchNorm-backward sums in the head's backward) against torch autograd of
    sigmoid(conv1x1(relu(batch_norm(y)))) on the same (dtype-rounded) y -- /root/reference/src/model.py:18-19,72,201."""
    L, ops = hip
    n, ci, h, w = 3, 64, 20, 24
    y = rnd(f"hb_y{co}", (n, ci, h, w)) * 1.5 + 0.3
    gamma, beta = rnd("hb_g", (ci,)) * 0.5 + 1.0, rnd("hb_b", (ci,)) * 0.2
    wt, b = rnd(f"hb_w{co}", (co, ci, 1, 1)) * 0.2, rnd(f"hb_bias{co}", (co,)) * 0.1
    g = rnd(f"hb_go{co}", (n, co, h, w))
    yq = q(y, dtype).requires_grad_(True)
    mean, istd, scale, shift = _bn_coefs(yq.detach(), gamma, beta)
    gq, bq = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    # reference: activation rounded to the compute dtype like the stored activation of the unfused path
    z = F.batch_norm(yq, None, None, gq, bq, True, 0.0, 1e-5)
    a = torch.relu(z)
    a_q = a + (q(a, dtype) - a).detach()
    wq, bb = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out_ref = F.conv2d(a_q, wq, bb)
    if sigmoid:
        out_ref = torch.sigmoid(out_ref)
    out_ref.backward(g)
    dt = ops._DT[dtype]
    yd = nhwc(y, dtype)
    coef = torch.stack([mean, istd, scale, shift]).to(dev())
    wd, bd, gd = wt.to(dev()).contiguous(), b.to(dev()), g.to(dev()).contiguous()
    out = torch.empty(n, co, h, w, device=dev())
    L.check(L.lib().unet_head_bnrelu_fwd(dt, p(yd), n, h, w, ci, p(coef[2]), p(coef[3]), p(wd), p(bd), co, int(sigmoid),
                                         p(out), st()), "head bnrelu fwd")
    check(out, out_ref, dtype, "fused head fwd", f32=2e-5, bf=2e-3)
    dz = ops._nhwc_empty(n, ci, h, w, dtype, dev())
    dwh, dbh = torch.empty(co, ci, 1, 1, device=dev()), torch.empty(co, device=dev())
    cap = L.lib().unet_head_bnrelu_max_parts()
    part = torch.zeros(cap, 2, ci, device=dev())
    nparts = C.c_int32(0)
    need = L.lib().unet_head_bwd_workspace(n, h, w, ci, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_head_bnrelu_bwd(dt, p(yd), p(coef[2]), p(coef[3]), p(coef[0]), p(out), p(gd), n, h, w, ci, p(wd), co,
                                         int(sigmoid), p(dz), p(dwh), p(dbh), p(part), C.byref(nparts), p(ws), need, st()),
            "head bnrelu bwd")
    assert 0 < nparts.value <= cap
    check(dwh, wq.grad, dtype, "fused head dW", f32=5e-5, bf=5e-3)
    check(dbh, bb.grad, dtype, "fused head db", f32=5e-5, bf=5e-3)
    # the partial sums are sums over the dz the kernel stored
    dzf = dz.float().cpu().double()
    sums = part[:nparts.value].double().sum(0).cpu()
    assert torch.allclose(sums[0], dzf.sum((0, 2, 3)), rtol=1e-4, atol=1e-3), "sum dz"
    ref1 = (dzf * (yq.detach().double() - mean.double()[None, :, None, None])).sum((0, 2, 3))
    assert torch.allclose(sums[1], ref1, rtol=1e-4, atol=1e-3), "sum dz*(y-mean)"
    dgam, dbet = torch.empty(ci, device=dev()), torch.empty(ci, device=dev())
    gam_d = gamma.to(dev())
    ws3 = torch.empty(3 * ci * 4, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_bn_bwd_premasked(dt, p(dz), p(yd), n * h * w, ci, p(gam_d), p(coef[0]), p(coef[1]), p(part),
                                          nparts.value, p(dgam), p(dbet), p(dz), p(ws3), ws3.numel(), st()), "bn premasked")
    check(dgam, gq.grad, dtype, "dgamma", f32=1e-4, bf=2e-2)
    check(dbet, bq.grad, dtype, "dbeta", f32=1e-4, bf=2e-2)
    check(dz, yq.grad, dtype, "dy (in place)", f32=1e-4, bf=3e-2)


DGRAD_BN_CASES = [  # n, c_dy, c_dx, h, w  -- two have > 256 work items per launch (persistent loop, block-mode sums);
    # 64 -> 64 runs on the weight-stationary streaming kernel (any frame size, several tiles per block, empty tile ranges)
    (2, 128, 128, 16, 32), (1, 256, 64, 32, 16), (3, 128, 256, 16, 16), (8, 128, 128, 128, 128), (5, 128, 64, 64, 96),
    (2, 64, 64, 24, 40), (1, 64, 64, 9, 21), (4, 64, 64, 128, 144),
    (40, 64, 64, 64, 64),       # conv3_ws16_kernel<false, 2>: 640 tiles, three per block (ring / counted-wait steady state)
    (6, 64, 64, 200, 136),      # ragged frame: conv3_ws_kernel<false, 2>, 702 tiles, three per block
    # >= 512 gradient channels: the ping-pong instantiation (conv3_pp128_bnbwd_kernel) with several work items per block --
    # 320 items on 256 blocks (ragged last round, per-tile partials) and 512 items (block-mode partials, two channel tiles)
    (20, 512, 512, 32, 32), (16, 512, 256, 64, 64)]


@pytest.mark.parametrize("case", DGRAD_BN_CASES, ids=str)
def test_conv3x3_dgrad_fused_relu_mask_and_bn_sums(hip, case):
    """unet_conv3x3_dgrad_bnrelu + unet_bn_bwd_premasked against torch: dz = conv_transpose(dy) * [bn(y) > 0], the
    two per-channel sums of the stored dz, then dy_prev / dgamma / dbeta of relu(batch_norm(y)) for that upstream
    gradient (/root/reference/src/model.py:15-17 backward)."""
    L, ops = hip
    n, cy, cx, h, w = case
    dtype = torch.bfloat16
    dt = ops._DT[dtype]
    assert L.lib().unet_conv3x3_dgrad_bnrelu_supported(dt, n, h, w, cy, cx) == 1
    if cy >= 128:
        assert L.lib().unet_conv3x3_dgrad_bnrelu_supported(dt, n, h + 1, w, cy, cx) == 0    # LDS-DMA kernels: 16-aligned frames
    dy = rnd(f"db_dy{case}", (n, cy, h, w))
    wt = rnd(f"db_w{case}", (cy, cx, 3, 3)) * (1.0 / (3 * cy ** 0.5))
    y = rnd(f"db_y{case}", (n, cx, h, w)) * 1.3 + 0.2
    gamma, beta = rnd("db_g", (cx,)) * 0.5 + 1.0, rnd("db_b", (cx,)) * 0.3
    dyq, wq, yq = q(dy, dtype), q(wt, dtype), q(y, dtype)
    mean, istd, scale, shift = _bn_coefs(yq, gamma, beta)
    da_ref = F.conv_transpose2d(dyq, wq, padding=1)                    # data gradient of conv2d(x, wt, padding=1)
    on = torch.addcmul(shift[None, :, None, None], yq, scale[None, :, None, None]) > 0   # ~ fma; ties have measure 0
    dz_ref = da_ref * on
    wd = wt.to(dev())
    wpd = ops.pack_weight(wd, L.PACK_CONV_DGRAD, cx, cy, dtype)
    dyd, yd = nhwc(dy, dtype), nhwc(y, dtype)
    coef = torch.stack([mean, istd, scale, shift]).to(dev())
    dz = ops._nhwc_empty(n, cx, h, w, dtype, dev())
    cap = L.lib().unet_conv3x3_stats_max_parts(n, h, w)
    part = torch.zeros(cap, 2, cx, device=dev())
    nparts = C.c_int32(0)
    L.check(L.lib().unet_conv3x3_dgrad_bnrelu(dt, n, h, w, p(dyd), cy, p(wpd), cx, p(yd), p(coef[2]), p(coef[3]), p(coef[0]),
                                              p(dz), p(part), C.byref(nparts), st()), "dgrad bnrelu")
    assert 0 < nparts.value <= cap
    # mask decisions can differ from the reference only where |scale*y+shift| is at rounding level: compare away from 0
    zabs = torch.addcmul(shift[None, :, None, None], yq, scale[None, :, None, None]).abs()
    safe = zabs > 1e-4
    err = ((dz.float().cpu() - dz_ref) * safe).abs().max()
    assert float(err) <= 1.2e-2 * max(1.0, float(dz_ref.abs().max())), f"dz: max err {float(err):.3e}"
    assert float((~safe).float().mean()) < 1e-3
    # unfused route through the same library: identical kernels up to the epilogue -> dz must match bit for bit
    da2 = ops._nhwc_empty(n, cx, h, w, dtype, dev())
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(dyd, 0, 0), None]), p(wpd), cx, views(L, [(da2, 0, 0), None]), cx, 0,
                                 L.K_CONV_DGRAD, st()), "dgrad plain")
    on_d = torch.addcmul(coef[3][None, :, None, None], yd.float(), coef[2][None, :, None, None]) > 0
    agree = (dz.float() == da2.float() * on_d).float().mean()
    assert float(agree) > 0.9999, f"fused vs unfused dz agree on {float(agree):.6f}"
    dzf = dz.float().cpu().double()
    sums = part[:nparts.value].double().sum(0).cpu()
    assert torch.allclose(sums[0], dzf.sum((0, 2, 3)), rtol=2e-4, atol=2e-2), "sum dz"
    ref1 = (dzf * (yq.double() - mean.double()[None, :, None, None])).sum((0, 2, 3))
    assert torch.allclose(sums[1], ref1, rtol=2e-4, atol=2e-2), "sum dz*(y-mean)"
    # second half: dy_prev, dgamma, dbeta vs autograd of relu(bn(y)) fed with the kernel's own (unmasked) gradient
    yr = yq.clone().requires_grad_(True)
    gq, bq = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    torch.relu(F.batch_norm(yr, None, None, gq, bq, True, 0.0, 1e-5)).backward(da2.float().cpu())
    dgam, dbet = torch.empty(cx, device=dev()), torch.empty(cx, device=dev())
    ws3 = torch.empty(3 * cx * 4, dtype=torch.uint8, device=dev())
    out = ops._nhwc_empty(n, cx, h, w, dtype, dev())
    L.check(L.lib().unet_bn_bwd_premasked(dt, p(dz), p(yd), n * h * w, cx, p(gamma.to(dev())), p(coef[0]), p(coef[1]),
                                          p(part), nparts.value, p(dgam), p(dbet), p(out), p(ws3), ws3.numel(), st()),
            "bn premasked")
    check(dgam, gq.grad, dtype, "dgamma", bf=5e-3)
    check(dbet, bq.grad, dtype, "dbeta", bf=5e-3)
    check(out, yr.grad, dtype, "dy_prev", bf=1.5e-2)
    # determinism: ordered partials
    part2 = torch.zeros_like(part)
    dz2 = torch.empty_like(dz)
    L.check(L.lib().unet_conv3x3_dgrad_bnrelu(dt, n, h, w, p(dyd), cy, p(wpd), cx, p(yd), p(coef[2]), p(coef[3]), p(coef[0]),
                                              p(dz2), p(part2), C.byref(nparts), st()), "dgrad bnrelu again")
    assert torch.equal(dz, dz2) and torch.equal(part, part2)


BIG_CONV_CASES = [  # > 256 work items of the persistent LDS-DMA kernels (the benchmark's code path), vs F.conv2d on the CPU
    # n, cin, cout, h, w
    (8, 128, 128, 128, 128),    # pdma128: 512 items, block-mode statistics (tiles % 256 == 0)
    (5, 128, 128, 64, 96),      # pdma128: 120 tiles -> per-tile statistics... and a partial last round
    (5, 128, 64, 128, 128),     # pdma64: 320 items, per-tile statistics
    (4, 256, 64, 128, 128),     # pdma64: 256 items per channel tile, block mode
    # >= 512 input channels: the PING-PONG schedule (conv3_pp128_kernel; 46 % of the dominant kernel's launches in bench.py)
    # walking several work items per block: cross-item DMA continuation, the after-epilogue vmcnt counts and the stagger
    # barrier of the two wave groups only execute then
    (20, 512, 512, 32, 32),     # 320 items on 256 blocks: ragged last round, per-tile statistics; dgrad is ping-pong too
    (16, 512, 256, 64, 64),     # 512 items, two channel tiles, block-mode statistics (the shape class of up2.conv.0)
]


@pytest.mark.parametrize("case", BIG_CONV_CASES, ids=str)
def test_conv3x3_persistent_kernels_many_work_items(hip, case):
    """The multi-work-item path of conv3_pdma{128,64}_kernel (cross-item DMA continuation, after-epilogue vmcnt counts,
    block-mode BatchNorm partials) against F.conv2d: forward + fused statistics, data gradient with and without
    accumulate, weight gradient.  These are the shapes class bench.py runs; every smaller case has <= 18 items."""
    L, ops = hip
    n, ci, co, h, w = case
    dtype = torch.bfloat16
    dt = ops._DT[dtype]
    x = rnd(f"big_x{case}", (n, ci, h, w))
    wt = rnd(f"big_w{case}", (co, ci, 3, 3)) * (1.0 / (3 * ci ** 0.5))
    gy = rnd(f"big_g{case}", (n, co, h, w))
    xq, wq, gq = q(x, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True), q(gy, dtype)
    ref = F.conv2d(xq, wq, padding=1)
    ref.backward(gq)
    xd, gd, wd = nhwc(x, dtype), nhwc(gy, dtype), wt.to(dev())
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wd, L.PACK_CONV_FWD, co, ci, dtype)
    cap = L.lib().unet_conv3x3_stats_max_parts(n, h, w)
    part = torch.zeros(cap * 2 * co, device=dev())
    nparts = C.c_int32(0)
    L.check(L.lib().unet_conv3x3_stats(dt, n, h, w, views(L, [(xd, 0, 0), None]), p(wp), co, p(y), p(part), C.byref(nparts),
                                       st()), "conv stats")
    check(y, ref, dtype, "conv3x3 fwd (many work items)")
    sums = part[:nparts.value * 2 * co].view(nparts.value, 2, co).double().sum(0).cpu()
    yf = y.float().cpu().double()
    assert torch.allclose(sums[0], yf.sum((0, 2, 3)), rtol=2e-4, atol=5e-2), "fused sum(y)"
    assert torch.allclose(sums[1], (yf * yf).sum((0, 2, 3)), rtol=2e-4, atol=5e-2), "fused sum(y^2)"
    y_plain = ops._nhwc_empty(n, co, h, w, dtype, dev())
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(xd, 0, 0), None]), p(wp), co, views(L, [(y_plain, 0, 0), None]),
                                 co, 0, L.K_CONV_FWD, st()), "conv fwd")
    assert torch.equal(y, y_plain)
    dx = ops._nhwc_empty(n, ci, h, w, dtype, dev())
    wpd = ops.pack_weight(wd, L.PACK_CONV_DGRAD, ci, co, dtype)
    if co >= 128:     # the data gradient of a layer with >= 128 output channels runs on the persistent kernels too
        L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), ci, views(L, [(dx, 0, 0), None]),
                                     ci, 0, L.K_CONV_DGRAD, st()), "conv dgrad")
        check(dx, xq.grad, dtype, "conv3x3 dgrad (many work items)")
        L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), ci, views(L, [(dx, 0, 0), None]),
                                     ci, 1, L.K_CONV_DGRAD, st()), "conv dgrad acc")
        check(dx, 2 * xq.grad, dtype, "conv3x3 dgrad accumulate (many work items)", bf=2.5e-2)
    dw = torch.empty(co, ci, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_wgrad_workspace(n, h, w, ci, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_wgrad(dt, n, h, w, views(L, [(xd, 0, 0), None]), p(gd), co, p(dw), ci, p(ws), need, st()),
            "conv wgrad")
    check(dw, wq.grad, dtype, "conv3x3 wgrad (many work items)", bf=5e-3)


TWO_SOURCE_CASES = [  # n, c_skip, c_up, cout, h, w
    (6, 64, 64, 128, 112, 128),     # lock-step kernel, 336 items
    (20, 512, 512, 512, 32, 32),    # ping-pong kernel (1024 input channels = up1.conv.0), 320 items on 256 blocks
]


@pytest.mark.parametrize("case", TWO_SOURCE_CASES, ids=str)
def test_conv3x3_persistent_two_source_many_work_items(hip, case):
    """Skip-concat form (two source views, src/model.py:65) on the persistent kernel with > 256 work items, and its
    two-destination data gradient."""
    L, ops = hip
    dtype = torch.bfloat16
    dt = ops._DT[dtype]
    n, c0, c1, co, h, w = case
    x2, x1 = rnd(f"b2_x2{case}", (n, c0, h, w)), rnd(f"b2_x1{case}", (n, c1, h, w))
    wt = rnd(f"b2_w{case}", (co, c0 + c1, 3, 3)) * (1.0 / (3 * (c0 + c1) ** 0.5))
    gy = rnd(f"b2_g{case}", (n, co, h, w))
    x2q, x1q, wq = q(x2, dtype).requires_grad_(True), q(x1, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True)
    ref = F.conv2d(torch.cat([x2q, x1q], 1), wq, padding=1)
    ref.backward(q(gy, dtype))
    x2d, x1d, gd, wd = nhwc(x2, dtype), nhwc(x1, dtype), nhwc(gy, dtype), wt.to(dev())
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wd, L.PACK_CONV_FWD, co, c0 + c1, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(x2d, 0, 0), (x1d, 0, 0)]), p(wp), co, views(L, [(y, 0, 0), None]), co, 0,
                                 L.K_CONV_FWD, st()), "conv fwd 2 src")
    check(y, ref, dtype, "two-source conv fwd (many work items)")
    d2, d1 = ops._nhwc_empty(n, c0, h, w, dtype, dev()), ops._nhwc_empty(n, c1, h, w, dtype, dev())
    wpd = ops.pack_weight(wd, L.PACK_CONV_DGRAD, c0 + c1, co, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), c0 + c1, views(L, [(d2, 0, 0), (d1, 0, 0)]),
                                 c0, 0, L.K_CONV_DGRAD, st()), "conv dgrad 2 dst")
    check(d2, x2q.grad, dtype, "two-destination dgrad, skip half")
    check(d1, x1q.grad, dtype, "two-destination dgrad, up half")
    # weight gradient over the two column sources (the XCD-aware block order of wgrad_dma_kernel)
    dw = torch.empty(co, c0 + c1, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_wgrad_workspace(n, h, w, c0 + c1, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_wgrad(dt, n, h, w, views(L, [(x2d, 0, 0), (x1d, 0, 0)]), p(gd), co, p(dw), c0 + c1, p(ws), need,
                                       st()), "conv wgrad 2 src")
    check(dw, wq.grad, dtype, "two-source wgrad", bf=5e-3)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 64, 64, 128, 17, 19, 16, 18, 0, 1), (3, 128, 128, 256, 9, 13, 8, 12, 1, 0),
                                  (2, 64, 128, 128, 33, 40, 32, 40, 0, 0)], ids=str)
def test_conv3x3_two_sources_with_centre_pad_offsets(hip, dtype, case):
    """The skip-concat of Up.forward with an up-sampled tensor SMALLER than the skip (src/model.py:57-65: F.pad to the
    skip's size, then cat): the second view sits at an offset inside the frame and is zero outside.  Forward, the
    two-destination data gradient and the weight gradient (>= 128 output channels: wgrad16_kernel in bf16, with frame
    widths that are not multiples of 32 and a source smaller than the frame) against F.conv2d on the padded concat."""
    L, ops = hip
    n, c0, c1, co, h, w, h1, w1, oy, ox = case
    x2, x1 = rnd(f"off_x2{case}", (n, c0, h, w)), rnd(f"off_x1{case}", (n, c1, h1, w1))
    wt = rnd(f"off_w{case}", (co, c0 + c1, 3, 3)) * (1.0 / (3 * (c0 + c1) ** 0.5))
    gy = rnd(f"off_g{case}", (n, co, h, w))
    x2q, x1q, wq = q(x2, dtype).requires_grad_(True), q(x1, dtype).requires_grad_(True), q(wt, dtype).requires_grad_(True)
    x1p = F.pad(x1q, [ox, w - w1 - ox, oy, h - h1 - oy])
    ref = F.conv2d(torch.cat([x2q, x1p], 1), wq, padding=1)
    ref.backward(q(gy, dtype))
    dt = ops._DT[dtype]
    x2d, x1d, gd, wd = nhwc(x2, dtype), nhwc(x1, dtype), nhwc(gy, dtype), wt.to(dev())
    src = views(L, [(x2d, 0, 0), (x1d, oy, ox)])
    y = ops._nhwc_empty(n, co, h, w, dtype, dev())
    wp = ops.pack_weight(wd, L.PACK_CONV_FWD, co, c0 + c1, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, src, p(wp), co, views(L, [(y, 0, 0), None]), co, 0, L.K_CONV_FWD, st()), "fwd")
    check(y, ref, dtype, "two-source forward with a padded second source")
    d2, d1 = ops._nhwc_empty(n, c0, h, w, dtype, dev()), ops._nhwc_empty(n, c1, h1, w1, dtype, dev())
    wpd = ops.pack_weight(wd, L.PACK_CONV_DGRAD, c0 + c1, co, dtype)
    L.check(L.lib().unet_conv3x3(dt, n, h, w, views(L, [(gd, 0, 0), None]), p(wpd), c0 + c1,
                                 views(L, [(d2, 0, 0), (d1, oy, ox)]), c0, 0, L.K_CONV_DGRAD, st()), "dgrad")
    check(d2, x2q.grad, dtype, "dgrad, skip half")
    check(d1, x1q.grad, dtype, "dgrad, up-sampled half (cropped back to its own size)")
    dw = torch.empty(co, c0 + c1, 3, 3, device=dev())
    need = L.lib().unet_conv3x3_wgrad_workspace(n, h, w, c0 + c1, co)
    ws = torch.empty(need, dtype=torch.uint8, device=dev())
    L.check(L.lib().unet_conv3x3_wgrad(dt, n, h, w, src, p(gd), co, p(dw), c0 + c1, p(ws), need, st()), "wgrad")
    check(dw, wq.grad, dtype, "wgrad over a padded second source", f32=5e-5, bf=5e-3)
