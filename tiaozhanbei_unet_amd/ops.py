"""autograd operators over the C-ABI of libunet_hip.so.

PyTorch is plumbing here: it owns device memory (the caching allocator), the stream and
the autograd tape; every FLOP of these operators runs in hand-written gfx950 HIP kernels
reached through ``_lib`` (ctypes).  Tensors between operators are logical-NCHW torch
tensors with ``channels_last`` strides (i.e. NHWC in memory) in the compute dtype
(``torch.bfloat16`` or ``torch.float32``).  There is no fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib as L

_DT = {torch.float32: L.UNET_F32, torch.bfloat16: L.UNET_BF16}
BN_EPS = 1e-5


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "tiaozhanbei_unet_amd runs only on an AMD GPU through libunet_hip.so; got a "
                f"{t.device} tensor (there is deliberately no CPU fallback)")


def _nhwc_empty(n, c, h, w, dtype, device):
    return torch.empty((n, c, h, w), dtype=dtype, device=device, memory_format=torch.channels_last)


def _is_nhwc(t: torch.Tensor) -> bool:
    """Dense NHWC in memory (strides of size-1 dims are irrelevant)."""
    n, c, h, w = t.shape
    want = (h * w * c, 1, w * c, c)
    return all(sz == 1 or st == wt for sz, st, wt in zip(t.shape, t.stride(), want))


def _as_nhwc(t: torch.Tensor, dtype) -> torch.Tensor:
    """Activation/gradient in the operators' layout.  (Only reached for tensors produced outside
    these operators, e.g. a test's upstream gradient.)"""
    if t.dtype == dtype and _is_nhwc(t):
        return t
    out = _nhwc_empty(*t.shape, dtype, t.device)
    out.copy_(t)
    return out


_workspaces = {}
_helpers = {}

# Data parallelism (ddp.GradientExchange) registers, per parameter, the slice of its flat all-reduce bucket: the
# weight-gradient kernels then write the gradient THERE (autograd adopts the returned tensor as ``param.grad`` when
# the parameter has no gradient yet), so no per-tensor copy into the bucket exists.
_grad_slots = {}           # parameter data_ptr -> (bucket view shaped like the parameter, weakref to the parameter)
_grad_taken = set()        # slots handed out in the current backward pass (a parameter used twice gets fresh memory)


def register_grad_slots(slots):
    _grad_slots.update(slots)


def unregister_grad_slots(keys):
    for k in keys:
        _grad_slots.pop(k, None)
        _grad_taken.discard(k)


def release_grad_slot(key):
    """The gradient of this parameter has been accumulated (post-accumulate hook): its slot may be handed out again."""
    _grad_taken.discard(key)


def grad_out(shape, device, key):
    """fp32 output tensor for the gradient of the parameter whose ``data_ptr()`` is ``key``: its registered bucket
    slice when the parameter holds no gradient yet (autograd then adopts the slice as ``param.grad``), else fresh memory."""
    slot = _grad_slots.get(key) if _grad_slots else None
    if slot is not None:
        view, ref = slot
        prm = ref()
        if prm is not None and prm.grad is None and key not in _grad_taken and tuple(view.shape) == tuple(shape) \
                and view.device == device:
            _grad_taken.add(key)
            return view.detach()            # a fresh alias: autograd adopts a gradient only if nobody else holds the tensor
    return torch.empty(tuple(shape), dtype=torch.float32, device=device)
WGRAD_SIDE_STREAM = __import__("os").environ.get("UNET_WGRAD_STREAM", "0") != "0"   # measured slower (-3 %): off


def _helper_stream(device, cur):
    """One helper stream per compute stream (weight gradients run beside data gradients)."""
    key = (device.index, cur.cuda_stream)
    if key not in _helpers:
        _helpers[key] = torch.cuda.Stream(device=device)
    return _helpers[key]



def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch arena per (device, stream): reuse is stream-ordered, so every compute stream has
    its own arena (the two decoders of AnomalyUNet run on two streams)."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _views(items) -> "L.View2":
    arr = L.View2()
    for i, it in enumerate(items):
        if it is None:
            arr[i] = L.View(None, 0, 0, 0, 0, 0)
        else:
            t, oy, ox = it
            arr[i] = L.View(t.data_ptr(), t.shape[1], t.shape[2], t.shape[3], oy, ox)
    return arr


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


# ----------------------------------------------------------------------------- layout
class PackInput(torch.autograd.Function):
    """NCHW fp32 image batch -> NHWC compute dtype, channels zero-padded to a multiple of 64
    (the tensor entering ``self.inc`` at /root/reference/src/model.py:190)."""

    @staticmethod
    def forward(ctx, x, dtype):
        _require_cuda(x)
        x = x.contiguous().float()
        n, c, h, w = x.shape
        cp = _pad64(c)
        out = _nhwc_empty(n, cp, h, w, dtype, x.device)
        L.check(L.lib().unet_nchw_to_nhwc(_ptr(x), _ptr(out), n, c, h, w, cp, _DT[dtype], _stream()),
                "unet_nchw_to_nhwc")
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, g):
        n, cp, h, w = g.shape
        g = _as_nhwc(g, g.dtype)
        out = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=g.device)
        L.check(L.lib().unet_nhwc_to_nchw(_ptr(g), _ptr(out), n, ctx.c, h, w, cp, _DT[g.dtype], _stream()),
                "unet_nhwc_to_nchw")
        return out, None


def to_operator_layout(x: torch.Tensor, dtype) -> torch.Tensor:
    """Accept what a caller of the reference modules would pass (NCHW fp32, any channel count) or an
    activation already in operator layout."""
    _require_cuda(x)
    if x.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got shape {tuple(x.shape)}")
    if x.dtype == dtype and _is_nhwc(x) and x.shape[1] % 64 == 0:
        return x
    if x.shape[1] % 64 == 0 and x.dtype == dtype:
        return _ToNHWC.apply(x)
    return PackInput.apply(x, dtype)


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _as_nhwc(x, x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def pack_weight(w: torch.Tensor, mode: int, rows: int, k: int, dtype) -> torch.Tensor:
    taps = 9 if mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD) else 4
    out = torch.empty(taps * rows * k, dtype=dtype, device=w.device)
    if mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD):
        co, ci = w.shape[0], w.shape[1]
    else:
        ci, co = w.shape[0], w.shape[1]
    L.check(L.lib().unet_pack_weight(_ptr(w), _ptr(out), co, ci, rows, k, mode, _DT[dtype], _stream()),
            "unet_pack_weight")
    return out


class PackCache:
    """Packed GEMM-layout copies of a model's conv / convT weights, refreshed by ONE batched launch whenever a
    parameter changed (its autograd version counter moved), instead of ~70 small pack launches per step."""

    def __init__(self, dtype):
        self.dtype = dtype
        self.items = []        # (weight, mode, rows, k)
        self.slots = {}        # (id(weight), mode) -> [packed view, rows, k, version]
        self.table = None
        self.ptrs = None
        self.gen = -1          # ops._train_generation when the packs were last written

    def add(self, weight, mode, rows, k):
        if (id(weight), mode) not in self.slots:
            self.items.append((weight, mode, rows, k))
            self.slots[(id(weight), mode)] = [None, rows, k, -1]

    def _build(self):
        import numpy as np
        es = 2 if self.dtype == torch.bfloat16 else 4
        sizes = [(9 if m <= L.PACK_CONV_DGRAD else 4) * r * k for (_, m, r, k) in self.items]
        offs = [0]
        for s_ in sizes:
            offs.append(offs[-1] + (s_ + 7) // 8 * 8)
        dev = self.items[0][0].device
        self.buf = torch.empty(offs[-1], dtype=self.dtype, device=dev)
        rec = np.zeros(len(self.items), dtype=[("w", "<u8"), ("out", "<u8"), ("c_out", "<i4"), ("c_in", "<i4"),
                                               ("rows", "<i4"), ("k", "<i4"), ("mode", "<i4"), ("pad", "<i4")])
        for i, (w, m, r, k) in enumerate(self.items):
            view = self.buf[offs[i]:offs[i] + sizes[i]]
            co, ci = (w.shape[0], w.shape[1]) if m <= L.PACK_CONV_DGRAD else (w.shape[1], w.shape[0])
            rec[i] = (w.data_ptr(), view.data_ptr(), co, ci, r, k, m, 0)
            self.slots[(id(w), m)][0] = view
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        self.ptrs = [w.data_ptr() for (w, _, _, _) in self.items]

    def refresh(self, force=False):
        """``force``: repack regardless of the version counters (fused optimisers update parameters without
        moving them, so a training forward always repacks: one ~0.1 ms launch).  An eval-mode forward repacks when
        any training forward happened since the last pack (``_train_generation`` moved): the optimiser step that
        followed it changed the weights through raw pointers that no version counter sees."""
        if not self.items:
            return
        if self.table is None or self.ptrs != [w.data_ptr() for (w, _, _, _) in self.items]:
            self._build()
            stale = True
        else:
            stale = force or self.gen != _train_generation or \
                any(self.slots[(id(w), m)][3] != w._version for (w, m, _, _) in self.items)
        if stale:
            L.check(L.lib().unet_pack_weights_batched(_ptr(self.table), len(self.items), _DT[self.dtype], _stream()),
                    "unet_pack_weights_batched")
            for (w, m, _, _) in self.items:
                self.slots[(id(w), m)][3] = w._version
            self.gen = _train_generation

    def get(self, weight, mode, rows, k):
        s_ = self.slots.get((id(weight), mode))
        if s_ is not None and s_[0] is not None and s_[1] == rows and s_[2] == k and s_[3] == weight._version:
            return s_[0]
        return None


_active_packs = None


def set_active_packs(cache):
    global _active_packs
    _active_packs = cache


def packed(weight, mode, rows, k, dtype):
    if _active_packs is not None and _active_packs.dtype == dtype:
        hit = _active_packs.get(weight, mode, rows, k)
        if hit is not None:
            return hit
    return pack_weight(weight, mode, rows, k, dtype)


# ----------------------------------------------------------------------------- gradient fan-in of the skips
class GradSink:
    """In-place gradient fan-in of an activation with several consumers (the skip tensors x1..x4 feed the
    next encoder level and both decoders, /root/reference/src/model.py:190-207).

    Autograd would materialise one gradient per consumer and add them with two more elementwise passes
    (9 tensor sweeps per skip).  With a sink the first consumer to run its backward WRITES the buffer and
    hands it to autograd, the later ones ACCUMULATE into it inside their own kernel epilogue
    (`accumulate` of unet_conv3x3 / unet_maxpool2_bwd) and return no gradient (5 sweeps).  The order is the
    autograd engine's (deterministic for a fixed graph); writers on different HIP streams are chained by
    events, and the producer's backward waits for the last one."""

    __slots__ = ("buf", "event")

    def __init__(self):
        self.buf = None
        self.event = None

    def begin(self, dev):
        """-> True when the caller must accumulate into ``self.buf`` (someone wrote it already)."""
        if self.buf is None:
            return False
        cur = torch.cuda.current_stream(dev)
        if self.event is not None:
            cur.wait_event(self.event)
        self.buf.record_stream(cur)
        return True

    def done(self, buf, dev):
        self.buf = buf
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(dev))

    def collect(self, grad, dev):
        """Called by the producer's backward with the gradient autograd delivered."""
        if self.buf is not None:
            if grad.data_ptr() != self.buf.data_ptr():
                raise RuntimeError("GradSink: autograd delivered a different tensor than the shared gradient buffer")
            cur = torch.cuda.current_stream(dev)
            if self.event is not None:
                cur.wait_event(self.event)
            self.buf.record_stream(cur)
        self.buf = None
        self.event = None


SHARE_SKIP_GRADS = __import__("os").environ.get("UNET_GRAD_SINK", "1") != "0"


def share_grad(t: torch.Tensor) -> torch.Tensor:
    """Mark an operator output whose gradient has several consumers (model code calls this on the skips)."""
    if SHARE_SKIP_GRADS and t.grad_fn is not None and hasattr(t.grad_fn, "out_sink"):
        sink = GradSink()
        t.grad_fn.out_sink = sink
        t._unet_sink = sink
    return t


FOLD_EVAL_BN = __import__("os").environ.get("UNET_FOLD_BN", "1") != "0"
_folded = {}                                             # id(weight) -> (weakref to it, {(dtype, ctot): (stamp, payload)})
_train_generation = 0                                    # bumped by every training forward (see below)


def _folded_pack(weight, gamma, beta, running_mean, running_var, co, ctot, dtype):
    """(packed folded weights, shift) of an eval-mode conv+BN layer.  Cached per weight Parameter; an entry is valid
    while the autograd version counters of the five tensors AND the training generation are unchanged -- the HIP
    kernels update running statistics (and fused optimisers update parameters) through raw pointers, which no version
    counter sees, so any training forward in between invalidates every entry."""
    stamp = (_train_generation, weight._version, gamma._version, beta._version, running_mean._version,
             running_var._version, running_mean.data_ptr(), running_var.data_ptr(), weight.data_ptr())
    import weakref
    slot = _folded.get(id(weight))
    if slot is None or slot[0]() is not weight:          # first use, or the id was recycled by another tensor
        key = id(weight)
        slot = (weakref.ref(weight, lambda _r, k=key: _folded.pop(k, None)), {})
        _folded[key] = slot
    per_weight = slot[1]
    hit = per_weight.get((dtype, ctot))
    if hit is not None and hit[0] == stamp:
        return hit[1]
    lib, st, dev = L.lib(), _stream(), weight.device
    ss = torch.empty((2, co), dtype=torch.float32, device=dev)
    L.check(lib.unet_bn_eval_coeffs(co, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), BN_EPS,
                                    _ptr(ss[0]), _ptr(ss[1]), st), "unet_bn_eval_coeffs")
    wq = torch.empty(9 * co * ctot, dtype=dtype, device=dev)
    L.check(lib.unet_pack_conv_weight_folded(_ptr(weight.detach()), _ptr(ss[0]), _ptr(wq), co, weight.shape[1], co, ctot,
                                             _DT[dtype], st), "unet_pack_conv_weight_folded")
    per_weight[(dtype, ctot)] = (stamp, (wq, ss[1]))
    return wq, ss[1]


# ----------------------------------------------------------------------------- conv3x3 + BN + ReLU
FUSE_BN_BWD = __import__("os").environ.get("UNET_FUSE_BN_BWD", "1") != "0"      # tuning hook (A/B runs)
FUSE_BN_HEAD = __import__("os").environ.get("UNET_FUSE_BN_HEAD", "1") != "0"
FUSE_BN_POOL = __import__("os").environ.get("UNET_FUSE_BN_POOL", "1") != "0"
FUSE_BN_CONVT = __import__("os").environ.get("UNET_FUSE_BN_CONVT", "1") != "0"


class BnLink:
    """Side channel between a conv-BN-ReLU layer and the ONE operator that consumes its activation (the second
    convolution of the same DoubleConv, /root/reference/src/model.py:14-19).  In the backward pass the consumer's data
    gradient kernel applies this layer's ReLU mask in its epilogue and leaves the BatchNorm-backward partial sums
    (sum dz, sum dz*(y - mean)) here, so the producer skips its reduction pass over (y, da).  Only valid for an
    activation with exactly one consumer -- the model code creates links for DoubleConv's internal tensor only."""

    __slots__ = ("y", "coef", "partial", "n_parts", "dz_ptr")

    def __init__(self):
        self.y = None          # raw conv output of the producer (NHWC compute dtype)
        self.coef = None       # [4, C]: mean, istd, scale, shift
        self.partial = None
        self.n_parts = 0
        self.dz_ptr = 0        # data_ptr of the premasked gradient the consumer returned (0: not premasked)


def _bn_stats_conv(lib, dt, n, h, w, src, wp, co, y, gamma, beta, running_mean, running_var, momentum, coef, dev, st):
    """conv + BatchNorm batch statistics in one call (the conv epilogue reduces sum / sum of squares per channel
    with wavefront shuffles, or one extra streaming pass for kernels without that epilogue) + fp64 finalize."""
    cap = lib.unet_conv3x3_stats_max_parts(n, h, w)
    part = _workspace(cap * 2 * co * 4, dev)
    nparts = C.c_int32(0)
    L.check(lib.unet_conv3x3_stats(dt, n, h, w, src, _ptr(wp), co, _ptr(y), _ptr(part), C.byref(nparts), st),
            "unet_conv3x3_stats")
    L.check(lib.unet_bn_finalize_partials(_ptr(part), nparts.value, n * h * w, co, _ptr(gamma), _ptr(beta),
                                          _ptr(running_mean), _ptr(running_var), momentum, BN_EPS,
                                          _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]), st),
            "unet_bn_finalize_partials")


def _bn_relu_backward(lib, dt, dtype, da, y, gamma, coef, link, out_sink, dev, st, frozen=False, beta_key=None):
    """Gradient w.r.t. the raw conv output of a conv-BN-ReLU layer -> (dy, dgamma/dbeta [2, C]).  Premasked path: the
    consumer's data-gradient kernel already applied the ReLU mask and reduced the BatchNorm-backward sums (BnLink).
    ``frozen``: the layer normalised with its running statistics (BatchNorm2d in eval mode inside a training graph)."""
    n, co, h, w = y.shape
    pixels = n * h * w
    dgb = (grad_out(gamma.shape, dev, gamma.data_ptr()), grad_out(gamma.shape, dev, beta_key))
    if link is not None and link.dz_ptr and link.dz_ptr == da.data_ptr() and da.dtype == dtype and _is_nhwc(da):
        dy = da
        ws = _workspace(3 * co * 4, dev)
        L.check(lib.unet_bn_bwd_premasked(dt, _ptr(da), _ptr(y), pixels, co, _ptr(gamma), _ptr(coef[0]),
                                          _ptr(coef[1]), _ptr(link.partial), link.n_parts, _ptr(dgb[0]),
                                          _ptr(dgb[1]), _ptr(dy), _ptr(ws), ws.numel(), st), "unet_bn_bwd_premasked")
    else:
        if link is not None and link.dz_ptr:
            raise RuntimeError("conv-BN-ReLU: the premasked gradient of a linked activation did not arrive unchanged "
                               "(the activation has a second consumer?)")
        if out_sink is not None:
            out_sink.collect(da, dev)
        da = _as_nhwc(da, dtype)
        dy = _nhwc_empty(n, co, h, w, dtype, dev)
        ws = _workspace(lib.unet_bn_workspace(pixels, co), dev)
        fn = lib.unet_bn_relu_bwd_frozen if frozen else lib.unet_bn_relu_bwd
        L.check(fn(dt, _ptr(da), _ptr(y), pixels, co, _ptr(gamma), _ptr(coef[0]), _ptr(coef[1]),
                   _ptr(coef[2]), _ptr(coef[3]), _ptr(dgb[0]), _ptr(dgb[1]), _ptr(dy),
                   _ptr(ws), ws.numel(), st), "unet_bn_relu_bwd")
    if link is not None:
        link.y = link.coef = link.partial = None
        link.dz_ptr = 0
    return dy, dgb


class ConvBnRelu(torch.autograd.Function):
    """relu(batch_norm(conv3x3(cat([x0, x1])))) -- one third of DoubleConv
    (/root/reference/src/model.py:14-16 / :17-19); ``x1`` (optional) is the up-sampled tensor
    of Up.forward, centre-padded to x0's size (src/model.py:57-65) without materialising pad or cat.

    Optional fusions of the training path (all value-preserving):
      * ``head_w`` / ``head_b`` / ``head_sigmoid``: the layer is followed by OutConv (src/model.py:72): the 1x1 head
        reads the RAW conv output and applies BatchNorm + ReLU on load, its backward writes the ReLU-masked gradient
        and the BatchNorm-backward sums -- no activation tensor, no BN-apply pass, no (y, da) reduction pass.
        The function then returns the head's NCHW fp32 output.
      * ``out_link`` / ``in_link`` (BnLink): see BnLink.
      * ``pool``: the layer closes an encoder level (src/model.py:18-19, next level's MaxPool2d :32): BatchNorm-apply +
        ReLU + 2x2 max pool in one pass; returns (a, maxpool2(a)).  The backward routes the pooled gradient, adds it to
        what the decoders left in the skip's gradient buffer, masks and reduces in one pass (no separate pool backward,
        no (y, da) reduction pass)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, gamma, beta, running_mean, running_var, training, momentum, fold=False,
                in_link=None, out_link=None, head_w=None, head_b=None, head_sigmoid=False, pool=False):
        _require_cuda(x0, weight)
        dtype = x0.dtype
        dt = _DT[dtype]
        n, c0, h, w = x0.shape
        c1 = 0 if x1 is None else x1.shape[1]
        oy = ox = 0
        if x1 is not None:
            dy_, dx_ = h - x1.shape[2], w - x1.shape[3]
            if dy_ < 0 or dx_ < 0:
                raise ValueError("Up: the skip tensor must be at least as large as the up-sampled one")
            oy, ox = dy_ // 2, dx_ // 2
        co, ci = weight.shape[0], weight.shape[1]
        ctot = c0 + c1
        if not (ci <= ctot < ci + 64):
            raise ValueError(f"conv weight expects {ci} input channels, activations carry {ctot}")
        if (head_w is not None or pool) and not training:
            raise RuntimeError("ConvBnRelu: the fused head / pool are training-path fusions")
        lib, st, dev = L.lib(), _stream(), x0.device
        y = _nhwc_empty(n, co, h, w, dtype, dev)
        src = _views([(x0, 0, 0), None if x1 is None else (x1, oy, ox)])
        pixels = n * h * w
        coef = torch.empty((4, co), dtype=torch.float32, device=dev)   # mean, istd, scale, shift
        fold = bool(fold) and FOLD_EVAL_BN and not training
        if not fold:
            wp = packed(weight, L.PACK_CONV_FWD, co, ctot, dtype)
        if training:
            global _train_generation
            _train_generation += 1
            _bn_stats_conv(lib, dt, n, h, w, src, wp, co, y, gamma, beta, running_mean, running_var, momentum, coef,
                           dev, st)
        elif fold:
            # inference: BatchNorm(eval) folded into the layer -- scale into the packed weights, shift + ReLU in the
            # convolution's epilogue: one kernel, the activation is written once
            wf = _folded_pack(weight, gamma, beta, running_mean, running_var, co, ctot, dtype)
            L.check(lib.unet_conv3x3_bias_relu(dt, n, h, w, src, _ptr(wf[0]), co, _ptr(y), _ptr(wf[1]), 1, st),
                    "unet_conv3x3_bias_relu")
            return y
        else:
            dst = _views([(y, 0, 0), None])
            L.check(lib.unet_conv3x3(dt, n, h, w, src, _ptr(wp), co, dst, co, 0, L.K_CONV_FWD, st), "unet_conv3x3")
            L.check(lib.unet_bn_eval_coeffs4(co, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                             BN_EPS, _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]), st),
                    "unet_bn_eval_coeffs4")
        ctx.geom = (oy, ox, training)
        ctx.keys = (weight.data_ptr(), beta.data_ptr(), 0 if head_w is None else head_w.data_ptr(),
                    0 if head_b is None else head_b.data_ptr())          # gradient bucket slots (data parallelism)
        ctx.sink0 = getattr(x0, "_unet_sink", None)     # x0 is a skip with a shared gradient buffer
        ctx.out_sink = None                              # set by share_grad() when THIS output is a skip
        ctx.in_link = in_link if (in_link is not None and in_link.y is not None and x1 is None) else None
        ctx.out_link = None
        ctx.head = None
        ctx.pool = False
        if pool:
            a = _nhwc_empty(n, co, h, w, dtype, dev)
            pooled = _nhwc_empty(n, co, h // 2, w // 2, dtype, dev)
            L.check(lib.unet_bn_relu_pool_fwd(dt, _ptr(y), n, h, w, co, _ptr(coef[2]), _ptr(coef[3]), _ptr(a),
                                              _ptr(pooled), st), "unet_bn_relu_pool_fwd")
            ctx.save_for_backward(x0, x1, y, weight, gamma, coef)
            ctx.pool = True
            ctx.set_materialize_grads(False)
            return a, pooled
        if head_w is not None:
            hc = head_w.shape[0]
            out = torch.empty((n, hc, h, w), dtype=torch.float32, device=dev)
            L.check(lib.unet_head_bnrelu_fwd(dt, _ptr(y), n, h, w, co, _ptr(coef[2]), _ptr(coef[3]), _ptr(head_w),
                                             _ptr(head_b), hc, int(head_sigmoid), _ptr(out), st), "unet_head_bnrelu_fwd")
            ctx.save_for_backward(x0, x1, y, weight, gamma, coef, head_w, out)
            ctx.head = bool(head_sigmoid)
            return out
        a = _nhwc_empty(n, co, h, w, dtype, dev)
        L.check(lib.unet_bn_relu_apply(dt, _ptr(y), pixels, co, _ptr(coef[2]), _ptr(coef[3]), _ptr(a), st),
                "unet_bn_relu_apply")
        ctx.save_for_backward(x0, x1, y, weight, gamma, coef)
        if out_link is not None and training:
            out_link.y, out_link.coef, out_link.dz_ptr = y, coef, 0
            ctx.out_link = out_link
        return a

    @staticmethod
    def backward(ctx, da, dpooled=None):
        saved = ctx.saved_tensors
        x0, x1, y, weight, gamma, coef = saved[:6]
        oy, ox, training = ctx.geom
        dtype = x0.dtype
        dt = _DT[dtype]
        n, c0, h, w = x0.shape
        co, ci = weight.shape[0], weight.shape[1]
        ctot = c0 + (0 if x1 is None else x1.shape[1])
        lib, st, dev = L.lib(), _stream(), x0.device
        pixels = n * h * w
        dhw = dhb = None
        link = ctx.out_link
        if ctx.head is not None:
            # OutConv backward + ReLU mask + BatchNorm-backward sums in one pass, then dy = A*dz + B*y + K in place
            head_w, out = saved[6], saved[7]
            hc = head_w.shape[0]
            dout = da.contiguous().float()
            dy = _nhwc_empty(n, co, h, w, dtype, dev)
            dgb = (grad_out(gamma.shape, dev, gamma.data_ptr()), grad_out(gamma.shape, dev, ctx.keys[1]))
            dhw = grad_out(head_w.shape, dev, ctx.keys[2])
            dhb = grad_out((hc,), dev, ctx.keys[3])
            part = torch.empty((lib.unet_head_bnrelu_max_parts(), 2, co), dtype=torch.float32, device=dev)
            nparts = C.c_int32(0)
            ws = _workspace(lib.unet_head_bwd_workspace(n, h, w, co, hc), dev)
            L.check(lib.unet_head_bnrelu_bwd(dt, _ptr(y), _ptr(coef[2]), _ptr(coef[3]), _ptr(coef[0]), _ptr(out),
                                             _ptr(dout), n, h, w, co, _ptr(head_w), hc, int(ctx.head), _ptr(dy),
                                             _ptr(dhw), _ptr(dhb), _ptr(part), C.byref(nparts), _ptr(ws), ws.numel(),
                                             st), "unet_head_bnrelu_bwd")
            ws = _workspace(3 * co * 4, dev)
            L.check(lib.unet_bn_bwd_premasked(dt, _ptr(dy), _ptr(y), pixels, co, _ptr(gamma), _ptr(coef[0]),
                                              _ptr(coef[1]), _ptr(part), nparts.value, _ptr(dgb[0]), _ptr(dgb[1]),
                                              _ptr(dy), _ptr(ws), ws.numel(), st), "unet_bn_bwd_premasked")
        elif ctx.pool and dpooled is not None:
            # pooled gradient routed + added to the skip's gradient buffer + ReLU mask + BatchNorm-backward sums: one pass
            own = False            # `da` is the GradSink buffer of this skip: nobody else holds it, safe to overwrite
            if da is not None:
                if ctx.out_sink is not None and ctx.out_sink.buf is not None and \
                        da.data_ptr() == ctx.out_sink.buf.data_ptr():
                    own = True
                if ctx.out_sink is not None:
                    ctx.out_sink.collect(da, dev)
                da = _as_nhwc(da, dtype)
            dpooled = _as_nhwc(dpooled, dtype)
            # in place over the skip's own gradient buffer; a gradient autograd delivered from elsewhere may be shared
            # with another node (e.g. add-backward hands one tensor to both inputs) and is left untouched
            dy = da if own else _nhwc_empty(n, co, h, w, dtype, dev)
            part = torch.empty((lib.unet_bn_relu_pool_max_parts(), 2, co), dtype=torch.float32, device=dev)
            nparts = C.c_int32(0)
            L.check(lib.unet_bn_relu_pool_bwd(dt, _ptr(y), _ptr(dpooled), _ptr(da), n, h, w, co, _ptr(coef[2]),
                                              _ptr(coef[3]), _ptr(coef[0]), _ptr(dy), _ptr(part), C.byref(nparts), st),
                    "unet_bn_relu_pool_bwd")
            dgb = (grad_out(gamma.shape, dev, gamma.data_ptr()), grad_out(gamma.shape, dev, ctx.keys[1]))
            ws = _workspace(3 * co * 4, dev)
            L.check(lib.unet_bn_bwd_premasked(dt, _ptr(dy), _ptr(y), pixels, co, _ptr(gamma), _ptr(coef[0]),
                                              _ptr(coef[1]), _ptr(part), nparts.value, _ptr(dgb[0]), _ptr(dgb[1]),
                                              _ptr(dy), _ptr(ws), ws.numel(), st), "unet_bn_bwd_premasked")
        else:
            if da is None:                       # (a pooled pair whose skip half nobody used, and no pooled gradient)
                da = torch.zeros_like(y)
            dy, dgb = _bn_relu_backward(lib, dt, dtype, da, y, gamma, coef, link, ctx.out_sink, dev, st,
                                        frozen=not training, beta_key=ctx.keys[1])
        src = _views([(x0, 0, 0), None if x1 is None else (x1, oy, ox)])
        dw = None
        wgrad_done = None
        if ctx.needs_input_grad[2]:
            # the weight gradient only depends on dy and the saved input: it runs on a helper stream, next to
            # the data gradient (their ramp-up / tail phases overlap); joined before this node returns
            cur = torch.cuda.current_stream(dev)
            helper = _helper_stream(dev, cur) if WGRAD_SIDE_STREAM else None
            if helper is not None:
                helper.wait_stream(cur)
                torch.cuda.set_stream(helper)
            try:
                dw = grad_out(weight.shape, dev, ctx.keys[0])
                need = lib.unet_conv3x3_wgrad_workspace(n, h, w, ctot, co)
                ws2 = _workspace(need, dev)
                L.check(lib.unet_conv3x3_wgrad(dt, n, h, w, src, _ptr(dy), co, _ptr(dw), ci, _ptr(ws2), ws2.numel(),
                                               _stream()), "unet_conv3x3_wgrad")
                if helper is not None:
                    wgrad_done = torch.cuda.Event()
                    wgrad_done.record(helper)
                    dw.record_stream(cur)
                    dy.record_stream(helper)
            finally:
                if helper is not None:
                    torch.cuda.set_stream(cur)
        dx0 = dx1 = None
        if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
            wp = packed(weight, L.PACK_CONV_DGRAD, ctot, co, dtype)
            sink = ctx.sink0
            ilink = ctx.in_link if (FUSE_BN_BWD and sink is None and x1 is None) else None
            if ilink is not None and ilink.y is not None and \
                    lib.unet_conv3x3_dgrad_bnrelu_supported(dt, n, h, w, co, c0):
                # data gradient + ReLU mask of the PRODUCER of x0 + its BatchNorm-backward sums in one kernel
                dx0 = _nhwc_empty(n, c0, h, w, dtype, dev)
                cap = lib.unet_conv3x3_stats_max_parts(n, h, w)
                part = torch.empty((cap, 2, c0), dtype=torch.float32, device=dev)
                nparts = C.c_int32(0)
                pc = ilink.coef
                L.check(lib.unet_conv3x3_dgrad_bnrelu(dt, n, h, w, _ptr(dy), co, _ptr(wp), c0, _ptr(ilink.y),
                                                      _ptr(pc[2]), _ptr(pc[3]), _ptr(pc[0]), _ptr(dx0), _ptr(part),
                                                      C.byref(nparts), st), "unet_conv3x3_dgrad_bnrelu")
                ilink.partial, ilink.n_parts, ilink.dz_ptr = part, nparts.value, dx0.data_ptr()
            else:
                fan_in = sink is not None and sink.begin(dev)
                dx0 = sink.buf if fan_in else _nhwc_empty(n, c0, h, w, dtype, dev)
                if x1 is not None:
                    dx1 = _nhwc_empty(*x1.shape, dtype, dev)
                dsrc = _views([(dy, 0, 0), None])
                ddst = _views([(dx0, 0, 0), None if x1 is None else (dx1, oy, ox)])
                L.check(lib.unet_conv3x3(dt, n, h, w, dsrc, _ptr(wp), ctot, ddst, c0, 1 if fan_in else 0,
                                         L.K_CONV_DGRAD, st), "unet_conv3x3(dgrad)")
                if sink is not None:
                    sink.done(dx0, dev)
                    if fan_in:
                        dx0 = None              # already inside the buffer the first consumer returned
        if wgrad_done is not None:
            torch.cuda.current_stream(dev).wait_event(wgrad_done)
        return dx0, dx1, dw, dgb[0], dgb[1], None, None, None, None, None, None, None, dhw, dhb, None, None


class FirstConvBnRelu(torch.autograd.Function):
    """relu(batch_norm(conv3x3(image))) for the first layer of the network (inc.double_conv.0..2,
    /root/reference/src/model.py:14-16) in bf16 mode, straight from the caller's fp32 NCHW image: with 9*Cin <= 32
    the reduction is one MFMA step (unet_conv3x3_first_*), no 64-channel padded copy of the image exists."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, training, momentum, out_link=None):
        _require_cuda(x, weight)
        x = x.contiguous().float()
        n, ci, h, w = x.shape
        co = weight.shape[0]
        dtype = torch.bfloat16
        lib, st, dev = L.lib(), _stream(), x.device
        y = _nhwc_empty(n, co, h, w, dtype, dev)
        pixels = n * h * w
        coef = torch.empty((4, co), dtype=torch.float32, device=dev)
        wq = weight.detach().contiguous()
        if training:
            cap = lib.unet_conv3x3_stats_max_parts(n, h, w)
            part = _workspace(cap * 2 * co * 4, dev)
            nparts = C.c_int32(0)
            L.check(lib.unet_conv3x3_first_stats(n, h, w, _ptr(x), ci, _ptr(wq), _ptr(y), _ptr(part), C.byref(nparts), st),
                    "unet_conv3x3_first_stats")
            L.check(lib.unet_bn_finalize_partials(_ptr(part), nparts.value, pixels, co, _ptr(gamma), _ptr(beta),
                                                  _ptr(running_mean), _ptr(running_var), momentum, BN_EPS,
                                                  _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]), st),
                    "unet_bn_finalize_partials")
        else:
            L.check(lib.unet_conv3x3_first_stats(n, h, w, _ptr(x), ci, _ptr(wq), _ptr(y), None, None, st),
                    "unet_conv3x3_first_stats")
            L.check(lib.unet_bn_eval_coeffs4(co, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                             BN_EPS, _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]), st),
                    "unet_bn_eval_coeffs4")
        a = _nhwc_empty(n, co, h, w, dtype, dev)
        L.check(lib.unet_bn_relu_apply(_DT[dtype], _ptr(y), pixels, co, _ptr(coef[2]), _ptr(coef[3]), _ptr(a), st),
                "unet_bn_relu_apply")
        ctx.save_for_backward(x, y, weight, gamma, coef)
        ctx.training = training
        ctx.keys = (weight.data_ptr(), beta.data_ptr())
        ctx.out_sink = None
        ctx.out_link = None
        if out_link is not None and training:
            out_link.y, out_link.coef, out_link.dz_ptr = y, coef, 0
            ctx.out_link = out_link
        return a

    @staticmethod
    def backward(ctx, da):
        x, y, weight, gamma, coef = ctx.saved_tensors
        dtype = torch.bfloat16
        dt = _DT[dtype]
        n, ci, h, w = x.shape
        co = weight.shape[0]
        lib, st, dev = L.lib(), _stream(), x.device
        link = ctx.out_link
        if FUSE_FIRST_BN_BWD and link is not None and link.dz_ptr and link.dz_ptr == da.data_ptr() and da.dtype == dtype \
                and _is_nhwc(da) and ctx.training and ctx.needs_input_grad[1]:
            # The image layer's dy has ONE consumer, its weight gradient (nothing flows back into the image): the
            # BatchNorm-backward apply pass is folded into that kernel's operand -- finalize only (dy = NULL), then
            # unet_conv3x3_first_wgrad_bn streams dz and y and forms dy = A*dz + B*y + K per lane (bit-identical dW).
            dgb = (grad_out(gamma.shape, dev, gamma.data_ptr()), grad_out(gamma.shape, dev, ctx.keys[1]))
            cf = torch.empty(3 * co, dtype=torch.float32, device=dev)
            L.check(lib.unet_bn_bwd_premasked(dt, None, None, n * h * w, co, _ptr(gamma), _ptr(coef[0]), _ptr(coef[1]),
                                              _ptr(link.partial), link.n_parts, _ptr(dgb[0]), _ptr(dgb[1]), None, _ptr(cf),
                                              cf.numel() * 4, st), "unet_bn_bwd_premasked(coefficients)")
            link.y = link.coef = link.partial = None
            link.dz_ptr = 0
            dw = grad_out(weight.shape, dev, ctx.keys[0])
            ws2 = _workspace(lib.unet_conv3x3_first_wgrad_workspace(n, h, w), dev)
            L.check(lib.unet_conv3x3_first_wgrad_bn(n, h, w, _ptr(x), ci, _ptr(da), _ptr(y), _ptr(cf), _ptr(dw), _ptr(ws2),
                                                    ws2.numel(), st), "unet_conv3x3_first_wgrad_bn")
            return None, dw, dgb[0], dgb[1], None, None, None, None, None
        dy, dgb = _bn_relu_backward(lib, dt, dtype, da, y, gamma, coef, ctx.out_link, ctx.out_sink, dev, st,
                                    frozen=not ctx.training, beta_key=ctx.keys[1])
        dw = None
        if ctx.needs_input_grad[1]:
            dw = grad_out(weight.shape, dev, ctx.keys[0])
            need = lib.unet_conv3x3_first_wgrad_workspace(n, h, w)
            ws2 = _workspace(need, dev)
            L.check(lib.unet_conv3x3_first_wgrad(n, h, w, _ptr(x), ci, _ptr(dy), _ptr(dw), _ptr(ws2), ws2.numel(), st),
                    "unet_conv3x3_first_wgrad")
        return None, dw, dgb[0], dgb[1], None, None, None, None, None


FIRST_LAYER_KERNELS = __import__("os").environ.get("UNET_FIRST_LAYER", "1") != "0"
FUSE_FIRST_BN_BWD = __import__("os").environ.get("UNET_FUSE_FIRST_BN", "1") != "0"      # tuning hook (A/B runs)


def first_layer_ok(x: torch.Tensor, conv, dtype) -> bool:
    """The image layer qualifies for the one-MFMA-step kernels: bf16 mode, a plain fp32 NCHW image that needs no
    gradient, <= 3 channels into 64, width a multiple of 16."""
    return (FIRST_LAYER_KERNELS and dtype == torch.bfloat16 and x.dim() == 4 and x.dtype == torch.float32
            and not x.requires_grad and x.shape[1] == conv.in_channels
            and bool(L.lib().unet_conv3x3_first_supported(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3])))


class ChannelDropout(torch.autograd.Function):
    """nn.Dropout2d on an NHWC activation (SegmentationUNet's bottleneck, /root/reference/src/model.py:129,146):
    ``noise`` = bernoulli(1-p)/(1-p) per (image, channel), drawn by the caller exactly as torch's feature dropout
    draws it; y = x * noise, dx = dy * noise (unet_channel_scale)."""

    @staticmethod
    def forward(ctx, x, noise):
        _require_cuda(x, noise)
        n, c, h, w = x.shape
        y = _nhwc_empty(n, c, h, w, x.dtype, x.device)
        L.check(L.lib().unet_channel_scale(_DT[x.dtype], _ptr(x), _ptr(noise), n, h * w, c, _ptr(y), _stream()),
                "unet_channel_scale")
        ctx.save_for_backward(noise)
        return y

    @staticmethod
    def backward(ctx, dy):
        (noise,) = ctx.saved_tensors
        n, c, h, w = dy.shape
        dy = _as_nhwc(dy, dy.dtype)
        dx = _nhwc_empty(n, c, h, w, dy.dtype, dy.device)
        L.check(L.lib().unet_channel_scale(_DT[dy.dtype], _ptr(dy), _ptr(noise), n, h * w, c, _ptr(dx), _stream()),
                "unet_channel_scale")
        return dx, None


def anomaly_score(reconstruction, original, l1=False):
    """(score map [N, H, W], image score [N]) of compute_anomaly_score (/root/reference/src/utils.py:205-215)."""
    _require_cuda(reconstruction, original)
    r = reconstruction.detach().contiguous().float()
    o = original.detach().contiguous().float()
    n, c = r.shape[0], r.shape[1]
    hw = int(r.numel() // (n * c))
    score = torch.empty((n,) + tuple(r.shape[2:]), dtype=torch.float32, device=r.device)
    img = torch.empty(n, dtype=torch.float32, device=r.device)
    lib = L.lib()
    ws = _workspace(lib.unet_anomaly_score_workspace(n, hw), r.device)
    L.check(lib.unet_anomaly_score(_ptr(r), _ptr(o), n, c, hw, 1 if l1 else 0, _ptr(score), _ptr(img), _ptr(ws), ws.numel(),
                                   _stream()), "unet_anomaly_score")
    return score, img


# ----------------------------------------------------------------------------- max pool
class MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2) (/root/reference/src/model.py:32)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        n, c, h, w = x.shape
        if h < 2 or w < 2:
            raise ValueError("MaxPool2d(2) needs at least 2x2 pixels")
        y = _nhwc_empty(n, c, h // 2, w // 2, x.dtype, x.device)
        L.check(L.lib().unet_maxpool2_fwd(_DT[x.dtype], _ptr(x), n, h, w, c, _ptr(y), _stream()), "unet_maxpool2_fwd")
        ctx.save_for_backward(x)
        ctx.sink = getattr(x, "_unet_sink", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, c, h, w = x.shape
        dy = _as_nhwc(dy, x.dtype)
        sink = ctx.sink
        fan_in = sink is not None and sink.begin(x.device)
        dx = sink.buf if fan_in else _nhwc_empty(n, c, h, w, x.dtype, x.device)
        L.check(L.lib().unet_maxpool2_bwd(_DT[x.dtype], _ptr(x), _ptr(dy), n, h, w, c, _ptr(dx), 1 if fan_in else 0,
                                          _stream()), "unet_maxpool2_bwd")
        if sink is not None:
            sink.done(dx, x.device)
            if fan_in:
                return None
        return dx


# ----------------------------------------------------------------------------- up-sampling
class ConvT2x2(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cin//2, kernel_size=2, stride=2) (/root/reference/src/model.py:51)."""

    @staticmethod
    def forward(ctx, x, weight, bias, in_link=None):
        """``in_link`` (BnLink, optional): ``x`` is the activation of a conv-BN-ReLU layer with no other consumer (the
        DoubleConv of the previous Up block): the data-gradient kernel then applies that layer's ReLU mask and reduces
        its BatchNorm-backward sums (unet_convt2x2_dgrad_bnrelu)."""
        _require_cuda(x, weight)
        n, ci, h, w = x.shape
        co = weight.shape[1]
        dtype = x.dtype
        wp = packed(weight, L.PACK_CONVT_FWD, co, ci, dtype)
        y = _nhwc_empty(n, co, 2 * h, 2 * w, dtype, x.device)
        L.check(L.lib().unet_convt2x2_fwd(_DT[dtype], n, h, w, _ptr(x), ci, _ptr(wp), _ptr(bias), _ptr(y), co,
                                          _stream()), "unet_convt2x2_fwd")
        ctx.save_for_backward(x, weight)
        ctx.keys = (weight.data_ptr(), bias.data_ptr())
        ctx.in_link = in_link if (in_link is not None and in_link.y is not None and
                                  in_link.y.data_ptr() != 0 and tuple(in_link.y.shape) == tuple(x.shape)) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = weight.shape[1]
        dtype = x.dtype
        lib, st, dev = L.lib(), _stream(), x.device
        dy = _as_nhwc(dy, dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            wp = packed(weight, L.PACK_CONVT_DGRAD, ci, co, dtype)
            dx = _nhwc_empty(n, ci, h, w, dtype, dev)
            ilink = ctx.in_link if (FUSE_BN_BWD and FUSE_BN_CONVT) else None
            if ilink is not None and ilink.y is not None and \
                    lib.unet_convt2x2_dgrad_bnrelu_supported(_DT[dtype], n, h, w, ci, co):
                part = torch.empty((lib.unet_convt2x2_dgrad_bnrelu_max_parts(), 2, ci), dtype=torch.float32, device=dev)
                nparts = C.c_int32(0)
                pc = ilink.coef
                L.check(lib.unet_convt2x2_dgrad_bnrelu(_DT[dtype], n, h, w, _ptr(dy), co, _ptr(wp), _ptr(ilink.y),
                                                       _ptr(pc[2]), _ptr(pc[3]), _ptr(pc[0]), _ptr(dx), ci, _ptr(part),
                                                       C.byref(nparts), st), "unet_convt2x2_dgrad_bnrelu")
                ilink.partial, ilink.n_parts, ilink.dz_ptr = part, nparts.value, dx.data_ptr()
            else:
                L.check(lib.unet_convt2x2_dgrad(_DT[dtype], n, h, w, _ptr(dy), co, _ptr(wp), _ptr(dx), ci, st),
                        "unet_convt2x2_dgrad")
        dw = grad_out(weight.shape, dev, ctx.keys[0])
        db = grad_out((co,), dev, ctx.keys[1])
        ws = _workspace(lib.unet_convt2x2_wgrad_workspace(n, h, w, ci, co), dev)
        L.check(lib.unet_convt2x2_wgrad(_DT[dtype], n, h, w, _ptr(x), ci, _ptr(dy), co, _ptr(dw), _ptr(db),
                                        _ptr(ws), ws.numel(), st), "unet_convt2x2_wgrad")
        return dx, dw, db, None


class Bilinear2x(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (/root/reference/src/model.py:48)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        n, c, h, w = x.shape
        y = _nhwc_empty(n, c, 2 * h, 2 * w, x.dtype, x.device)
        L.check(L.lib().unet_upsample_bilinear2x_fwd(_DT[x.dtype], _ptr(x), n, h, w, c, _ptr(y), _stream()),
                "unet_upsample_bilinear2x_fwd")
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = _as_nhwc(dy, dy.dtype)
        dx = _nhwc_empty(n, c, h, w, dy.dtype, dy.device)
        L.check(L.lib().unet_upsample_bilinear2x_bwd(_DT[dy.dtype], _ptr(dy), n, h, w, c, _ptr(dx), _stream()),
                "unet_upsample_bilinear2x_bwd")
        return dx


# ----------------------------------------------------------------------------- 1x1 head
class Head(torch.autograd.Function):
    """OutConv (1x1 conv + bias, /root/reference/src/model.py:72) with the optional sigmoid of
    AnomalyUNet.forward (src/model.py:201,208).  Output: NCHW fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, sigmoid):
        _require_cuda(x, weight)
        n, ci, h, w = x.shape
        co = weight.shape[0]
        out = torch.empty((n, co, h, w), dtype=torch.float32, device=x.device)
        L.check(L.lib().unet_head_fwd(_DT[x.dtype], _ptr(x), n, h, w, ci, _ptr(weight), _ptr(bias), co,
                                      int(sigmoid), _ptr(out), _stream()), "unet_head_fwd")
        ctx.save_for_backward(x, weight, out)
        ctx.sigmoid = bool(sigmoid)
        ctx.keys = (weight.data_ptr(), bias.data_ptr())
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        n, ci, h, w = x.shape
        co = weight.shape[0]
        lib, dev = L.lib(), x.device
        dout = dout.contiguous().float()
        dx = _nhwc_empty(n, ci, h, w, x.dtype, dev)
        dw = grad_out(weight.shape, dev, ctx.keys[0])
        db = grad_out((co,), dev, ctx.keys[1])
        ws = _workspace(lib.unet_head_bwd_workspace(n, h, w, ci, co), dev)
        L.check(lib.unet_head_bwd(_DT[x.dtype], _ptr(x), _ptr(out), _ptr(dout), n, h, w, ci, _ptr(weight), co,
                                  int(ctx.sigmoid), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), _stream()),
                "unet_head_bwd")
        return dx, dw, db, None


# ----------------------------------------------------------------------------- losses
class MseFocal(torch.autograd.Function):
    """(MSE(recon, image), focal(amap, mask)) -- /root/reference/src/train_utils.py:23-35."""

    @staticmethod
    def forward(ctx, recon, amap, image, mask, alpha, gamma):
        _require_cuda(recon, amap, image, mask)
        recon, amap = recon.contiguous().float(), amap.contiguous().float()
        image, mask = image.contiguous().float(), mask.contiguous().float()
        if recon.shape != image.shape or amap.shape != mask.shape:
            raise ValueError("CombinedLoss: prediction/target shapes differ")
        lib, dev = L.lib(), recon.device
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        d_recon, d_amap = torch.empty_like(recon), torch.empty_like(amap)
        ws = _workspace(lib.unet_loss_workspace(recon.numel()), dev)
        L.check(lib.unet_loss_mse_focal(_ptr(recon), _ptr(image), recon.numel(), _ptr(amap), _ptr(mask),
                                        amap.numel(), float(alpha), float(gamma), _ptr(losses), _ptr(d_recon),
                                        _ptr(d_amap), _ptr(ws), ws.numel(), _stream()), "unet_loss_mse_focal")
        ctx.save_for_backward(d_recon, d_amap)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_mse, g_focal):
        d_recon, d_amap = ctx.saved_tensors
        return d_recon * g_mse, d_amap * g_focal, None, None, None, None


class Ssim(torch.autograd.Function):
    """SSIMLoss.forward (/root/reference/src/train_utils.py:89-104): a 0-d loss (size_average=True) or one value per
    image (size_average=False, :84-87)."""

    @staticmethod
    def forward(ctx, img1, img2, window_size, size_average=True):
        _require_cuda(img1, img2)
        img1, img2 = img1.contiguous().float(), img2.contiguous().float()
        n, c, h, w = img1.shape
        lib, dev = L.lib(), img1.device
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        d1 = torch.empty_like(img1) if need else None
        d2 = torch.empty_like(img2) if need else None
        ctx.per_image = not size_average
        if size_average:
            loss = torch.empty(1, dtype=torch.float32, device=dev)
            ws = _workspace(lib.unet_ssim_workspace(n * c, h, w), dev)
            L.check(lib.unet_ssim_loss(_ptr(img1), _ptr(img2), n * c, h, w, int(window_size), _ptr(loss), _ptr(d1),
                                       _ptr(d2), _ptr(ws), ws.numel(), _stream()), "unet_ssim_loss")
        else:
            loss = torch.empty(n, dtype=torch.float32, device=dev)
            ws = _workspace(lib.unet_ssim_workspace(c, h, w), dev)
            L.check(lib.unet_ssim_loss_per_image(_ptr(img1), _ptr(img2), n, c, h, w, int(window_size), _ptr(loss),
                                                 _ptr(d1), _ptr(d2), _ptr(ws), ws.numel(), _stream()),
                    "unet_ssim_loss_per_image")
        if need:
            ctx.save_for_backward(d1, d2)
        return loss[0] if size_average else loss

    @staticmethod
    def backward(ctx, g):
        d1, d2 = ctx.saved_tensors
        if ctx.per_image:
            g = g.reshape(-1, 1, 1, 1)
        return d1 * g, d2 * g, None, None


def threshold_confusion(pred, truth, thresholds, select=None, counts=None):
    """Confusion counts [K][4] = {tp, fp, fn, tn} (int64 DEVICE tensor; pass it back as ``counts`` to keep accumulating
    over batches, read it once at the end) of (pred > t) against (truth > 0.5) for each threshold, over the images with
    select[n] true (None: all): the pixel-metric epilogue of /root/reference/src/test.py:79-106 and
    src/train_utils.py:232-245 on the device (unet_threshold_confusion)."""
    _require_cuda(pred)
    p = pred.detach().contiguous().float()
    t = truth.detach().to(p.device).contiguous().float()
    if p.shape != t.shape or p.dim() < 2:
        raise ValueError("threshold_confusion: prediction / truth shapes differ")
    n = p.shape[0]
    per = p.numel() // n
    values = [float(v) for v in thresholds]
    if not values:
        raise ValueError("threshold_confusion: at least one threshold")
    sel = None if select is None else torch.as_tensor(select).to(device=p.device, dtype=torch.uint8).contiguous()
    if counts is None:
        counts = torch.zeros((len(values), 4), dtype=torch.int64, device=p.device)
    if tuple(counts.shape) != (len(values), 4):
        raise ValueError("threshold_confusion: counts must be [len(thresholds), 4]")
    for k in range(0, len(values), 8):           # the kernel takes up to 8 thresholds per pass over the maps
        thr = torch.tensor(values[k:k + 8], dtype=torch.float32, device=p.device)
        L.check(L.lib().unet_threshold_confusion(_ptr(p), _ptr(t), _ptr(sel), n, per, _ptr(thr), thr.numel(),
                                                 _ptr(counts[k:k + 8]), _stream()), "unet_threshold_confusion")
    return counts


def preprocess_u8(images_u8, flips=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """uint8 [N, H, W, 3] device batch -> normalised fp32 NCHW (ToTensor + Normalize, optional per-sample horizontal
    flip): /root/reference/src/dataset.py:134-146, src/kolektorsdd_dataset.py:133-150, on the GPU."""
    _require_cuda(images_u8)
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3:
        raise ValueError("preprocess_u8 expects a uint8 [N, H, W, 3] tensor")
    images_u8 = images_u8.contiguous()
    n, h, w, _ = images_u8.shape
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=images_u8.device)
    fl = None
    if flips is not None:
        fl = flips.to(device=images_u8.device, dtype=torch.uint8).contiguous()
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s_ = (C.c_float * 3)(*[float(v) for v in std])
    L.check(L.lib().unet_preprocess_u8(_ptr(images_u8), _ptr(fl), _ptr(out), n, h, w, m, s_, _stream()),
            "unet_preprocess_u8")
    return out


# ----------------------------------------------------------------------------- profiling / optimiser
def prof_enable(on: bool) -> None:
    L.check(L.lib().unet_prof_enable(int(on)), "unet_prof_enable")


def prof_collect():
    ms = (C.c_double * L.K_COUNT)()
    launches = (C.c_int64 * L.K_COUNT)()
    flops = (C.c_double * L.K_COUNT)()
    L.check(L.lib().unet_prof_collect(ms, launches, flops), "unet_prof_collect")
    return {L.KCLASS_NAMES[i]: {"ms": ms[i], "launches": launches[i], "flops": flops[i]}
            for i in range(L.K_COUNT)}


def prof_kernels():
    """Per-kernel breakdown of the brackets consumed by the last prof_collect(): {kernel name: ms, launches, flops}."""
    out, i = {}, 0
    lib = L.lib()
    while True:
        name, ms, n, fl = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double()
        if lib.unet_prof_kernel_stats(i, C.byref(name), C.byref(ms), C.byref(n), C.byref(fl)) != 0:
            break
        by = C.c_double()
        lib.unet_prof_kernel_bytes(i, C.byref(by))
        out[name.value.decode()] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        i += 1
    return out


def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0):
    """In-place fused Adam over flat fp32 arenas (torch.optim.Adam semantics, train_utils.py:266)."""
    _require_cuda(param, grad, exp_avg, exp_avg_sq)
    L.check(L.lib().unet_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(),
                                   lr, beta1, beta2, eps, weight_decay, grad_scale, int(step), _stream()),
            "unet_adam_step")
