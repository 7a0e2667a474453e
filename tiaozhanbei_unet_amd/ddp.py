"""Data parallelism for the training hot path: one process per GPU, gradients summed with RCCL
all-reduce over xGMI (``torch.distributed`` backend "nccl" IS RCCL on ROCm), overlapped with backward.

The reference is single-process; this is the one parallel strategy the path needs: images are
independent, only the gradient (43.2 M fp32 = 172.9 MB for AnomalyUNet) is exchanged, once per step.

Design for xGMI (7 point-to-point links per GPU, per-link bound rings): few, LARGE buckets, each one flat fp32
buffer.  The weight-gradient kernels write a parameter's gradient STRAIGHT into its bucket slice (``ops.grad_out``:
autograd adopts the slice as ``param.grad``, so there is no per-tensor copy; a gradient that arrives elsewhere -- a
parameter that already held one -- is copied in by the post-accumulate hook).  When the last slice of a bucket has
landed, its all-reduce is enqueued asynchronously (RCCL's own stream, chained by events to every stream that wrote a
slice: the two decoder branches of AnomalyUNet run on two), so the big decoder/bottleneck buckets travel while the
encoder backward is still running.  ``finish()`` waits for the outstanding collectives before the optimiser step.

Bucket ORDER: the first step uses reverse registration order (~ the order backward produces gradients); the order in
which the hooks actually fired in that step ON RANK 0 is recorded and broadcast, and before the next forward every rank
rebuilds its buckets in it -- a bucket then fills with gradients that complete together (registration order interleaves
the two decoders, which finish on different streams), and all ranks provably share one layout.  BatchNorm statistics
stay per-GPU (standard DDP semantics).
"""
from __future__ import annotations

import weakref
from typing import Iterable, List

import torch
import torch.distributed as dist

from . import ops


DEFAULT_RESERVED_CUS = 0


def configure_overlap(reserved_cus: int = None) -> int:
    """Call BEFORE ``init_process_group`` in a multi-GPU run.  The persistent conv / weight-gradient kernels partition
    their work statically over one block per CU with 125-160 KiB of LDS each; a collective kernel resident on a CU during
    the backward pass (one block per RCCL channel, tens of KiB of LDS) leaves no room for such a block, and the displaced
    blocks of a launch sized for all 256 CUs run as a second round.  So RCCL is capped at ``reserved_cus`` channels
    (``NCCL_MAX_NCHANNELS``, unless the user set it) and the launchers are sized for the remaining CUs
    (``unet_set_reserved_cus``).  The one-GPU probe (tools/cu_share_probe.py, profiles/r03_cu_share_probe.txt) prices the
    trade: workgroups are dealt round-robin to the 8 XCDs and, inside an XCD, to its 4 shader engines of 8 CUs, so ONE
    occupied CU per engine already takes a full engine's worth of blocks out of a launch: 8 .. 32 resident 48-KiB
    workgroups cost +21 % WHILE they are resident with the launchers at 256, 248 or 232 blocks and +8 % at 224 (7 blocks
    per engine -> the only useful reservation is 32); giving up the 32 CUs costs +5 % of EVERY step.  The gradient
    all-reduce of this model (172.9 MB) keeps RCCL kernels resident for an estimated 5-10 % of a step, where the
    unreserved penalty (<= 21 % of that) stays below the reservation's 5 %: the default is therefore 0 (no reservation,
    RCCL's own channel count); ``UNET_DDP_RESERVED_CUS=32`` switches the reservation on for an A/B on a multi-GPU node."""
    import os
    from . import _lib as L
    if reserved_cus is None:
        reserved_cus = int(os.environ.get("UNET_DDP_RESERVED_CUS", DEFAULT_RESERVED_CUS))
    reserved_cus = max(0, int(reserved_cus))
    if reserved_cus:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(reserved_cus))
    if torch.cuda.is_available():
        L.check(L.lib().unet_set_reserved_cus(reserved_cus), "unet_set_reserved_cus")
    return reserved_cus


class GradientExchange:
    """Bucketed, overlapped gradient averaging for ``params`` (any device torch.distributed supports)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 48 << 20, process_group=None,
                 comm_dtype=None):
        """``comm_dtype`` (optional, ``torch.bfloat16``; env ``UNET_DDP_BF16_BUCKETS=1``): the collectives carry a bf16
        copy of each bucket (86 MB instead of 173 MB over xGMI for AnomalyUNet); the fp32 bucket -- what the
        weight-gradient kernels write and the optimiser reads -- is refilled from the reduced copy in ``finish()``.  The
        average is then rounded to 8 mantissa bits; master parameters and optimiser state stay fp32."""
        import os
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if comm_dtype is None and os.environ.get("UNET_DDP_BF16_BUCKETS", "0") != "0":
            comm_dtype = torch.bfloat16
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype: torch.bfloat16 or None")
        self.comm_dtype = None if comm_dtype == torch.float32 else comm_dtype
        self._comm = {}             # bucket index -> low-precision copy in flight
        # rank 0 OF THE GROUP as a global rank (broadcast's ``src`` is global even when ``group`` is given)
        self._src = dist.get_global_rank(process_group, 0) if (dist.is_initialized() and process_group is not None) else 0
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets: List[torch.Tensor] = []
        self._slots = {}            # param -> (bucket index, view)
        self._pending: List[int] = []
        self._count: List[int] = []
        self._handles = []
        self._hooks = []
        self._events = {}
        self._bucket_bytes = bucket_bytes
        self._fired: List[torch.nn.Parameter] = []     # hook order of the current step
        self._reorder = None                           # completion order to rebuild the buckets in (once)
        self._reordered = False
        self.copies = 0                                # gradients that had to be copied into their slice (diagnostic)
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"          # RCCL reduces with AVG in-kernel; gloo needs sum + scale
        self._build(bucket_bytes, list(reversed(self.params)))
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- layout -----------------------------------------------------------------------------
    def _build(self, bucket_bytes: int, order) -> None:
        ops.unregister_grad_slots([p.data_ptr() for p in self._slots])
        self.buckets, self._slots, self._count = [], {}, []
        cur, cur_bytes, plan = [], 0, []
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                plan.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            plan.append(cur)
        for bi, ps in enumerate(plan):
            # every slice starts on a 16-byte boundary
            sizes = [(p.numel() + 3) // 4 * 4 for p in ps]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=ps[0].device)
            off = 0
            for p, sz in zip(ps, sizes):
                self._slots[p] = (bi, flat[off:off + p.numel()].view_as(p))
                off += sz
            self.buckets.append(flat)
            self._count.append(len(ps))
        self._pending = list(self._count)
        if self.world > 1:
            ops.register_grad_slots({p.data_ptr(): (view, weakref.ref(p)) for p, (_, view) in self._slots.items()
                                     if p.is_cuda})

    def maybe_rebuild(self) -> None:
        """Before a forward: re-bucket in the completion order recorded during the first backward (once)."""
        order, self._reorder = self._reorder, None
        if order is None or self._handles:
            return
        old = {p: view for p, (_, view) in self._slots.items()}
        self._build(self._bucket_bytes, order)
        for p, (_, view) in self._slots.items():          # a gradient still held keeps its values, in its new slice
            if p.grad is not None and p.grad.data_ptr() == old[p].data_ptr():
                view.copy_(p.grad)
                p.grad = view
        self._reordered = True

    def bucket_sizes_mb(self):
        return [round(b.numel() * 4 / 2 ** 20, 2) for b in self.buckets]

    # -- start of training: identical replicas ----------------------------------------------------
    def broadcast(self, tensors: Iterable[torch.Tensor], src: int = None) -> None:
        """Make every rank's copy equal to the one on ``src`` (a GLOBAL rank; default: rank 0 of this exchange's group)."""
        if self.world == 1:
            return
        src = self._src if src is None else src
        for t in tensors:
            dist.broadcast(t.data if isinstance(t, torch.nn.Parameter) else t, src=src, group=self.group)

    # -- per step ---------------------------------------------------------------------------------
    def _on_grad(self, p: torch.nn.Parameter) -> None:
        bi, view = self._slots[p]
        ops.release_grad_slot(p.data_ptr())
        if p.grad.data_ptr() != view.data_ptr():          # not written in place (e.g. the parameter already held a gradient)
            view.copy_(p.grad)
            p.grad = view                   # the optimiser reads the reduced bucket slice
            self.copies += 1
        if not self._reordered:
            self._fired.append(p)
        if p.is_cuda:
            # gradients of one bucket may be produced on different streams (the two decoder branches):
            # remember where each slice was written so that the collective waits for all of them
            ev = torch.cuda.Event()
            ev.record()
            self._events.setdefault(bi, []).append(ev)
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi: int) -> None:
        flat = self.buckets[bi]
        if flat.is_cuda:
            cur = torch.cuda.current_stream(flat.device)
            for ev in self._events.pop(bi, []):
                cur.wait_event(ev)
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.comm_dtype is not None:
            flat = self._comm[bi] = flat.to(self.comm_dtype)
        self._handles.append((bi, dist.all_reduce(flat, op=op, group=self.group, async_op=True)))

    def finish(self) -> None:
        """Wait for every outstanding all-reduce (call between backward() and optimizer.step())."""
        if self.world == 1:
            return
        for bi, left in enumerate(self._pending):
            if 0 < left < self._count[bi]:      # some parameter of the bucket got no gradient this step
                self._launch(bi)
        for bi, h in self._handles:
            h.wait()
            if self.comm_dtype is not None:
                self.buckets[bi].copy_(self._comm.pop(bi))
            if not self._avg:
                self.buckets[bi].div_(self.world)
        self._handles.clear()
        self._pending = list(self._count)
        if not self._reordered and self._reorder is None:
            # Every rank must lay its buckets out identically (they all-reduce flat buffers): rank 0's completion order
            # is the one everybody adopts -- the engine's hook order is not guaranteed to agree across ranks (two
            # decoder streams).  One small broadcast, once per run; a rank-0 step that missed a parameter keeps the
            # initial layout everywhere.
            index = {id(p): i for i, p in enumerate(self.params)}
            mine = [index[id(p)] for p in self._fired] if len(self._fired) == len(self.params) else []
            msg = torch.tensor([1 if mine else 0] + (mine or [0] * len(self.params)), dtype=torch.int64,
                               device=self.buckets[0].device)
            dist.broadcast(msg, src=self._src, group=self.group)
            vals = msg.tolist()
            order = [self.params[i] for i in vals[1:]] if vals[0] else None
            if order is not None and sorted(vals[1:]) == list(range(len(self.params))) and \
                    [id(p) for p in order] != [id(p) for ps in self._bucket_plan() for p in ps]:
                self._reorder = order
            else:
                self._reordered = True
        self._fired = []

    def _bucket_plan(self):
        plan = [[] for _ in self.buckets]
        for p, (bi, _) in self._slots.items():
            plan[bi].append(p)
        return plan

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        ops.unregister_grad_slots([p.data_ptr() for p in self._slots])


class DataParallel(torch.nn.Module):
    """Thin wrapper: ``forward`` is the wrapped module's; ``finish_gradients()`` completes the exchange.
    ``state_dict()`` is the wrapped module's (un-prefixed keys)."""

    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 48 << 20, process_group=None, comm_dtype=None):
        super().__init__()
        self.module = module
        self.exchange = GradientExchange(module.parameters(), bucket_bytes, process_group, comm_dtype)
        self.exchange.broadcast(list(module.parameters()) + list(module.buffers()))

    def forward(self, *args, **kwargs):
        self.exchange.maybe_rebuild()
        return self.module(*args, **kwargs)

    def finish_gradients(self) -> None:
        self.exchange.finish()

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)
