"""Adam / AdamW of the reference's ``get_optimizer`` (/root/reference/src/train_utils.py:263-270) as ONE kernel launch
per step over every parameter tensor (libunet_hip's unet_adam_multi), instead of torch's per-tensor walk -- SURVEY 8(f-2).

Same update rule and the same ``state_dict`` layout as ``torch.optim.Adam`` / ``AdamW`` (per-parameter ``step``,
``exp_avg``, ``exp_avg_sq``; param_groups with lr / betas / eps / weight_decay), so checkpoints written by either load into
the other and the lr schedulers of ``get_scheduler`` drive it unchanged.  ``grad_scale`` (default 1) multiplies every
gradient inside the kernel: data parallelism folds its 1/world there when the collective sums instead of averaging.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                      decoupled=bool(decoupled)))
        self.grad_scale = 1.0
        self._tables = {}          # group index -> cached descriptor / chunk tables

    def _normalise_loaded(self):
        """After ``load_state_dict`` / unpickling: a checkpoint written by ``torch.optim.Adam`` / ``AdamW`` (the reference's
        optimisers, src/train_utils.py:266-268) carries no ``decoupled`` key -- torch names it ``decoupled_weight_decay`` --
        and, loaded with ``map_location=device`` (src/utils.py:52), its per-parameter ``step`` scalars sit on the GPU:
        the update rule is taken from torch's key and every ``step`` goes back to a host fp32 scalar (reading a device
        scalar would be one host sync per parameter per step)."""
        for group in self.param_groups:
            # torch < 2.7 wrote AdamW checkpoints without ``decoupled_weight_decay``: then the rule this optimiser was
            # constructed with (get_optimizer's --optimizer choice) stands
            group.setdefault("decoupled", bool(group.get("decoupled_weight_decay", self.defaults.get("decoupled", False))))
        for st in self.state.values():
            if "step" in st:
                v = st["step"]
                st["step"] = torch.tensor(float(v), dtype=torch.float32)       # (one sync per tensor, once per load)
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._normalise_loaded()

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("grad_scale", 1.0)
        self._normalise_loaded()

    def _init_state(self, p):
        st = self.state[p]
        if not st:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)            # host scalar, like torch's default
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _table(self, gi, params, states):
        """Device descriptor table [n][5 x int64] (+ the chunk table, which depends on the sizes only)."""
        ptrs = tuple(v for p, st in zip(params, states)
                     for v in (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
        sizes = tuple(p.numel() for p in params)
        dev = params[0].device
        tab = self._tables.get(gi)
        if tab is None or tab["sizes"] != sizes or tab["device"] != dev:
            chunk = L.lib().unet_adam_chunk_elems()
            rows = [(t, 0, first) for t, n in enumerate(sizes) for first in range(0, n, chunk)]
            ck = np.zeros(len(rows), dtype=[("tensor", "<i4"), ("reserved", "<i4"), ("first", "<i8")])
            for i, r in enumerate(rows):
                ck[i] = r
            tab = {"sizes": sizes, "device": dev, "ptrs": None, "n_chunks": len(rows),
                   "chunks": torch.from_numpy(ck.view(np.uint8).copy()).to(dev),
                   # a ring of pinned staging buffers: one is rewritten only after the copy that last read it is done
                   "host": [torch.empty((len(sizes), 5), dtype=torch.int64).pin_memory() for _ in range(4)],
                   "host_done": [None] * 4, "host_i": 0,
                   "descs": torch.empty((len(sizes), 5), dtype=torch.int64, device=dev)}
            self._tables[gi] = tab
        if tab["ptrs"] != ptrs:
            i = tab["host_i"] = (tab["host_i"] + 1) % 4
            h = tab["host"][i]
            if tab["host_done"][i] is not None:
                tab["host_done"][i].synchronize()
            rows = h.numpy()
            rows[:, :4] = np.asarray(ptrs, dtype=np.int64).reshape(-1, 4)
            rows[:, 4] = sizes
            tab["descs"].copy_(h, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            tab["host_done"][i] = ev
            tab["ptrs"] = ptrs
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("FusedAdam handles dense fp32 parameters on the GPU (libunet_hip.so) only")
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("FusedAdam needs contiguous parameters and gradients")
            states = [self._init_state(p) for p in params]
            steps = {int(st["step"]) for st in states}
            if len(steps) != 1:
                raise RuntimeError("FusedAdam: parameters of one group are at different steps")
            step = steps.pop() + 1
            tab = self._table(gi, params, states)
            b1, b2 = group["betas"]
            st = C.c_void_p(torch.cuda.current_stream(params[0].device).cuda_stream)
            L.check(L.lib().unet_adam_multi(C.c_void_p(tab["descs"].data_ptr()), C.c_void_p(tab["chunks"].data_ptr()),
                                            tab["n_chunks"], float(group["lr"]), float(b1), float(b2),
                                            float(group["eps"]), float(group["weight_decay"]), float(self.grad_scale),
                                            step, int(group["decoupled"]), st), "unet_adam_multi")
            for s_ in states:
                s_["step"] += 1
        return loss
