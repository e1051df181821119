"""Fused multi-tensor AdamW on the GPU.

Stands in for ``torch.optim.AdamW(params, lr)`` as used at src/model.py:164, 359-361 (torch
defaults: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01, decoupled decay, bias correction).
State keys follow torch's (``step``, ``exp_avg``, ``exp_avg_sq``) so optimiser state_dicts map.

hipGraph-safe: tensor pointers travel by value in the kernel arguments (no device table, no
pinned-memory staging) and the step count lives in device memory (``_step_dev``), advanced by a
device op -- a captured ``step()`` keeps counting on replay.  The Python-side ``state['step']``
integers are refreshed from it by ``sync_step_counts()``.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_dev = {}          # group index -> int64 device tensor [1]
        self._calls = {}             # group index -> number of step() calls that had gradients

    def _active(self, group) -> List[torch.nn.Parameter]:
        out = []
        for p in group["params"]:
            if p.grad is None:
                continue
            if not p.is_cuda:
                raise _lib.Mi355Error("FusedAdamW runs on the GPU only")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.Mi355Error("FusedAdamW expects contiguous f32 parameters")
            out.append(p)
        return out

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            plist = self._active(group)
            if not plist:
                continue
            dev = plist[0].device
            counter = self._step_dev.get(gi)
            if counter is None:
                counter = torch.zeros(1, dtype=torch.int64, device=dev)
                self._step_dev[gi] = counter
            counter.add_(1)                                   # device op: captured launches keep counting
            self._calls[gi] = self._calls.get(gi, 0) + 1
            keep, ptrs, sizes = [], [], []
            for p in plist:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1
                g = p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.float().contiguous()
                keep.append(g)
                ptrs += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()]
                sizes.append(p.numel())
                torch.autograd.graph.increment_version(p)     # raw-pointer update: packed-weight caches see it
            # parameters that joined later than the others (a gradient appeared for the first time) would
            # need their own counter; the path never does that (unused heads never receive gradients)
            steps = {int(self.state[p]["step"]) for p in plist}
            n = len(plist)
            b1, b2 = group["betas"]
            ptr_arr = (C.c_void_p * (4 * n))(*ptrs)
            size_arr = (C.c_int64 * n)(*sizes)
            if len(steps) == 1 and next(iter(steps)) == self._calls[gi]:
                step_dev, step_host = counter.data_ptr(), 0
            else:                                             # mixed ages: host-side step per sub-list
                step_dev, step_host = None, 0
            if step_dev is not None:
                _lib.check(lib.mi355_adamw_multi(ptr_arr, size_arr, n, group["lr"], b1, b2, group["eps"],
                                                 group["weight_decay"], step_dev, step_host,
                                                 torch.cuda.current_stream().cuda_stream), "adamw_multi")
            else:
                for sv in sorted(steps):
                    idx = [i for i, p in enumerate(plist) if int(self.state[p]["step"]) == sv]
                    pa = (C.c_void_p * (4 * len(idx)))(*[ptrs[4 * i + k] for i in idx for k in range(4)])
                    sa = (C.c_int64 * len(idx))(*[sizes[i] for i in idx])
                    _lib.check(lib.mi355_adamw_multi(pa, sa, len(idx), group["lr"], b1, b2, group["eps"],
                                                     group["weight_decay"], None, sv,
                                                     torch.cuda.current_stream().cuda_stream), "adamw_multi")
        return loss

    def load_state_dict(self, state_dict):
        """torch's AdamW keeps ``step`` as a tensor; here it is a Python int mirrored by a device counter."""
        super().load_state_dict(state_dict)
        self._step_dev.clear()
        self._calls.clear()
        for gi, group in enumerate(self.param_groups):
            steps = set()
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] = int(st["step"].item()) if isinstance(st["step"], torch.Tensor) else int(st["step"])
                    steps.add(st["step"])
                    for k in ("exp_avg", "exp_avg_sq"):
                        st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
            if len(steps) == 1 and next(iter(steps)) > 0:
                t = next(iter(steps))
                dev = next(p.device for p in group["params"])
                self._step_dev[gi] = torch.full((1,), t, dtype=torch.int64, device=dev)
                self._calls[gi] = t

    def sync_step_counts(self):
        """Refresh the Python-side ``state['step']`` from the device counters (after graph replays)."""
        for gi, group in enumerate(self.param_groups):
            counter = self._step_dev.get(gi)
            if counter is None:
                continue
            t = int(counter.item())
            for p in group["params"]:
                if self.state.get(p):
                    self.state[p]["step"] = t
