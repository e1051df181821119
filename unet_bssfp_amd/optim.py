"""Fused multi-tensor AdamW on the GPU (one launch per optimiser step).

Stands in for ``torch.optim.AdamW(params, lr)`` as used at src/model.py:164, 359-361 (torch
defaults: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01, decoupled decay, bias correction).
State keys follow torch's (``step``, ``exp_avg``, ``exp_avg_sq``) so optimiser state_dicts map.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import _lib, functional


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            by_step: Dict[int, List[torch.nn.Parameter]] = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise _lib.Mi355Error("FusedAdamW runs on the GPU only")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.Mi355Error("FusedAdamW expects contiguous f32 parameters")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] = int(st["step"]) + 1
                by_step.setdefault(st["step"], []).append(p)
            for step, plist in by_step.items():
                ptrs, sizes = [], []
                keep = []
                for p in plist:
                    g = p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.float().contiguous()
                    keep.append(g)
                    st = self.state[p]
                    ptrs += [p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()]
                    sizes.append(p.numel())
                dev = plist[0].device
                table = torch.tensor(ptrs + sizes, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
                n = len(plist)
                b1, b2 = group["betas"]
                _lib.check(lib.mi355_adamw_multi(table.data_ptr(), table.data_ptr() + 8 * 4 * n, n, max(sizes),
                                                 group["lr"], b1, b2, group["eps"], group["weight_decay"], step,
                                                 torch.cuda.current_stream().cuda_stream), "adamw_multi")
        functional.bump_weight_epoch()      # raw-pointer update: invalidate the packed-weight caches
        return loss
