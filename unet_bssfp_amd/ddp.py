"""Data-parallel gradient exchange: one process per GPU, bucketed all-reduce (RCCL over xGMI via
``torch.distributed`` backend "nccl"; "gloo" on CPU for the tests) launched from
grad-ready hooks so that it overlaps the remaining backward kernels.

Replaces what Lightning's ``'ddp_find_unused_parameters_true'`` strategy did implicitly for the
reference (src/train.py:30): gradient averaging per phase, initial parameter
broadcast, BatchNorm-buffer broadcast from rank 0, and the six ``sync_dist`` scalar logs (here:
one reduce).  BatchNorm statistics are computed per rank, as in the reference (no SyncBatchNorm); the running
buffers follow DDP's ``broadcast_buffers=True`` through ``broadcast_buffers()`` (rank 0's values, every ``k`` steps).

Buckets are static and filled in reverse parameter order (decoder gradients are ready first).
Parameters that receive no gradient in a phase (the unused modality heads; a frozen network)
simply never fire: ``finish()`` reduces whatever was produced, zero-filling the holes of
partially filled buckets, identically on every rank.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.flat: Optional[torch.Tensor] = None
        self.ready = 0
        self.filled = [False] * len(params)
        self.work = None
        self.redo = False          # a gradient was accumulated again after the bucket had been exchanged


class GradSync:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 25.0, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        plist = [p for p in params]
        # de-duplicate (ModuleDicts share modules) while keeping order
        seen, uniq = set(), []
        for p in plist:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets: List[_Bucket] = []
        cur, cur_n = [], 0
        for p in reversed(uniq):
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(_Bucket(cur))
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        self._handles = []
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[id(p)] = (b, i)
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))

    # ------------------------------------------------------------------ hooks
    def _hook(self, p: torch.nn.Parameter):
        if self.world == 1:
            return
        b, i = self._where[id(p)]
        if b.flat is None or b.flat.device != p.device:
            b.flat = torch.zeros(b.numel, dtype=torch.float32, device=p.device)
        view = b.flat[b.offsets[i]: b.offsets[i] + p.numel()]
        if b.filled[i]:
            # a second backward() accumulated into this .grad within the phase (gradient accumulation): p.grad holds the
            # running LOCAL sum.  Not yet exchanged: refresh the copy.  Already exchanged: the buffer holds cross-rank sums,
            # so the whole bucket is rebuilt from the local .grad values and exchanged again at finish().
            if b.work is not None:
                b.work.wait()
                b.work, b.redo = None, True
            elif not b.redo:
                view.copy_(p.grad.reshape(-1))
            return
        view.copy_(p.grad.reshape(-1))
        b.filled[i] = True
        b.ready += 1
        if b.ready == len(b.params):
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ------------------------------------------------------------------ per-phase barrier
    def finish(self):
        """Wait for the exchanges and point every ``.grad`` at the averaged values."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b.ready == 0:
                continue
            if b.work is None:                    # partially filled (or to be redone): zero the holes, reduce now
                for i, p in enumerate(b.params):
                    view = b.flat[b.offsets[i]: b.offsets[i] + p.numel()]
                    if not b.filled[i]:
                        view.zero_()
                    elif b.redo:
                        view.copy_(p.grad.reshape(-1))
                self._launch(b)
        inv = 1.0 / self.world
        for b in self.buckets:
            if b.work is None:
                continue
            b.work.wait()                         # stream-level wait on the GPU (no host sync with nccl)
            b.flat.mul_(inv)
            for i, p in enumerate(b.params):
                if b.filled[i]:
                    p.grad = b.flat[b.offsets[i]: b.offsets[i] + p.numel()].view_as(p)
            b.work, b.ready, b.filled, b.redo = None, 0, [False] * len(b.params), False
            b.flat = None                         # grads alias the buffer until zero_grad(); next phase gets a new one

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def broadcast_module_state(module: torch.nn.Module, src: int = 0, group=None, buffers_only: bool = False):
    """Rank ``src``'s parameters and buffers to every rank (one flat broadcast per dtype)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors, seen = [], set()
    items = list(module.buffers()) if buffers_only else list(module.parameters()) + list(module.buffers())
    for t in items:
        if id(t) not in seen:
            seen.add(id(t))
            tensors.append(t)
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    with torch.no_grad():
        for _, ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for t in ts:
                t.copy_(flat[off: off + t.numel()].view_as(t))
                off += t.numel()


def reduce_logs(stacked: torch.Tensor, group=None) -> torch.Tensor:
    """Mean of the stacked step scalars over ranks: one collective instead of six."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stacked, op=dist.ReduceOp.SUM, group=group)
        stacked = stacked / dist.get_world_size(group)
    return stacked


def attach(model, bucket_mb: float = 25.0, group=None, broadcast: bool = True):
    """Wire a ``gan.bSSFPToDWITensorModel`` for data-parallel training."""
    if broadcast:
        broadcast_module_state(model.gen, 0, group)
        broadcast_module_state(model.discr, 0, group)
    modality = getattr(model, "input_modality", None)
    if getattr(model, "enable_grad_sinks", None) is not None and model.enable_grad_sinks(group, distributed=True):
        # HIP networks: the gradient kernels write into flat per-stage buckets (gradsink.py); a bucket is all-reduced the
        # moment its last gradient kernel has been enqueued, under the rest of the backward pass
        return model
    model.grad_sync_gen = GradSync(used_parameters(model.gen, modality), bucket_mb, group)
    model.grad_sync_discr = GradSync(used_parameters(model.discr, modality), bucket_mb, group)
    return model


def flatten_buffers(model):
    """Move every buffer of ``model.gen`` / ``model.discr`` (BatchNorm running statistics, batch counters) into ONE flat tensor
    per dtype and re-register the buffers as views of it: ``broadcast_buffers`` is then one ``dist.broadcast`` per dtype with no
    gather / scatter copies (it was 4 collectives and ~30 small kernels between two graph replays, every step).  Call it once the
    model sits on its device and BEFORE a step is captured (the kernels keep writing through the buffers' pointers);
    ``load_state_dict`` copies in place and keeps the views.  Returns {dtype: flat tensor}."""
    flats = getattr(model, "_flat_buffers", None)
    if flats is not None:
        return flats
    by_dtype, seen = {}, set()
    for net in (model.gen, model.discr):
        for mod in net.modules():
            for name, b in list(mod._buffers.items()):
                if b is not None and id(b) not in seen:
                    seen.add(id(b))
                    by_dtype.setdefault((b.dtype, b.device), []).append((mod, name, b))
    flats = {}
    with torch.no_grad():
        for (dt, dev), items in by_dtype.items():
            flat = torch.empty(sum(b.numel() for _, _, b in items), dtype=dt, device=dev)
            off = 0
            for mod, name, b in items:
                view = flat[off: off + b.numel()].view_as(b)
                view.copy_(b)
                # (a module registered under several names -- the PatchGAN's d1 / blocks -- is visited once; every alias of the
                #  buffer OBJECT inside other modules' tables is re-pointed as well)
                for other in (model.gen, model.discr):
                    for m2 in other.modules():
                        for n2, b2 in m2._buffers.items():
                            if b2 is b:
                                m2._buffers[n2] = view
                off += b.numel()
            flats[(dt, dev)] = flat
    model._flat_buffers = flats
    return flats


def broadcast_buffers(model, every: int = 1, step: int = 0, group=None):
    """DDP's ``broadcast_buffers=True`` (implied by src/train.py:30): rank 0's BatchNorm running statistics to every rank.
    The statistics are computed per rank (no SyncBatchNorm, as in the reference), so without this the eval-mode outputs
    of the generator head and of the PatchGAN drift apart between ranks.  ``every`` = k broadcasts only every k-th step
    (the buffers are ~8 KB; k = 1 reproduces DDP's per-forward broadcast at step granularity).  After ``flatten_buffers`` this is
    one collective per dtype on the flat storage, without copies."""
    if every > 0 and step % every == 0:
        flats = getattr(model, "_flat_buffers", None)
        if flats is not None:
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                for flat in flats.values():
                    dist.broadcast(flat, src=0, group=group)
            return
        broadcast_module_state(model.gen, 0, group, buffers_only=True)
        broadcast_module_state(model.discr, 0, group, buffers_only=True)


def used_parameters(net: torch.nn.Module, modality) -> List[torch.nn.Parameter]:
    """Parameters that receive gradients for ``modality``: the modality heads not selected
    (src/model.py:29-34, 74-78) are excluded statically instead of DDP's dynamic
    ``find_unused_parameters`` (src/train.py:30)."""
    unused = set()
    table = getattr(net, "blocks", None)
    if isinstance(table, torch.nn.ModuleDict) and modality in table:
        head = table[modality]
        for key, m in table.items():
            if key != "unet" and m is not head:
                unused.update(id(p) for p in m.parameters())
    return [p for p in net.parameters() if id(p) not in unused]
