"""Gradient storage owned by the path: every parameter's ``.grad`` is a permanent view into ONE flat f32 buffer per
bucket, and the weight-gradient / affine-gradient kernels write their results straight into it.

Why (replaces what DDP's bucket copies did for the reference, src/train.py:30-32):

* the all-reduce payload of a bucket is its flat buffer as it stands -- no per-parameter gather / scatter copies
  (round 1: ~150 tiny copy kernels per phase), and the buffer can be handed to RCCL the moment the bucket's last
  gradient kernel has been enqueued, while the rest of the backward pass still runs;
* autograd's AccumulateGrad does nothing for these parameters (the backward functions return ``None`` for them): no
  per-step allocation of gradient tensors, no add kernels where a layer is used twice in one backward pass (the
  discriminator in ``_discr_step``: the second contribution is accumulated by the kernel itself), and the addresses
  the fused AdamW reads are the same in every step (hipGraph replay).

Protocol: ``begin_phase()`` before a backward pass marks every parameter "fresh" (its first contribution overwrites,
later ones accumulate -- nothing is zeroed); the backward functions ask ``sink_of(param)`` and call ``written(param)``
after enqueuing a contribution; ``finish()`` waits for the exchanges (world size > 1) before the optimiser step.
``.grad`` is therefore what it would be after ``loss.backward()`` with stock modules, averaged over ranks.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def sink_of(p: Optional[torch.Tensor]):
    """The GradBuckets object that owns ``p.grad`` (None: ordinary autograd accumulation)."""
    return getattr(p, "_mi355_sink", None) if p is not None else None


def _join_side_stream():
    import sys
    fn = sys.modules.get(__package__ + ".functional")
    if fn is not None:
        fn.DeferredReduce.flush()                    # pending slab reductions write into the buckets as well
        fn.SideStream.join()


def _flush_deferred_if_completing(sink):
    import sys
    fn = sys.modules.get(__package__ + ".functional")
    if fn is not None and fn.DeferredReduce._after and sink.completes_with(fn.DeferredReduce._after):
        fn.DeferredReduce.flush()


def sink_grad(p: torch.Tensor) -> torch.Tensor:
    """``p.grad`` of a parameter owned by a GradBuckets object -- the permanent view into its bucket.  A caller that ran
    ``optimizer.zero_grad()`` (``set_to_none=True`` is torch's default) after the sinks were enabled has dropped that view:
    fail loudly instead of handing the gradient kernels a null pointer."""
    g = p.grad
    if g is None:
        raise RuntimeError("unet_bssfp_amd: this parameter's gradient lives in a gradient bucket (gradsink.GradBuckets) but "
                           "its .grad view is gone -- do not call zero_grad() on a model whose gradient sinks are enabled "
                           "(the buckets are overwritten by the next backward pass; model.use_grad_sinks = False restores "
                           "plain autograd accumulation)")
    return g


class GradBuckets:
    def __init__(self, buckets: Sequence[Iterable[torch.nn.Parameter]], group=None, uses_per_phase: int = 1,
                 distributed: bool = False, force_collectives: bool = False):
        """``buckets``: parameter lists in the order their gradients become ready (one flat buffer each);
        ``uses_per_phase``: gradient contributions every parameter receives per backward pass (2 for the discriminator
        in the discriminator phase: fake and real batch, src/model.py:185-186).
        ``distributed``: exchange the buckets over ``group`` (only ``ddp.attach`` and ``GraphedTrainingStep`` turn this
        on: a model that was never attached must not start collectives just because a process group exists -- its
        parameters were never broadcast and other ranks may not be training at all).
        ``force_collectives``: issue the all-reduces even on a one-rank group (rehearsal of the RCCL + hipGraph interplay
        on a single GPU: ``bench.py --force-collectives``)."""
        self.group = group
        self.world = dist.get_world_size(group) if (distributed and dist.is_initialized()) else 1
        self.exchange = self.world > 1 or (force_collectives and distributed and dist.is_initialized())
        # RCCL averages inside the collective (ncclAvg); gloo has no AVG: sum, then one scale pass (CPU tests only)
        self._avg = self.exchange and dist.get_backend(group) == "nccl"
        self.uses = uses_per_phase
        self.params: List[List[torch.nn.Parameter]] = []
        self.flat: List[torch.Tensor] = []
        self._where: Dict[int, int] = {}
        self._count: Dict[int, int] = {}
        self._pending: List[int] = []
        self._work: List[Optional[object]] = []
        self.launch_order: List[int] = []            # bucket indices in the order their exchange was launched (tests)
        seen = set()
        for plist in buckets:
            uniq = []
            for p in plist:
                if id(p) not in seen:
                    seen.add(id(p))
                    uniq.append(p)
            if not uniq:
                continue
            dev = uniq[0].device
            flat = torch.zeros(sum(p.numel() for p in uniq), dtype=torch.float32, device=dev)
            off = 0
            for p in uniq:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("GradBuckets expects contiguous f32 parameters")
                p.grad = flat[off: off + p.numel()].view_as(p)
                p._mi355_sink = self
                self._where[id(p)] = len(self.flat)
                self._count[id(p)] = 0
                off += p.numel()
            self.params.append(uniq)
            self.flat.append(flat)
            self._pending.append(0)
            self._work.append(None)

    # ------------------------------------------------------------------ used by the backward functions
    def fresh(self, p) -> bool:
        """True: the next contribution to ``p.grad`` overwrites; False: it accumulates."""
        return self._count[id(p)] == 0

    def written(self, p):
        """A gradient contribution of ``p`` has been enqueued on the current stream."""
        self._count[id(p)] += 1
        if self._count[id(p)] == self.uses:
            b = self._where[id(p)]
            self._pending[b] -= 1
            if self._pending[b] == 0 and self.exchange and self.auto_launch:
                self.launch(b)
            elif self.exchange and self.auto_launch:
                _flush_deferred_if_completing(self)  # only deferred slab reductions may be missing from this bucket now

    def completes_with(self, owed) -> bool:
        """Would the ``written`` calls in ``owed`` [(sink, parameter), ...] complete a bucket that is exchanged eagerly?
        (functional.DeferredReduce: such a bucket's reductions are flushed at once -- its all-reduce must be enqueued while
        backward kernels are still to come, not at the end of the pass.)"""
        if not (self.exchange and self.auto_launch):
            return False
        done = {}
        for sink, p in owed:
            if sink is self and self._count[id(p)] + 1 == self.uses:
                b = self._where[id(p)]
                done[b] = done.get(b, 0) + 1
        return any(self._pending[b] == n and self._work[b] is None for b, n in done.items())

    # ------------------------------------------------------------------ per-phase control
    auto_launch = True      # eager steps: exchange a bucket as soon as its last gradient kernel is enqueued

    def begin_phase(self, uses_per_phase: Optional[int] = None):
        if uses_per_phase is not None:
            self.uses = uses_per_phase
        for b, plist in enumerate(self.params):
            self._pending[b] = len(plist)
            for p in plist:
                self._count[id(p)] = 0
        self.launch_order = []

    def launch(self, b: int):
        """Sum bucket ``b`` over the ranks, asynchronously: RCCL runs on its own stream and first waits for the
        kernels already enqueued on the current stream (torch.distributed semantics); the backward kernels enqueued
        afterwards overlap it."""
        if not self.exchange or self._work[b] is not None:
            return
        _join_side_stream()                          # weight-gradient kernels on the side stream write into this bucket
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._work[b] = dist.all_reduce(self.flat[b], op=op, group=self.group, async_op=True)
        self.launch_order.append(b)

    def finish(self):
        """Every bucket exchanged and averaged; ``.grad`` views are valid for the optimiser step."""
        if not self.exchange:
            return
        for b in range(len(self.flat)):
            if self._work[b] is None:
                self.launch(b)
        inv = 1.0 / self.world
        for b, w in enumerate(self._work):
            w.wait()                                  # stream-level wait with nccl, host wait with gloo
            if not self._avg and self.world > 1:
                self.flat[b].mul_(inv)                # (gloo only: with RCCL the collective itself averages)
            self._work[b] = None

    def complete(self) -> bool:
        """All parameters received all their contributions in this phase (debug / tests)."""
        return all(n == 0 for n in self._pending)

    def detach(self):
        for plist in self.params:
            for p in plist:
                p.grad = None
                if hasattr(p, "_mi355_sink"):
                    del p._mi355_sink


def size_buckets(params: Sequence[torch.nn.Parameter], bucket_mb: float) -> List[List[torch.nn.Parameter]]:
    """Reverse parameter order (gradients of the last layers are ready first), cut every ``bucket_mb`` MiB."""
    cap = int(bucket_mb * 1024 * 1024 / 4)
    out, cur, n = [], [], 0
    seen = set()
    for p in reversed(list(params)):
        if id(p) in seen:
            continue
        seen.add(id(p))
        if cur and n + p.numel() > cap:
            out.append(cur)
            cur, n = [], 0
        cur.append(p)
        n += p.numel()
    if cur:
        out.append(cur)
    return out
