"""GAN training step of the reference LightningModule, without Lightning.

``bSSFPToDWITensorModel`` mirrors the attribute and method names of the reference class
(src/model.py:141-213, 259-281, 359-361) so that the parity tests read like the reference:
``gen``, ``discr``, ``unpack_batch``, ``_gen_step``, ``_discr_step``, ``compute_recon_loss``,
``training_step``, ``configure_optimizers``.  What Lightning did implicitly is explicit here:

* ``toggle_optimizer`` / ``untoggle_optimizer``  -> ``requires_grad_`` on the other network
* ``manual_backward``                             -> ``loss.backward()``
* ``self.log(..., sync_dist=True)`` x6            -> ONE stacked tensor, reduced once per step
  by the caller (no host sync inside the step)
* DDP gradient all-reduce                         -> ``GradSync`` hooks (``ddp.py``), one per network

The Perceptual term (src/model.py:127-129) needs remotely fetched MedicalNet weights; it is a
pluggable slot (``extra_recon_terms``) and absent by default, as stated wherever numbers are
reported.  The module is agnostic of where ``gen``/``discr`` come from, so the same step logic
drives the CPU oracle modules in the tests.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

DATA = "data"  # tio.DATA


def dist_world(group=None) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

# the reference's log names (src/model.py:177, 204-210, 266, 276): EarlyStopping monitors 'val_gen_loss_recon' (src/train.py:19)
LOG_KEYS = ("train_gen_loss_adversarial", "train_gen_loss_recon_L1", "train_gen_loss_recon", "train_gen_loss",
            "train_discr_loss")


class bSSFPToDWITensorModel(nn.Module):
    def __init__(self, input_modality, lr=1e-3, batch_size=8, perceptual_factor=1e3, recon_factor=1e2,
                 gen: Optional[nn.Module] = None, discr: Optional[nn.Module] = None,
                 l1_fn: Optional[Callable] = None, optimizer_class=None,
                 extra_recon_terms: Optional[Dict[str, Callable]] = None, recon_divisor: Optional[int] = None,
                 reference_quirks: bool = False):
        super().__init__()
        self.input_modality = input_modality
        if gen is None or discr is None:
            from .nn import Discriminator, Generator
            gen = Generator(input_modality) if gen is None else gen
            discr = Discriminator(input_modality) if discr is None else discr
        self.gen, self.discr = gen, discr
        self.recon_factor, self.lr, self.batch_size = recon_factor, lr, batch_size
        self.perceptual_factor = perceptual_factor
        self.l1_fn = l1_fn
        self.extra_recon_terms = extra_recon_terms or {}
        self.optimizer_class = optimizer_class
        # src/model.py:209 divides the summed terms by their NUMBER (L1 and Perceptual: 2).  None = the number of terms
        # present here (1 without the Perceptual slot: recon = 100 * L1); 2 reproduces the reference's weighting of the
        # L1 term (50 * L1) while the Perceptual term is absent.
        self.recon_divisor = recon_divisor
        # True: test_step feeds compute_metrics what the reference feeds it (src/model.py:303-307: the aggregated INPUT
        # volume under the name pred_tensor) instead of the aggregated prediction
        self.reference_quirks = reference_quirks
        self.grad_sync_gen = None      # set by ddp.attach(); called after each phase's backward
        self.grad_sync_discr = None
        self.sinks_gen = None          # gradsink.GradBuckets of the HIP networks (created at the first training step)
        self.sinks_discr = None
        self.use_grad_sinks = True
        self._stage = None
        self._optimizers = None
        self.last_logs: Dict[str, torch.Tensor] = {}

    # ------------------------------------------------------------------ reference API
    def forward(self, x):
        return self.gen(x)

    def configure_optimizers(self):
        cls = self.optimizer_class
        if cls is None:
            from .optim import FusedAdamW
            cls = FusedAdamW if next(self.gen.parameters()).is_cuda else torch.optim.AdamW
        return cls(self.gen.parameters(), lr=self.lr), cls(self.discr.parameters(), lr=self.lr)

    def optimizers(self):
        if self._optimizers is None:
            self._optimizers = self.configure_optimizers()
        return self._optimizers

    def unpack_batch(self, batch, test=False):
        x = batch[self.input_modality][DATA]
        y = batch["dwi-tensor" if test else "dwi-tensor_orig"][DATA]
        return x, y

    def _l1(self, a, b):
        if self.l1_fn is not None:
            return self.l1_fn(a, b)
        if a.is_cuda:
            from .functional import l1_loss
            return l1_loss(a, b)
        return F.l1_loss(a, b)

    def compute_recon_loss(self, y_hat, y, logs, prefix):
        terms = OrderedDict(L1=self._l1(y_hat, y))
        for name, fn in self.extra_recon_terms.items():
            terms[name] = fn(y_hat, y)
        total = None
        for name, t in terms.items():
            logs[f"{prefix}_loss_recon_{name}"] = t.detach()
            total = t if total is None else total + t
        total = total / (self.recon_divisor or len(terms)) * self.recon_factor
        logs[f"{prefix}_loss_recon"] = total.detach()
        return total

    fused_loss_heads = True             # HIP path: the loss heads as one launch each way (functional.GanGenLossFn / GanDiscrLossFn)

    def _fused_losses(self, *tensors) -> bool:
        """The default loss configuration on device f32 tensors: BCEWithLogits + L1 and their arithmetic as single launches
        (same formulas; ~45 scalar-sized torch launches per step otherwise, ~5 us each in a graph replay)."""
        return (self.fused_loss_heads and self.l1_fn is None and not self.extra_recon_terms
                and all(t.is_cuda and t.dtype == torch.float32 for t in tensors))

    def _gen_for_discr(self, x):
        """``self.gen(x)`` on its way into the discriminator: the HIP generator then also hands over S(output) (Fn.GenOutFn)"""
        if hasattr(self.gen, "emit_s2d") and next(self.gen.parameters()).is_cuda:
            self.gen.emit_s2d = True
            try:
                return self.gen(x)
            finally:
                self.gen.emit_s2d = False
        return self.gen(x)

    @staticmethod
    def _detach(y):
        """``y.detach()`` that keeps the space-to-depth companion of a generator output"""
        out = y.detach()
        s = getattr(y, "_mi355_s2d", None)
        if s is not None:
            out._mi355_s2d = s.detach()
        return out

    def _gen_step(self, x, y, logs, step_name="train"):
        y_hat = self._gen_for_discr(x)
        logits = self.discr(x, y_hat)
        if self._fused_losses(logits, y_hat, y):
            from .functional import GanGenLossFn
            total, parts = GanGenLossFn.apply(logits, y_hat, y, float(self.recon_divisor or 1), float(self.recon_factor))
            logs[f"{step_name}_gen_loss_recon_L1"] = parts[0]
            logs[f"{step_name}_gen_loss_recon"] = parts[1]
            logs[f"{step_name}_gen_loss_adversarial"] = parts[2]
            return total, y_hat
        adv = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
        recon = self.compute_recon_loss(y_hat, y, logs, step_name + "_gen")
        logs[f"{step_name}_gen_loss_adversarial"] = adv.detach()
        return adv + recon, y_hat

    pair_discriminator_calls = True     # HIP discriminator: its two calls of the discriminator phase as one stacked pass

    def _discr_pair(self) -> bool:
        return self.pair_discriminator_calls and hasattr(self.discr, "forward_pair") and next(self.discr.parameters()).is_cuda

    def _discr_uses(self, x, y) -> int:
        """Gradient contributions every discriminator parameter receives in the discriminator phase: 1 when ``forward_pair``
        takes its stacked pass, 2 when the phase is two calls (extents that are not multiples of 32, inputs that require
        grad, pairing switched off).  Derived from the predicate ``forward_pair`` itself branches on -- announcing 1 while two
        calls ran would let an eagerly attached bucket be all-reduced after the first call's contributions."""
        if self._discr_pair() and self.discr.pair_single_pass(x, x.new_empty(0), y):   # (y_hat is detached: never requires grad)
            return 1
        return 2

    def _discr_step(self, x, y):
        y_hat = self._detach(self._gen_for_discr(x))
        if self._discr_pair():
            both = self.discr.forward_pair(x, y_hat, y, stacked=self._fused_losses(x, y))
            if isinstance(both, torch.Tensor):                      # one pass over both inputs: fake first, then real
                from .functional import GanDiscrLossFn
                return GanDiscrLossFn.apply(both, None)
            logits_hat, logits = both
        else:
            logits_hat = self.discr(x, y_hat)
            logits = self.discr(x, y)
        if self._fused_losses(logits_hat, logits):
            from .functional import GanDiscrLossFn
            return GanDiscrLossFn.apply(logits_hat, logits)
        loss_hat = F.binary_cross_entropy_with_logits(logits_hat, torch.zeros_like(logits_hat))
        loss = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
        return (loss + loss_hat) / 2

    @staticmethod
    def _repack(module: nn.Module):
        """Batched re-pack of the network's packed weights right after its optimiser step (HIP modules only)."""
        if next(module.parameters()).is_cuda:
            from .functional import repack_weights
            repack_weights(module)

    @staticmethod
    def _toggle(module: nn.Module, flag: bool):
        for p in module.parameters():
            p.requires_grad_(flag)

    # ------------------------------------------------------------------ gradient storage / exchange
    def enable_grad_sinks(self, group=None, distributed: bool = False, force_collectives: bool = False) -> bool:
        """HIP networks on the GPU: parameter gradients live in flat per-stage buckets that the gradient kernels write
        in place (gradsink.py).  Bucket 0 = the layers whose gradients are ready first (decoder + bottleneck of the
        U-Net; d3..final of the PatchGAN), bucket 1 = the rest; with world size > 1 a bucket is all-reduced as soon as
        it is complete, under the remaining backward kernels.  ``distributed`` is set by ``ddp.attach`` and
        ``GraphedTrainingStep`` only: a plain ``training_step`` never starts collectives on its own."""
        if self.sinks_gen is not None:
            want = distributed and (dist_world(group) > 1 or force_collectives)
            if not want or self.sinks_gen.exchange:
                return True
            # created local-only by an earlier plain step: rebuild with the exchange switched on
            self.sinks_gen.detach()
            self.sinks_discr.detach()
            self.sinks_gen = self.sinks_discr = None
        if not self.use_grad_sinks:
            return False
        from .nn import Discriminator, Generator, backward_stages
        if not (isinstance(self.gen, Generator) and isinstance(self.discr, Discriminator) and next(self.gen.parameters()).is_cuda):
            return False
        from .gradsink import GradBuckets
        self.sinks_gen = GradBuckets(backward_stages(self.gen, self.input_modality), group, uses_per_phase=1,
                                     distributed=distributed, force_collectives=force_collectives)
        self.sinks_discr = GradBuckets(backward_stages(self.discr, self.input_modality), group, uses_per_phase=2,
                                       distributed=distributed, force_collectives=force_collectives)
        return True

    def _finish_grads(self, which: str):
        sinks = self.sinks_gen if which == "gen" else self.sinks_discr
        sync = self.grad_sync_gen if which == "gen" else self.grad_sync_discr
        if sinks is not None:
            sinks.finish()
        elif sync is not None:
            sync.finish()

    def _backward(self, loss, staged: bool):
        """``manual_backward``.  staged: only the late stage (down to the activations marked by Fn.StageBoundary); the
        early stage follows in ``_backward_early`` -- the caller exchanges the late bucket in between."""
        if not staged:
            with self._side_scope():
                loss.backward()
            return
        from .functional import StageBoundary
        boundary = StageBoundary.end()
        sinks = self._stage_sinks
        late = [p for p in sinks.params[0] if p.requires_grad]
        with self._side_scope():
            grads = torch.autograd.grad([loss], boundary + late, retain_graph=True, allow_unused=True)
        assert all(g is None for g in grads[len(boundary):]), "staged backward needs gradient sinks"
        self._stage = (boundary, list(grads[: len(boundary)]), [p for p in sinks.params[1] if p.requires_grad])

    def _side_scope(self):
        """Weight gradients of the small layers on a side stream for the duration of this backward pass (HIP networks)."""
        import contextlib
        import sys
        fn = sys.modules.get(__package__ + ".functional")
        if fn is None or self.sinks_gen is None:
            return contextlib.nullcontext()
        stack = contextlib.ExitStack()
        stack.enter_context(fn.SideStream.scope())
        stack.enter_context(fn.DeferredReduce.scope())      # slab reductions of the pass batched into a few launches (exits first)
        return stack

    @staticmethod
    def _close_stage_boundary():
        """An exception between StageBoundary.begin() and the staged backward must not leave the process-global collector
        open (every later forward would keep appending -- and keeping alive -- activations)."""
        import sys
        fn = sys.modules.get(__package__ + ".functional")       # (never imported on the CPU-module path: nothing to close)
        if fn is not None and fn.StageBoundary.active():
            fn.StageBoundary.end()

    def _backward_early(self):
        boundary, grads, early = self._stage
        self._stage = None
        with self._side_scope():
            torch.autograd.backward(boundary, grads, inputs=early)

    # The step is split at the points where gradients cross ranks, so that it can run eagerly
    # (training_step) or as hipGraph segments with the collectives in between (GraphedTrainingStep).
    def _phase_gen(self, batch, logs, staged=False):
        """toggle(gen_opt) -> _gen_step -> manual_backward          (src/model.py:264-268)"""
        x, y = self.unpack_batch(batch)
        if x.is_cuda:
            from .functional import DropoutState, StageBoundary
            DropoutState.advance(x.device)
            if self.enable_grad_sinks():
                self.sinks_gen.begin_phase(1)
            if staged:
                self._stage_sinks = self.sinks_gen
                StageBoundary.begin()
        self._toggle(self.discr, False)
        try:
            loss, _ = self._gen_step(x, y, logs)
            logs["train_gen_loss"] = loss.detach()
            self._backward(loss, staged)
        finally:
            self._close_stage_boundary()

    def _update_gen(self):
        """gen_opt.step/zero_grad/untoggle                             (:269-271)"""
        gen_opt, _ = self.optimizers()
        gen_opt.step()
        if self.sinks_gen is None:
            gen_opt.zero_grad()             # (gradient sinks are never zeroed: the next phase's first write overwrites)
        self._repack(self.gen)
        self._toggle(self.discr, True)

    def _phase_discr(self, batch, logs, staged=False):
        """toggle(discr_opt) -> _discr_step -> manual_backward         (:274-278)"""
        x, y = self.unpack_batch(batch)
        self._toggle(self.gen, False)
        if self.sinks_discr is not None:
            self.sinks_discr.begin_phase(self._discr_uses(x, y))             # gradient contributions per parameter in this phase
        if staged:
            from .functional import StageBoundary
            self._stage_sinks = self.sinks_discr
            StageBoundary.begin()
        try:
            loss = self._discr_step(x, y)
            logs["train_discr_loss"] = loss.detach()
            self._backward(loss, staged)
        finally:
            self._close_stage_boundary()

    def _phase_gen_update_discr(self, batch, logs):
        self._update_gen()
        self._phase_discr(batch, logs)

    def _phase_discr_update(self):
        """discr_opt.step/zero_grad/untoggle                          (:279-281)"""
        _, discr_opt = self.optimizers()
        discr_opt.step()
        if self.sinks_discr is None:
            discr_opt.zero_grad()
        self._repack(self.discr)
        self._toggle(self.gen, True)

    def training_step(self, batch, batch_idx=0):
        logs: Dict[str, torch.Tensor] = {}
        self._phase_gen(batch, logs)
        self._finish_grads("gen")
        self._phase_gen_update_discr(batch, logs)
        self._finish_grads("discr")
        self._phase_discr_update()
        self.last_logs = logs
        return None

    def generator_only_step(self, batch, batch_idx=0):
        """BASELINE.json configs[1] ("3D U-Net generator only"): generator forward + backward under the L1 loss + AdamW,
        with the gradient plumbing of ``training_step`` (gradient sinks begin / exchange / finish, no ``zero_grad`` on sink
        parameters), so that an attached model averages its gradients over the ranks here as well."""
        x, y = self.unpack_batch(batch)
        if x.is_cuda:
            from .functional import DropoutState
            DropoutState.advance(x.device)
            if self.enable_grad_sinks():
                self.sinks_gen.begin_phase(1)
        loss = self._l1(self.gen(x), y)
        with self._side_scope():
            loss.backward()
        self._finish_grads("gen")
        gen_opt, _ = self.optimizers()
        gen_opt.step()
        if self.sinks_gen is None:
            gen_opt.zero_grad()
        self._repack(self.gen)
        self.last_logs = {"train_gen_loss_recon_L1": loss.detach()}
        return None

    def compute_metrics(self, y_hat, y, step_name, logs=None):
        """``self.log(f'{step_name}_metric_{name}', metric_fn(y_hat, y).mean())`` for every entry of
        ``metric_fns`` (src/model.py:215-220).  Default list: PSNR(1), SSIM(3-D, data_range 1), L1 on the
        device (``metrics.reference_metric_fns``); the reference's FID entry needs remote weights."""
        if getattr(self, "metric_fns", None) is None:
            from .metrics import reference_metric_fns
            self.metric_fns = reference_metric_fns()
        logs = self.last_logs if logs is None else logs
        for metric_fn, name in self.metric_fns:
            logs[f"{step_name}_metric_{name}"] = metric_fn(y_hat, y).mean()
        return logs

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        """src/model.py:283-289: generator step without an update, ``val_loss`` and the metrics."""
        logs: Dict[str, torch.Tensor] = {}
        x, y = self.unpack_batch(batch)
        loss, y_hat = self._gen_step(x, y, logs, "val")
        logs["val_loss"] = loss.detach()
        self.compute_metrics(y_hat, y, "val", logs)
        self.last_logs = logs
        return loss

    @torch.no_grad()
    def predict_step(self, batch, batch_idx=0, dataloader_idx=None):
        """Grid inference of one subject (src/model.py:314-333): ``batch`` is the reference's
        ``(sampler, i_agg, t_agg, o_agg)`` with the device-side ``inference.GridSampler`` /
        ``GridAggregator``; the patch loop, ``unpack_batch(test=True)`` and the three ``add_batch`` calls
        are the reference's.  Returns all three aggregated volumes named by content -- the reference
        returns the one it calls ``pred_tensor``, which is the aggregated INPUT (``o_agg`` gets ``x``).
        Metrics and NIfTI saving (:327-331) stay with the caller."""
        from .inference import LOCATION, GridPrediction
        sampler, i_agg, t_agg, o_agg = batch
        for patch_batch in sampler.batches(self.batch_size):
            x, y = self.unpack_batch(patch_batch, test=True)
            loc = patch_batch[LOCATION]
            y_hat = self(x)
            i_agg.add_batch(y_hat, loc)
            t_agg.add_batch(y, loc)
            o_agg.add_batch(x, loc)
        return GridPrediction(i_agg.get_output_tensor(), t_agg.get_output_tensor(), o_agg.get_output_tensor())

    @torch.no_grad()
    def test_step(self, batch, batch_idx=0):
        """src/model.py:291-312: the grid loop of ``predict_step`` with ``_gen_step(x, y, 'test')`` per patch batch,
        the summed generator loss logged as ``test_gen_loss_subject`` and the metrics computed on the aggregated
        volumes.  The reference passes its mislabelled ``pred_tensor`` -- the aggregated INPUT -- to
        ``compute_metrics`` (src/model.py:303-307); ``reference_quirks=True`` reproduces exactly that (it needs as many
        input as target channels, or metric functions that accept the mismatch, as in the reference); the default compares
        the aggregated prediction with the aggregated target."""
        from .inference import LOCATION
        sampler, i_agg, t_agg, o_agg = batch
        logs: Dict[str, torch.Tensor] = {}
        tot_loss = None
        for patch_batch in sampler.batches(self.batch_size):
            x, y = self.unpack_batch(patch_batch, test=True)
            loc = patch_batch[LOCATION]
            loss, y_hat = self._gen_step(x, y, logs, "test")
            tot_loss = loss.detach() if tot_loss is None else tot_loss + loss.detach()
            i_agg.add_batch(y_hat, loc)
            t_agg.add_batch(y, loc)
            o_agg.add_batch(x, loc)
        y_hat_vol, y_vol = i_agg.get_output_tensor(), t_agg.get_output_tensor()
        if self.reference_quirks:
            y_hat_vol = o_agg.get_output_tensor()           # what the reference calls pred_tensor
        self.compute_metrics(y_hat_vol.unsqueeze(0), y_vol.unsqueeze(0), "test", logs)
        logs["test_gen_loss_subject"] = tot_loss
        self.last_logs = logs
        return tot_loss

    def stacked_logs(self) -> torch.Tensor:
        """The step's scalars as ONE tensor (order: LOG_KEYS) -- a single all-reduce replaces the
        reference's six ``sync_dist`` logs."""
        return torch.stack([self.last_logs[k].reshape(()).float() for k in LOG_KEYS])


# hipStreamCaptureModeThreadLocal: with a process group alive, c10d's watchdog thread polls the events of earlier
# collectives (hipEventQuery); under the default global mode such a call from ANOTHER thread invalidates a capture in
# progress.  Nothing unsafe is called from the capturing thread itself.
CAPTURE_MODE = "thread_local"


class GraphedTrainingStep:
    """The training step as hipGraph replays (HIP graphs instead of ~600 eager launches per step).

    world_size 1: the whole step is one graph.

    world_size > 1 (src/train.py:30-32 -- DDP's bucketed all-reduce overlapped with backward): the step is FIVE
    graph segments cut where a gradient bucket is complete; each bucket lives in one flat buffer that the gradient
    kernels write in place (gradsink.py: no gather / scatter copies), and its all-reduce is launched asynchronously
    right after the segment that completes it has been enqueued, so RCCL (its own stream, xGMI) runs UNDER the next
    segment's backward kernels:

        [G fwd, D fwd, backward of D + U-Net decoder / bottleneck]   -> all-reduce gen bucket 0 (84 MB)  ...
        [backward of down_2, down_1, conv_0, head (full resolution)] -> all-reduce gen bucket 1          ... wait
        [gen AdamW, G fwd, D fwd x2, backward of final, d5, d4, d3]  -> all-reduce discr bucket 0 (44 MB) ...
        [backward of d2, d1 (the full-resolution PatchGAN layers)]   -> all-reduce discr bucket 1        ... wait
        [discr AdamW]

    The cut points are activations marked during the forward pass (Fn.StageBoundary); the two backward stages are
    ``torch.autograd.grad`` down to them and ``torch.autograd.backward`` from them.

    The batch tensors are static inputs: ``load(batch)`` copies new volumes into them.  Dropout masks
    change per replay (device-side step counter), AdamW bias correction advances on the device.
    """

    def __init__(self, model: bSSFPToDWITensorModel, batch, warmup: int = 3, group=None, force_segments: bool = False,
                 force_collectives: bool = False, broadcast_buffers_every: int = 0):
        import torch.distributed as dist
        self.model = model
        self.batch = batch
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.segmented = self.world > 1 or force_segments or force_collectives   # (force_*: the multi-rank structure on one rank)
        # DDP's broadcast_buffers=True (src/train.py:30): rank 0's BatchNorm running statistics to every rank before every
        # k-th step (0 = never; the buffers are ~8 KB, the broadcast runs between two graph replays)
        self.broadcast_buffers_every = broadcast_buffers_every if self.world > 1 else 0
        self.step_index = 0
        if model.grad_sync_gen is not None or model.grad_sync_discr is not None:
            raise RuntimeError("GraphedTrainingStep does its own gradient exchange: do not ddp.attach() the model")
        if self.segmented:
            if not model.enable_grad_sinks(group, distributed=True, force_collectives=force_collectives):
                raise RuntimeError("GraphedTrainingStep with world size > 1 needs the HIP networks on the GPU")
            model.sinks_gen.auto_launch = model.sinks_discr.auto_launch = False      # launched between the segments
        if self.broadcast_buffers_every:
            from . import ddp
            ddp.flatten_buffers(model)                       # one flat tensor per dtype: the per-step broadcast is one collective each, no copies
        self.launch_log = []                                # (what, position) of the latest step: tests
        for _ in range(max(2, warmup)):                     # eager: allocations, caches, optimiser state
            self._eager_step()
        torch.cuda.synchronize()
        self._assert_fp8_slots_primed()
        self.logs: Dict[str, torch.Tensor] = {}
        self.graphs = self._capture(batch, None, self.logs)
        # further instances of the SAME step over other static input sets (``add_instance``): a feed that alternates between two
        # sets copies the next host batch straight into the set that is not running -- no staging buffer, no device-to-device
        # copy at the step boundary (bench.py --fresh-batch; src/data_module.py:185-188 delivers a new batch every step).
        # Every instance keeps its OWN log tensors (its capture's outputs): ``__call__(i)`` points ``model.last_logs`` at them.
        self.instances = [(batch, self.graphs, self.logs)]
        model.last_logs = self.logs
        torch.cuda.synchronize()

    def _assert_fp8_slots_primed(self):
        """Delayed-scaling slots (functional.Fp8Scales) carry HOST flags that a capture bakes in: a slot that is touched but
        not yet primed would be captured on its first-step path (in-step amax pass) for ever.  The eager warm-up steps
        (at least two) prime every slot the step uses; anything else is a programming error."""
        import sys
        fn = sys.modules.get(__package__ + ".functional")
        if fn is None:
            return
        for chunks in fn.Fp8Scales._chunks.values():
            for _table, slots, _sat in chunks:
                bad = [i for i, s in enumerate(slots) if s.touched and not s.primed]
                if bad:
                    raise RuntimeError(f"fp8 delayed-scaling slots {bad} are in use but not primed at capture time")

    def _capture(self, batch, pool, logs):
        from .functional import Fp8Scales
        dev = next(self.model.gen.parameters()).device
        primed = Fp8Scales.primed_slots(dev)
        graphs = self._capture_graphs(batch, pool, logs)
        Fp8Scales.check_capture(dev, primed)
        return graphs

    def _capture_graphs(self, batch, pool, logs):
        model = self.model
        if not self.segmented:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=CAPTURE_MODE):
                model._phase_gen(batch, logs)
                model._phase_gen_update_discr(batch, logs)
                model._phase_discr_update()
            return [g]
        gs = [torch.cuda.CUDAGraph() for _ in range(5)]
        with torch.cuda.graph(gs[0], pool=pool, capture_error_mode=CAPTURE_MODE):
            model._phase_gen(batch, logs, staged=True)
        pool = gs[0].pool() if pool is None else pool
        with torch.cuda.graph(gs[1], pool=pool, capture_error_mode=CAPTURE_MODE):
            model._backward_early()
        with torch.cuda.graph(gs[2], pool=pool, capture_error_mode=CAPTURE_MODE):
            model._update_gen()
            model._phase_discr(batch, logs, staged=True)
        with torch.cuda.graph(gs[3], pool=pool, capture_error_mode=CAPTURE_MODE):
            model._backward_early()
        with torch.cuda.graph(gs[4], pool=pool, capture_error_mode=CAPTURE_MODE):
            model._phase_discr_update()
        return gs

    def add_instance(self, batch) -> int:
        """Capture the step once more over another static input set (same shapes); the instances share one memory pool, so
        they must never run concurrently -- they do not: one stream, one replay at a time.  Returns the instance index for
        ``__call__(instance)``."""
        torch.cuda.synchronize()
        logs: Dict[str, torch.Tensor] = {}
        graphs = self._capture(batch, self.graphs[0].pool(), logs)
        self.instances.append((batch, graphs, logs))
        torch.cuda.synchronize()
        return len(self.instances) - 1

    def _segments(self, run_segment, after=None):
        """The multi-rank step: ``run(i)`` executes segment i (eagerly or as a graph replay); ``after(i)`` (optional) is
        called on the host right after segment i has been enqueued (input feeds hook their copies in here)."""
        m = self.model

        def run(i):
            run_segment(i)
            if after is not None:
                after(i)
        self.launch_log = []
        m.sinks_gen.launch_order = []
        m.sinks_discr.launch_order = []
        run(0)
        m.sinks_gen.launch(0); self.launch_log.append(("allreduce gen[0]", "before gen backward stage 2"))
        run(1)
        m.sinks_gen.launch(1)
        m.sinks_gen.finish()
        run(2)
        m.sinks_discr.launch(0); self.launch_log.append(("allreduce discr[0]", "before discr backward stage 2"))
        run(3)
        m.sinks_discr.launch(1)
        m.sinks_discr.finish()
        run(4)

    def _eager_step(self):
        m = self.model
        logs: Dict[str, torch.Tensor] = {}
        if not self.segmented:
            m._phase_gen(self.batch, logs)
            m._phase_gen_update_discr(self.batch, logs)
            m._phase_discr_update()
        else:
            def run(i):
                if i == 0:
                    m._phase_gen(self.batch, logs, staged=True)
                elif i == 1 or i == 3:
                    m._backward_early()
                elif i == 2:
                    m._update_gen()
                    m._phase_discr(self.batch, logs, staged=True)
                else:
                    m._phase_discr_update()
            self._segments(run)
        m.last_logs = logs

    def load(self, batch):
        """Copy a new batch into the static input tensors."""
        for k, v in batch.items():
            if k in self.batch and isinstance(v, dict) and DATA in v:
                self.batch[k][DATA].copy_(v[DATA], non_blocking=True)

    def __call__(self, instance: int = 0, after=None):
        if self.broadcast_buffers_every:
            from . import ddp
            ddp.broadcast_buffers(self.model, every=self.broadcast_buffers_every, step=self.step_index, group=self.group)
        self.step_index += 1
        graphs = self.instances[instance][1]
        if not self.segmented:
            graphs[0].replay()
            if after is not None:
                after(0)
        else:
            self._segments(lambda i: graphs[i].replay(), after)
        self.model.last_logs = self.instances[instance][2]


def synthetic_batch(n: int, s, seed: int, modality: str = "bssfp", device="cpu"):
    """The dict layout ``unpack_batch`` expects (src/model.py:195-199), U[0,1) volumes
    (all modalities are min-max normalised: doc/thesis/03-methods.tex:670)."""
    cin = 24 if modality in ("bssfp", "pc-bssfp") else 6
    if isinstance(s, int):
        s = (s, s, s)
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.rand(n, cin, *s, generator=g)
    y = torch.rand(n, 6, *s, generator=g)
    x, y = x.to(device), y.to(device)
    return {modality: {DATA: x}, "dwi-tensor_orig": {DATA: y}, "dwi-tensor": {DATA: y}}
