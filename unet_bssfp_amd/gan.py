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

    def _gen_step(self, x, y, logs, step_name="train"):
        y_hat = self.gen(x)
        logits = self.discr(x, y_hat)
        adv = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
        recon = self.compute_recon_loss(y_hat, y, logs, step_name + "_gen")
        logs[f"{step_name}_gen_loss_adversarial"] = adv.detach()
        return adv + recon, y_hat

    def _discr_step(self, x, y):
        y_hat = self.gen(x).detach()
        logits_hat = self.discr(x, y_hat)
        logits = self.discr(x, y)
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

    # The step is split at the two points where gradients cross ranks, so that it can run eagerly
    # (training_step) or as hipGraph segments with the collectives in between (GraphedTrainingStep).
    def _phase_gen(self, batch, logs):
        """toggle(gen_opt) -> _gen_step -> manual_backward          (src/model.py:264-268)"""
        x, y = self.unpack_batch(batch)
        if x.is_cuda:
            from .functional import DropoutState
            DropoutState.advance(x.device)
        self._toggle(self.discr, False)
        loss, _ = self._gen_step(x, y, logs)
        logs["train_gen_loss"] = loss.detach()
        loss.backward()

    def _phase_gen_update_discr(self, batch, logs):
        """gen_opt.step/zero_grad/untoggle -> toggle(discr_opt) -> _discr_step -> manual_backward  (:269-278)"""
        x, y = self.unpack_batch(batch)
        gen_opt, _ = self.optimizers()
        gen_opt.step()
        gen_opt.zero_grad()
        self._repack(self.gen)
        self._toggle(self.discr, True)
        self._toggle(self.gen, False)
        loss = self._discr_step(x, y)
        logs["train_discr_loss"] = loss.detach()
        loss.backward()

    def _phase_discr_update(self):
        """discr_opt.step/zero_grad/untoggle                          (:279-281)"""
        _, discr_opt = self.optimizers()
        discr_opt.step()
        discr_opt.zero_grad()
        self._repack(self.discr)
        self._toggle(self.gen, True)

    def training_step(self, batch, batch_idx=0):
        logs: Dict[str, torch.Tensor] = {}
        self._phase_gen(batch, logs)
        if self.grad_sync_gen is not None:
            self.grad_sync_gen.finish()
        self._phase_gen_update_discr(batch, logs)
        if self.grad_sync_discr is not None:
            self.grad_sync_discr.finish()
        self._phase_discr_update()
        self.last_logs = logs
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


class GraphedTrainingStep:
    """The training step as hipGraph replays (HIP graphs instead of ~1 000 eager launches per step).

    world_size 1: the whole step is one graph.  world_size > 1: three graph segments with the two
    gradient all-reduces (generator, discriminator) issued eagerly between them on flat buffers --
    the payload (90.6 MB + 44.9 MB f32) is ~1 ms on xGMI against a ~20 ms step, so it is not
    overlapped here; the hook-driven overlapped path is ``ddp.GradSync`` (eager mode).

    The batch tensors are static inputs: ``load(batch)`` copies new volumes into them.  Dropout masks
    change per replay (device-side step counter), AdamW bias correction advances on the device.
    """

    def __init__(self, model: bSSFPToDWITensorModel, batch, warmup: int = 3, group=None):
        import torch.distributed as dist
        from .ddp import used_parameters
        self.model = model
        self.batch = batch
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if model.grad_sync_gen is not None or model.grad_sync_discr is not None:
            raise RuntimeError("GraphedTrainingStep does its own gradient exchange: do not ddp.attach() the model")
        self._gparams = used_parameters(model.gen, model.input_modality)
        self._dparams = used_parameters(model.discr, model.input_modality)
        dev = next(model.parameters()).device
        self._flat_g = torch.zeros(sum(p.numel() for p in self._gparams), device=dev) if self.world > 1 else None
        self._flat_d = torch.zeros(sum(p.numel() for p in self._dparams), device=dev) if self.world > 1 else None
        for _ in range(max(2, warmup)):                     # eager: allocations, caches, optimiser state
            self._eager_step()
        torch.cuda.synchronize()
        self.graphs = []
        logs: Dict[str, torch.Tensor] = {}
        if self.world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                model._phase_gen(batch, logs)
                model._phase_gen_update_discr(batch, logs)
                model._phase_discr_update()
            self.graphs = [g]
        else:
            g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                model._phase_gen(batch, logs)
                self._gather(self._gparams, self._flat_g)
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._scatter(self._gparams, self._flat_g)
                model._phase_gen_update_discr(batch, logs)
                self._gather(self._dparams, self._flat_d)
            with torch.cuda.graph(g3, pool=g1.pool()):
                self._scatter(self._dparams, self._flat_d)
                model._phase_discr_update()
            self.graphs = [g1, g2, g3]
        model.last_logs = logs
        torch.cuda.synchronize()

    # ---- flat gradient buffers (world > 1)
    @staticmethod
    def _gather(params, flat):
        off = 0
        for p in params:
            n = p.numel()
            if p.grad is not None:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            else:
                flat[off:off + n].zero_()
            off += n

    def _scatter(self, params, flat):
        flat.mul_(1.0 / self.world)
        off = 0
        for p in params:
            n = p.numel()
            p.grad = flat[off:off + n].view_as(p)
            off += n

    def _reduce(self, flat):
        import torch.distributed as dist
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def _eager_step(self):
        m = self.model
        logs: Dict[str, torch.Tensor] = {}
        m._phase_gen(self.batch, logs)
        if self.world > 1:
            self._gather(self._gparams, self._flat_g)
            self._reduce(self._flat_g)
            self._scatter(self._gparams, self._flat_g)
        m._phase_gen_update_discr(self.batch, logs)
        if self.world > 1:
            self._gather(self._dparams, self._flat_d)
            self._reduce(self._flat_d)
            self._scatter(self._dparams, self._flat_d)
        m._phase_discr_update()
        m.last_logs = logs

    def load(self, batch):
        """Copy a new batch into the static input tensors."""
        for k, v in batch.items():
            if k in self.batch and isinstance(v, dict) and DATA in v:
                self.batch[k][DATA].copy_(v[DATA], non_blocking=True)

    def __call__(self):
        if self.world == 1:
            self.graphs[0].replay()
            return
        g1, g2, g3 = self.graphs
        g1.replay()
        self._reduce(self._flat_g)
        g2.replay()
        self._reduce(self._flat_d)
        g3.replay()


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
