"""Drop-in ``torch.nn.Module``s for the three construction sites of the reference hot path
(SURVEY.md 8(b)): same constructor signatures, same ``state_dict`` keys and shapes, f32
``nn.Parameter`` leaves, but every FLOP runs in the hand-written gfx950 kernels behind the C ABI.

* ``DownSampleConv``  <- src/model.py:42-65
* ``Discriminator``   <- src/model.py:68-92
* ``BasicUNet``       <- monai.networks.nets.BasicUNet as constructed at src/model.py:22-28
* ``Generator``       <- src/model.py:15-39

Public ``forward`` takes / returns logical NCDHW tensors like the reference modules.  Between our
own modules activations travel as NDHWC buffers (``forward_act``); a tensor returned by ``forward``
is an NCDHW *view* of such a buffer and carries it along (``_mi355_act``), so a reference-side
``Generator.forward`` that chains head -> unet pays no layout conversion.

GPU only: there is no CPU path here (the CPU restatement lives in ``oracle/`` and is test-only).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .ops import round_up

_DEFAULT_DTYPE = torch.float32


def set_default_compute_dtype(dtype: torch.dtype):
    global _DEFAULT_DTYPE
    assert dtype in (torch.float32, torch.bfloat16)
    _DEFAULT_DTYPE = dtype


FP8 = "fp8"     # bf16 storage + e4m3 operands for the full-resolution 3x3x3 convolutions (BASELINE.json configs[4])


def compute_dtype_from_name(name):
    """'f32' (parity mode), 'bf16' (throughput mode), 'fp8' (bf16 + e4m3 convolutions where they pay) -> what
    ``set_compute_dtype`` takes."""
    if isinstance(name, torch.dtype) or name == FP8:
        return name
    table = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
    if name not in table:
        raise ValueError(f"unknown compute dtype {name!r} (choose from {sorted(table)})")
    return table[name]


def set_compute_dtype(module: nn.Module, dtype: torch.dtype) -> nn.Module:
    """Arithmetic/storage type of the activations (f32: parity mode, bf16: throughput mode)."""
    dtype = compute_dtype_from_name(dtype)
    fp8 = dtype == FP8
    if fp8:
        dtype = torch.bfloat16
    assert dtype in (torch.float32, torch.bfloat16)
    for m in module.modules():
        if isinstance(m, _Mi355Module):
            m.compute_dtype = dtype
        if isinstance(m, Conv3d):
            # e4m3 operands (per-tensor scales, f32 accumulate) for forward and data gradient of the 3x3x3 stride-1 layers
            # the fp8 kernel covers (32 input channels at full resolution: decided per call); weight gradients, statistics,
            # master weights and every other layer stay as in the bf16 mode
            m.fp8 = fp8 and m.kernel_size == (3, 3, 3) and m.stride == (1, 1, 1) and m.padding == (1, 1, 1)
    return module


class _Mi355Module(nn.Module):
    def __init__(self):
        super().__init__()
        self.compute_dtype = _DEFAULT_DTYPE

    # ---- boundary helpers ----------------------------------------------------------------
    def _to_act(self, x: torch.Tensor, cp: Optional[int] = None) -> torch.Tensor:
        ops.require_cuda(x)
        act = getattr(x, "_mi355_act", None)
        if act is not None and act.dtype == self.compute_dtype and (cp is None or act.shape[4] == cp):
            return act
        if x.dim() != 5:
            raise ValueError(f"expected a 5-D (N,C,D,H,W) tensor, got {tuple(x.shape)}")
        cp = round_up(x.shape[1], 16) if cp is None else cp
        if x.requires_grad or torch.is_grad_enabled() and x.grad_fn is not None:
            return Fn.PackFn.apply(cp, self.compute_dtype, x)
        # a constant input (the batch's x) entering the path more than once in a training step -- generator phase and
        # discriminator phase -- is packed once; the memo is emptied at the start of every step (Fn.PackMemo)
        hit = Fn.PackMemo.get(x, cp, self.compute_dtype)
        if hit is None:
            hit = Fn.PackFn.apply(cp, self.compute_dtype, x)
            Fn.PackMemo.put(x, cp, self.compute_dtype, hit)
        return hit

    @staticmethod
    def _from_act(act: torch.Tensor, c: int) -> torch.Tensor:
        v = act.permute(0, 4, 1, 2, 3)
        if c != act.shape[4]:
            v = v[:, :c]
        v._mi355_act = act
        return v


# ------------------------------------------------------------------------------------------
class Conv3d(_Mi355Module):
    """torch.nn.Conv3d(in, out, kernel_size, stride, padding, bias) -- cubic kernels 1..4."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        s = stride if isinstance(stride, int) else stride[0]
        p = padding if isinstance(padding, int) else padding[0]
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = (k,) * 3, (s,) * 3, (p,) * 3
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k, k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.spec = Fn.ConvSpec("conv", in_channels, out_channels, k, s, p)
        self.fp8 = False
        self.reset_parameters()

    def reset_parameters(self):
        # identical to torch.nn.modules.conv._ConvNd.reset_parameters (same RNG consumption)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward_act(self, x0, x1=None, want_stats=False, zero_bias_grad=False, s2d_cp=0, lazy_dx=False):
        return Fn.ConvFn.apply(x0, x1, self.weight, self.bias, self.spec, want_stats, zero_bias_grad, s2d_cp, self.fp8, lazy_dx)

    def forward(self, x):
        z, _ = self.forward_act(self._to_act(x))
        return self._from_act(z, self.out_channels)


class ConvTranspose3d(_Mi355Module):
    """torch.nn.ConvTranspose3d(in, out, kernel_size=2, stride=2, bias) (MONAI UpSample 'deconv')."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, bias=True):
        super().__init__()
        if kernel_size != 2 or stride != 2:
            raise NotImplementedError("only the k2 s2 transposed convolution of BasicUNet is implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 2, 2, 2))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.spec = Fn.ConvSpec("deconv2", in_channels, out_channels, 2, 2, 0)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward_act(self, x):
        return Fn.ConvFn.apply(x, None, self.weight, self.bias, self.spec, False)[0]

    def forward(self, x):
        return self._from_act(self.forward_act(self._to_act(x)), self.out_channels)


class _NormParams(nn.Module):
    """Parameter holder with torch's names: weight, bias (+ BatchNorm buffers)."""

    def __init__(self, channels, batch: bool):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        if batch:
            self.register_buffer("running_mean", torch.zeros(channels))
            self.register_buffer("running_var", torch.ones(channels))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


# ------------------------------------------------------------------------------------------
class DownSampleConv(_Mi355Module):
    """src/model.py:42-65: Conv3d -> [BatchNorm3d] -> [LeakyReLU(0.2)], defaults k4 s2 p1."""

    def __init__(self, in_channels, out_channels, kernel=4, strides=2, padding=1, activation=True, batchnorm=True):
        super().__init__()
        self.activation = activation
        self.batchnorm = batchnorm
        self.conv = Conv3d(in_channels, out_channels, kernel, strides, padding)
        if batchnorm:
            self.bn = _NormParams(out_channels, batch=True)
        self.cfg = Fn.NormCfg("batch" if batchnorm else "none", out_channels, eps=1e-5, momentum=0.1,
                              slope=0.2 if activation else 1.0)

    def forward_act(self, x0, x1=None, s2d_cp=0, s2d_out=False, bn_groups=1, split=None):
        """s2d_cp > 0: x0 is the space-to-depth tensor S(a) with s2d_cp channels per block (k4 s2 p1 only);
        s2d_out: return S(output) for the next k4 s2 p1 block instead of the plain activation;
        bn_groups: BatchNorm statistics per consecutive sample group (Discriminator.forward_pair);
        split = (cx, cy): x0 = S(x) and x1 = S(y) are the two parts of cat([x, y], 1), x constant (Fn.SplitS2dConvFn)."""
        fuse = self.batchnorm and self.training
        n, di, hi, wi = (x1 if split else x0).shape[:4]
        ext = (di - 1, hi - 1, wi - 1) if (s2d_cp or split) else tuple(self.conv.spec.out_extent(e) for e in (di, hi, wi))
        # small outputs (the last PatchGAN blocks): the norm kernel computes the statistics itself, in one launch
        small = fuse and ops.norm_is_small(n, *ext, round_up(self.conv.out_channels, 16), bn_groups)
        if split:
            z, part = Fn.SplitS2dConvFn.apply(x0, x1, self.conv.weight, self.conv.bias, self.conv.spec, split[0], split[1],
                                              fuse and not small)
        else:
            z, part = self.conv.forward_act(x0, x1, want_stats=fuse and not small, zero_bias_grad=fuse, s2d_cp=s2d_cp)
        if not (self.batchnorm or self.activation):
            assert not s2d_out
            return z
        if self.batchnorm:
            # num_batches_tracked is advanced by the statistics kernel (no launch of its own)
            return Fn.NormActFn.apply(z, part if (fuse and not small) else None, self.bn.weight, self.bn.bias, self.conv.bias,
                                      self.cfg, self.training, self.bn.running_mean, self.bn.running_var, s2d_out,
                                      self.bn.num_batches_tracked if self.training else None, small, bn_groups)
        return Fn.NormActFn.apply(z, None, None, None, None, self.cfg, self.training, None, None, s2d_out)

    def forward(self, x):
        return self._from_act(self.forward_act(self._to_act(x)), self.conv.out_channels)


class Discriminator(_Mi355Module):
    """src/model.py:68-92 (PatchGAN): cat(x, y) -> d1[modality] -> d2..d5 -> 1x1x1 conv, raw logits."""

    def __init__(self, modality):
        super().__init__()
        self.modality = modality
        d1_bssfp = DownSampleConv(30, 32, batchnorm=False)
        d1_dwi = DownSampleConv(12, 32, batchnorm=False)
        self.d1 = self.blocks = nn.ModuleDict({"dwi-tensor": d1_dwi, "pc-bssfp": d1_bssfp,
                                               "bssfp": d1_bssfp, "t1w": d1_dwi})
        self.d2 = DownSampleConv(32, 64)
        self.d3 = DownSampleConv(64, 128)
        self.d4 = DownSampleConv(128, 256)
        self.d5 = DownSampleConv(256, 512)
        self.final = Conv3d(512, 1, kernel_size=1)

    def forward(self, x, y):
        ops.require_cuda(x, y)
        cin = x.shape[1] + y.shape[1]
        cp = round_up(cin, 16)
        blocks = (self.d1[self.modality], self.d2, self.d3, self.d4, self.d5)
        if all(e % 32 == 0 for e in x.shape[2:]):
            # k4 s2 p1 == dense k2 s1 on space-to-depth tensors: every block reads S(prev) and writes
            # S(own output), so the whole PatchGAN runs on the stride-1 implicit-GEMM kernels
            split = self._split_first_block(x, y, x.shape[0])
            if split:
                h = self._packed_x(x, split[2])                         # S(x): constant, packed once per training step
                h1 = self._s2d_of(y, split[3])                          # S(y): the part that changes (and may need a gradient)
            else:
                h = Fn.PackFn.apply(-cp, self.compute_dtype, x, y)      # cat([x, y], 1) + layout + s2d
            for i, blk in enumerate(blocks):
                if i == 0 and split:
                    h = blk.forward_act(h, h1, s2d_out=True, split=split[:2])
                else:
                    h = blk.forward_act(h, s2d_cp=cp, s2d_out=i < 4)
                cp = round_up(blk.conv.out_channels, 16)
                if i == 1 and blk.conv.weight.requires_grad:                    # (not while D is frozen in the generator phase)
                    Fn.StageBoundary.mark(h)                            # backward stage cut: {final, d5, d4, d3} | {d2, d1}
        else:
            h = Fn.PackFn.apply(cp, self.compute_dtype, x, y)           # torch.cat([x, y], 1) + layout
            for i, blk in enumerate(blocks):
                h = blk.forward_act(h)
                if i == 1 and blk.conv.weight.requires_grad:
                    Fn.StageBoundary.mark(h)
        z, _ = self.final.forward_act(h)
        return Fn.UnpackFn.apply(z, 1)

    split_first_block = True      # bf16: the first block as x-part (once per step) + y-part (Fn.SplitS2dConvFn)

    def _split_first_block(self, x, y, n_grid):
        """(cx, cy, cp_x, cp_y) when the first block runs split into the parts of cat([x, y], 1) (Fn.SplitS2dConvFn: bf16 mode,
        x constant, both parts on the marching k2 kernel), else None."""
        blk = self.d1[self.modality]
        cx, cy = x.shape[1], y.shape[1]
        if (not self.split_first_block or self.compute_dtype != torch.bfloat16 or x.requires_grad or blk.batchnorm
                or blk.conv.in_channels != cx + cy):
            return None
        cpx, cpy = round_up(cx, 16), round_up(cy, 8)     # (8 blocks of 8 channels: 64 space-to-depth channels for the 6 of y)
        se = tuple(e // 2 + 1 for e in x.shape[2:])
        coutp = round_up(blk.conv.out_channels, 32)
        if not (ops.conv_k2_marches(x.shape[0], se, 8 * cpx, coutp) and ops.conv_k2_marches(n_grid, se, 8 * cpy, coutp)):
            return None
        return cx, cy, cpx, cpy

    def _s2d_of(self, y, cpy):
        """S(y) with cpy channels per block: the tensor the generator produced beside y (Fn.GenOutFn), or a pack of y"""
        s = getattr(y, "_mi355_s2d", None)
        if s is not None and s.dtype == self.compute_dtype and s.shape[4] == 8 * cpy and s.shape[0] == y.shape[0]:
            return s
        return Fn.PackFn.apply(-cpy, self.compute_dtype, y)

    def _packed_x(self, x, cpx):
        hit = Fn.PackMemo.get(x, -cpx, self.compute_dtype)
        if hit is None:
            plain = Fn.PackMemo.get(x, cpx, self.compute_dtype)     # the generator packed the same batch (NDHWC): re-lay it out
            hit = ops.s2d_repack(plain) if plain is not None else Fn.PackFn.apply(-cpx, self.compute_dtype, x.detach())
            Fn.PackMemo.put(x, -cpx, self.compute_dtype, hit)
        return hit

    @staticmethod
    def pair_single_pass(x, y_a, y_b) -> bool:
        """Does ``forward_pair`` take its single stacked pass on these inputs (one gradient contribution per parameter), or
        does it fall back to two calls (two contributions)?  The caller that announces the number of contributions to the
        gradient buckets (gan._phase_discr -> GradBuckets.begin_phase) asks THIS predicate, so the two cannot diverge."""
        return (all(e % 32 == 0 for e in x.shape[2:])
                and not (x.requires_grad or y_a.requires_grad or y_b.requires_grad))

    def forward_pair(self, x, y_a, y_b, stacked: bool = False):
        """(self(x, y_a), self(x, y_b)) in ONE pass over the two inputs stacked along the batch -- the discriminator phase
        calls the network on the fake and on the real batch back to back (src/model.py:184-186).  Every BatchNorm normalises
        each half with its own batch statistics and updates the running statistics in call order (first y_a, then y_b), so
        the results are those of the two separate calls; the weight gradients arrive as one sum over both halves.  Half the
        launches of a phase whose kernels (32^3 x 64 ... 4^3 x 512) cannot fill the chip one call at a time.
        stacked=True: when the single pass is taken, return its (2N, 1, ...) logits as ONE tensor (y_a's first) instead of the
        two halves, so that the loss can hand one gradient tensor back."""
        ops.require_cuda(x, y_a, y_b)
        n = x.shape[0]
        cin = x.shape[1] + y_a.shape[1]
        cp = round_up(cin, 16)
        blocks = (self.d1[self.modality], self.d2, self.d3, self.d4, self.d5)
        if not self.pair_single_pass(x, y_a, y_b):
            return self(x, y_a), self(x, y_b)                       # (general case: two calls)
        d, hh, w = x.shape[2:]
        split = self._split_first_block(x, y_a, 2 * n)
        if split:
            h = self._packed_x(x, split[2])                         # ONE S(x) under both halves of the stacked S(y)
            h1 = Fn._new_s2d(ops.s2d_shape(2 * n, d, hh, w, split[3]), self.compute_dtype, x.device)
            for half, y in enumerate((y_a, y_b)):
                s = getattr(y, "_mi355_s2d", None)
                if s is not None and s.dtype == self.compute_dtype and tuple(s.shape) == tuple(h1[half * n:(half + 1) * n].shape):
                    h1[half * n:(half + 1) * n].copy_(s.detach())   # the generator already produced S(y) (Fn.GenOutFn)
                else:
                    ops.pack_ncdhw_s2d(y.detach().to(torch.float32).contiguous(), h1[half * n:(half + 1) * n], split[3], 0, split[3])
        else:
            h = Fn._new_s2d(ops.s2d_shape(2 * n, d, hh, w, cp), self.compute_dtype, x.device)
            x32 = x.detach().to(torch.float32).contiguous()
            for half, y in enumerate((y_a, y_b)):
                ops.pack2(x32, y.detach().to(torch.float32).contiguous(), h[half * n:(half + 1) * n], 0, cp, s2d_cblk=cp)
        for i, blk in enumerate(blocks):
            if i == 0 and split:
                h = blk.forward_act(h, h1, s2d_out=True, bn_groups=2, split=split[:2])
            else:
                h = blk.forward_act(h, s2d_cp=cp, s2d_out=i < 4, bn_groups=2)
            cp = round_up(blk.conv.out_channels, 16)
            if i == 1 and blk.conv.weight.requires_grad:
                Fn.StageBoundary.mark(h)
        z, _ = self.final.forward_act(h)
        logits = Fn.UnpackFn.apply(z, 1)
        return logits if stacked else (logits[:n], logits[n:])


# ------------------------------------------------------------------------------------------
# monai.networks.nets.BasicUNet (module/parameter names follow MONAI's state_dict)
class _ADN(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.N = _NormParams(channels, batch=False)


class Convolution(_Mi355Module):
    """MONAI Convolution(k3, s1, p1, adn_ordering="NDA"): Conv3d -> InstanceNorm3d(affine) ->
    Dropout(p) -> LeakyReLU(0.1), fused as conv(+statistics) -> finalize -> one apply pass."""

    def __init__(self, cin, cout, dropout, slope=0.1, eps=1e-5):
        super().__init__()
        self.conv = Conv3d(cin, cout, 3, 1, 1, bias=True)
        self.adn = _ADN(cout)
        self.cfg = Fn.NormCfg("instance", cout, eps=eps, slope=slope, p=float(dropout or 0.0))

    def forward_act(self, x0, x1=None, feeds: Optional[Conv3d] = None, up_from=None, final: Optional[Conv3d] = None,
                    pool_after: bool = False):
        """feeds: the convolution that consumes the result (fp8 mode: the norm kernel writes its e4m3 operand as well);
        up_from = (x_low, deconv module, tables): the second source is ConvTranspose3d(x_low), which is NOT materialised
        (Fn.UpCatConvFn: the up-branch as one transposed 4x4x4 convolution of the low-resolution tensor);
        pool_after: the result's only consumer is Down.forward_skip (skip connection + MaxPool3d(2)): the norm + act launch writes
        the pooled tensor as well (Fn.PoolSide)"""
        n, d, h, w = x0.shape[:4]
        cp = round_up(self.conv.out_channels, 16)
        # small levels (16^3, 8^3): the norm kernel computes the instance statistics itself, in one launch
        small = ops.norm_is_small(n, d, h, w, cp)
        shift = self.conv.bias
        if up_from is not None:
            x_low, deconv, tables = up_from
            z, part = Fn.UpCatConvFn.apply(x0, x_low, deconv.weight, deconv.bias, self.conv.weight, self.conv.bias, self.conv.spec,
                                           tables, not small)
            shift = tables.bufs[2]         # the statistics are those of z - (b_c + the interior share of the deconv bias)
        else:
            z, part = self.conv.forward_act(x0, x1, want_stats=not small, zero_bias_grad=True)
        emit8 = emit8_bwd = None
        if cp == 32 and z.dtype == torch.bfloat16 and Fn.Fp8Scales.producer_side:
            # BASELINE.json configs[4]: e4m3 operands come from the kernel that produces the bf16 tensor (delayed scaling)
            if feeds is not None and feeds.fp8 and ops.conv_fp8_layer_ok(n, d, h, w, cp, feeds.out_channels):
                emit8 = feeds.spec.fp8_slot("x", z.device)
            cin = x0.shape[4] + (x1.shape[4] if x1 is not None else 0)
            if (self.conv.fp8 and up_from is None and torch.is_grad_enabled() and (x0.requires_grad or (x1 is not None and x1.requires_grad))
                    and ops.conv_fp8_layer_ok(n, d, h, w, cp, cin)):
                emit8_bwd = self.conv.spec.fp8_slot("g", z.device)
        fin = None
        if (final is not None and not small and z.dtype == torch.bfloat16 and cp == 32 and emit8 is None and final.out_channels <= 8
                and final.in_channels <= cp and Fn.LazyDx.enabled):
            # `final` is the 1x1x1 convolution that is this block's only consumer: evaluated by the norm + act launch itself
            fin = (final.weight, final.bias, not torch.is_grad_enabled())
        return Fn.NormActFn.apply(z, part if not small else None, self.adn.N.weight, self.adn.N.bias, shift, self.cfg,
                                  self.training, None, None, False, None, small, 1, emit8, emit8_bwd, fin, pool_after)


class TwoConv(_Mi355Module):
    def __init__(self, cin, cout, dropout):
        super().__init__()
        self.conv_0 = Convolution(cin, cout, dropout)
        self.conv_1 = Convolution(cout, cout, dropout)

    def forward_act(self, x0, x1=None, up_from=None, final=None, pool_after=False):
        return self.conv_1.forward_act(self.conv_0.forward_act(x0, x1, feeds=self.conv_1.conv, up_from=up_from), final=final,
                                       pool_after=pool_after)


class Down(_Mi355Module):
    def __init__(self, cin, cout, dropout):
        super().__init__()
        self.convs = TwoConv(cin, cout, dropout)

    def forward_act(self, x):
        return self.convs.forward_act(Fn.MaxPoolFn.apply(x))

    def forward_skip(self, x, pool_after=False):
        """-> (x for the skip connection, this level's output): both uses of x leave one autograd node.
        pool_after: this level's output goes to the next level's forward_skip and nowhere else"""
        skip, pooled = Fn.SkipPoolFn.apply_to(x)
        return skip, self.convs.forward_act(pooled, pool_after=pool_after)


class _UpSample(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.deconv = ConvTranspose3d(cin, cout, 2, 2, bias=True)


class UpCat(_Mi355Module):
    def __init__(self, cin, ccat, cout, dropout, halves=True):
        super().__init__()
        cup = cin // 2 if halves else cin
        self.upsample = _UpSample(cin, cup)
        self.convs = TwoConv(ccat + cup, cout, dropout)

    fuse_up_branch = True      # bf16, even extents: ConvTranspose3d + concat + Conv3d without the up-sampled tensor (Fn.UpCatConvFn)

    def _fused(self, x, x_e) -> bool:
        deconv, conv = self.upsample.deconv, self.convs.conv_0.conv
        co, cl = conv.out_channels, deconv.in_channels
        return (self.fuse_up_branch and self.compute_dtype == torch.bfloat16 and x.dtype == torch.bfloat16
                and tuple(x_e.shape[1:4]) == tuple(2 * e for e in x.shape[1:4])
                and co == 32 and cl % 32 == 0 and x.shape[4] == cl and x_e.shape[4] % 16 == 0    # (upcat_1; 64 output channels measured slower)
                and x_e.shape[4] + deconv.out_channels == conv.in_channels and deconv.bias is not None
                and x.shape[3] >= 32 and not ops.norm_is_small(x_e.shape[0], *x_e.shape[1:4], co))

    def forward_act(self, x, x_e, final=None):
        if self._fused(x, x_e):
            if not hasattr(self, "upcat_tables"):
                self.upcat_tables = Fn.UpCatTables()
            return self.convs.forward_act(x_e, None, up_from=(x, self.upsample.deconv, self.upcat_tables), final=final)
        x_0 = self.upsample.deconv.forward_act(x)
        if x_0.shape[1:4] != x_e.shape[1:4]:
            # MONAI UpCat (is_pad): a level whose extent is odd loses its last plane in MaxPool3d(2); the up-sampled map is
            # then one short of the skip and is replicate-padded by one at the END of every such axis.  Never taken by
            # the reference's own volumes (src/model.py:109-110 asserts divisibility by 16); plain tensor ops, autograd
            # handles the gradient (the padded plane's gradient is added to the last real one).
            for ax in (1, 2, 3):
                if x_0.shape[ax] + 1 == x_e.shape[ax]:
                    x_0 = torch.cat([x_0, x_0.narrow(ax, x_0.shape[ax] - 1, 1)], ax)
            if x_0.shape[1:4] != x_e.shape[1:4]:
                raise ValueError(f"skip {tuple(x_e.shape[1:4])} and up-sampled map {tuple(x_0.shape[1:4])} differ by more than one")
        return self.convs.forward_act(x_e, x_0, final=final)        # virtual cat([x_e, x_0], 1): skip first


class BasicUNet(_Mi355Module):
    """BasicUNet(spatial_dims=3, in_channels, out_channels, features, act, norm, bias, dropout, upsample)."""

    lazy_final_dx = True       # bf16: the final 1x1x1 convolution's data gradient is formed inside upcat_1's norm backward (Fn.LazyDx)

    def __init__(self, spatial_dims: int = 3, in_channels: int = 1, out_channels: int = 2,
                 features: Sequence[int] = (32, 32, 64, 128, 256, 32),
                 act=("LeakyReLU", {"negative_slope": 0.1, "inplace": True}),
                 norm=("instance", {"affine": True}), bias: bool = True, dropout=0.0, upsample: str = "deconv"):
        super().__init__()
        if spatial_dims not in (2, 3):
            raise NotImplementedError("spatial_dims must be 2 or 3")
        self.spatial_dims = spatial_dims
        rng_state = torch.get_rng_state() if spatial_dims == 2 else None
        if upsample != "deconv" or not bias:
            raise NotImplementedError("only upsample='deconv', bias=True (the reference's configuration)")
        act_name = act[0] if isinstance(act, (tuple, list)) else act
        norm_name = norm[0] if isinstance(norm, (tuple, list)) else norm
        if str(act_name).lower() != "leakyrelu" or str(norm_name).lower() != "instance":
            raise NotImplementedError("only act=LeakyReLU / norm=instance (MONAI defaults used by the reference)")
        f = tuple(features)
        if len(f) != 6:
            raise ValueError("features must have 6 entries")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.conv_0 = TwoConv(in_channels, f[0], dropout)
        self.down_1 = Down(f[0], f[1], dropout)
        self.down_2 = Down(f[1], f[2], dropout)
        self.down_3 = Down(f[2], f[3], dropout)
        self.down_4 = Down(f[3], f[4], dropout)
        self.upcat_4 = UpCat(f[4], f[3], f[3], dropout)
        self.upcat_3 = UpCat(f[3], f[2], f[2], dropout)
        self.upcat_2 = UpCat(f[2], f[1], f[1], dropout)
        self.upcat_1 = UpCat(f[1], f[0], f[5], dropout, halves=False)
        self.final_conv = Conv3d(f[5], out_channels, kernel_size=1)
        if spatial_dims == 2:
            self._make_2d(rng_state)

    # ---- spatial_dims = 2 (BASELINE.json configs[0], the reference's CPU-runnable plumbing case) ---------------------------
    # The 2-D network runs on the 3-D kernels: parameters are registered with MONAI's 2-D names and shapes ((co, ci, 3, 3),
    # (ci, co, 2, 2)) and embedded per call -- a 3x3 kernel becomes the kd = 1 plane of a 3x3x3 kernel (the other planes
    # zero), a 2x2 transposed-conv kernel is repeated along kd -- and the (N, C, H, W) input is repeated over 16 d-slices,
    # which four 2x poolings reduce to one.  Every slice then holds the 2-D network's values (a convolution never mixes
    # slices, pooling and statistics of identical slices are the 2-D ones) and the mean over the slices is returned, so that
    # the incoming gradient is spread over them and every parameter gradient is the 2-D one (backward is linear and the
    # activations are identical across slices).  16x the 2-D FLOPs: a plumbing configuration, not a performance path.
    # Element-wise dropout draws an independent mask per slice (same expectation, 1/16 of the variance).
    _REPL = 16

    def _make_2d(self, rng_state):
        torch.set_rng_state(rng_state)                 # consume the RNG exactly like the 2-D torch modules would
        for m in self.modules():
            if isinstance(m, (Conv3d, ConvTranspose3d)):
                k = m.weight.shape[-1]
                m.weight = nn.Parameter(torch.empty(*m.weight.shape[:2], k, k))
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                if m.bias is not None:
                    fan_in, _ = nn.init._calculate_fan_in_and_fan_out(m.weight)
                    bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                    nn.init.uniform_(m.bias, -bound, bound)

    def _embedded_parameters(self):
        deconv_w = {id(m.weight) for m in self.modules() if isinstance(m, ConvTranspose3d)}
        out = {}
        for name, p in self.named_parameters():
            if p.dim() == 4 and id(p) in deconv_w:
                out[name] = p.unsqueeze(2).expand(-1, -1, 2, -1, -1).contiguous()
            elif p.dim() == 4:
                w = p.unsqueeze(2)
                out[name] = torch.nn.functional.pad(w, (0, 0, 0, 0, 1, 1)) if p.shape[-1] == 3 else w
            else:
                out[name] = p
        return out

    def _forward_2d(self, x):
        if x.dim() != 4:
            raise ValueError(f"spatial_dims=2 expects (N, C, H, W), got {tuple(x.shape)}")
        x3 = x.unsqueeze(2).expand(-1, -1, self._REPL, -1, -1).contiguous()
        y3 = torch.func.functional_call(self, self._embedded_parameters(), (x3,), {"_embedded": True})
        return y3.mean(2)

    def forward_act(self, x):
        d, h, w = x.shape[1:4]
        if min(d, h, w) < 16 or (d >> 4) * (h >> 4) * (w >> 4) < 2:
            raise ValueError(f"four 2x poolings need extents >= 16, and InstanceNorm more than one element at the bottom "
                             f"level; got {tuple(x.shape[1:4])}")
        # (pool_after: a level's output goes to the next level's forward_skip and nowhere else -- its norm + act launch writes the
        #  pooled tensor too, Fn.PoolSide, and its backward kernels form the pool's gradient themselves, Fn.LazyPool)
        x0 = self.conv_0.forward_act(x, pool_after=True)
        s0, x1 = self.down_1.forward_skip(x0, pool_after=True)
        s1, x2 = self.down_2.forward_skip(x1, pool_after=True)
        if Fn.StageBoundary.active():
            # backward stage cut: everything from down_3 on | {down_2, down_1, conv_0, head}.  The skip tensors get a node of
            # their own: a gradient captured AT the two-output SkipPoolFn node would make autograd execute every path
            # into that node -- the pooled branch through down_1 / down_2 included -- already in the first stage.
            s0, s1 = s0.view_as(s0), s1.view_as(s1)
            Fn.StageBoundary.mark(x2, s1, s0)
        s2, x3 = self.down_3.forward_skip(x2, pool_after=True)
        s3, x4 = self.down_4.forward_skip(x3)
        u4 = self.upcat_4.forward_act(x4, s3)
        u3 = self.upcat_3.forward_act(u4, s2)
        u2 = self.upcat_2.forward_act(u3, s1)
        # u1 is the output of upcat_1's second norm + act node and feeds nothing but the final 1x1x1 convolution: that node's
        # launches evaluate the convolution (forward: Fn.FusedFinal) and form its data gradient (backward: Fn.LazyDx) themselves
        # (bf16, <= 8 output channels)
        lazy = self.lazy_final_dx and u2.dtype == torch.bfloat16
        u1 = self.upcat_1.forward_act(u2, s0, final=self.final_conv if lazy else None)
        z, _ = self.final_conv.forward_act(u1, lazy_dx=lazy and not ops.norm_is_small(*u1.shape))
        return z

    def forward(self, x, _embedded: bool = False):
        if self.spatial_dims == 2 and not _embedded:
            return self._forward_2d(x)
        z = self.forward_act(self._to_act(x, round_up(self.in_channels, 16)))
        return Fn.UnpackFn.apply(z, self.out_channels)


def backward_stages(net: nn.Module, modality):
    """(late, early) parameter lists of a Generator / Discriminator for the two backward stages cut at
    ``Fn.StageBoundary`` (the unused modality heads are in neither list: src/model.py:29-34, 74-78)."""
    if isinstance(net, Generator):
        unet = net.blocks["unet"]
        early_mods = [net.blocks[modality], unet.conv_0, unet.down_1, unet.down_2]
        late_mods = [unet.down_3, unet.down_4, unet.upcat_4, unet.upcat_3, unet.upcat_2, unet.upcat_1, unet.final_conv]
    elif isinstance(net, Discriminator):
        early_mods = [net.d1[modality], net.d2]
        late_mods = [net.d3, net.d4, net.d5, net.final]
    else:
        raise TypeError("backward_stages: Generator or Discriminator expected")
    late = [p for m in reversed(late_mods) for p in reversed(list(m.parameters()))]
    early = [p for m in reversed(early_mods) for p in reversed(list(m.parameters()))]
    return late, early


class Generator(_Mi355Module):
    """src/model.py:15-39: modality head (1x1x1 DownSampleConv) -> BasicUNet(24 -> 6)."""

    def __init__(self, input_modality, dropout=0.05, features=(32, 64, 128, 256, 512, 32)):
        super().__init__()
        self.input_modality = input_modality
        dwi_tensor_input = DownSampleConv(6, 24, kernel=1, strides=1, padding=0)
        bssfp_input = DownSampleConv(24, 24, kernel=1, strides=1, padding=0)
        unet = BasicUNet(spatial_dims=3, in_channels=24, out_channels=6, features=features, dropout=dropout)
        self.blocks = nn.ModuleDict({"dwi-tensor": dwi_tensor_input, "pc-bssfp": bssfp_input,
                                     "bssfp": bssfp_input, "t1w": dwi_tensor_input, "unet": unet})

    emit_s2d = False      # set by the training harness: also hand S(output) to the PatchGAN (Fn.GenOutFn), bf16 mode, even extents

    def forward(self, x):
        head = self.blocks[self.input_modality]
        a = head.forward_act(head._to_act(x))
        z = self.blocks["unet"].forward_act(a)
        if self.emit_s2d and z.dtype == torch.bfloat16 and all(e % 2 == 0 for e in z.shape[1:4]):
            y, s = Fn.GenOutFn.apply(z, 6, 8)
            y._mi355_s2d = s
            return y
        return Fn.UnpackFn.apply(z, 6)
