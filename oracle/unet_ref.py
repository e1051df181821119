"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

Plain-PyTorch (CPU, fp32) restatement of the reference hot path, used only by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as
the checker / reported baseline.  The shipped path (``unet_bssfp_amd``) is
hand-written HIP behind a C-ABI and raises if that library is missing.

What is restated, and from where (citations into /root/reference):

* ``RefDownSampleConv``   -- src/model.py:42-65   (Conv3d -> BatchNorm3d? -> LeakyReLU(0.2)?)
* ``RefDiscriminator``    -- src/model.py:68-92   (cat -> d1..d5 -> 1x1x1 conv, shared heads)
* ``RefGenerator``        -- src/model.py:15-39   (modality head -> BasicUNet)
* ``RefBasicUNet``        -- call site src/model.py:22-28.  The arithmetic lives in the
  un-vendored third-party package ``monai==1.3.0`` (requirements.txt:3;
  monai/networks/nets/basic_unet.py + blocks/convolutions.py + blocks/acti_norm.py +
  blocks/upsample.py).  MONAI is absent from this image and the reference holds no
  test or golden vector for it, so this class follows MONAI's published algorithm
   and its parity with real MONAI is **UNPINNED**; the only
  corroboration is the op order in doc/thesis/img/model.onnx.png.
* ``gan_training_step``   -- src/model.py:170-193, 201-213, 259-281, 359-361 (manual
  optimisation: generator phase then discriminator phase, two AdamW(lr=1e-3)).
  The Perceptual term (src/model.py:127-129) needs remotely fetched MedicalNet
  weights and is excluded everywhere (slot kept: ``extra_recon_terms``).

Pinned parts: DownSampleConv / Discriminator / Generator wiring are checked against
the reference's own classes (AST-extracted from src/model.py at fixture-generation
time by ``oracle/gen_golden.py``; outputs committed under tests/golden/).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------------
# Constants of the path, in one table so a MONAI-derived value can be fixed in one place
# ----------------------------------------------------------------------------------
UNET_FEATURES = (32, 64, 128, 256, 512, 32)     # src/model.py:26
UNET_DROPOUT = 0.05                             # src/model.py:27
UNET_LRELU_SLOPE = 0.1                          # MONAI BasicUNet default act  [3P, unpinned]
NORM_EPS = 1e-5                                 # torch InstanceNorm3d / BatchNorm3d default
BN_MOMENTUM = 0.1                               # torch BatchNorm3d default
PATCHGAN_LRELU_SLOPE = 0.2                      # src/model.py:57
MODALITY_IN_CHANNELS = {"dwi-tensor": 6, "t1w": 6, "pc-bssfp": 24, "bssfp": 24}  # src/model.py:19-21
HEAD_SHARING = {"dwi-tensor": "dwi-tensor", "t1w": "dwi-tensor",
                "pc-bssfp": "pc-bssfp", "bssfp": "pc-bssfp"}                      # src/model.py:29-33


def _conv_nd(dims: int):
    return {2: nn.Conv2d, 3: nn.Conv3d}[dims]


# ----------------------------------------------------------------------------------
# Storage emulation of the HIP path's throughput mode (NOT part of the reference): with
# ``storage_emulation(torch.bfloat16)`` active, every tensor the HIP bf16 mode keeps in bf16 is rounded
# through bf16 at the same point -- network inputs (pack), packed weights (f32 master weights, straight-through
# gradient), conv / transposed-conv outputs, norm+activation outputs, and the gradients flowing back through
# the same points (the HIP data-gradient and norm-backward kernels store bf16) -- while all arithmetic,
# statistics and weight gradients stay f32, as in the kernels.  The f32 oracle tells how far bf16 storage moves
# a result; the emulated oracle tells whether the KERNELS do anything beyond that (tests/test_gpu_bf16.py).
# ----------------------------------------------------------------------------------
_STORAGE = [None]


class storage_emulation:
    def __init__(self, dtype):
        assert dtype in (None, torch.bfloat16)
        self.dtype = dtype

    def __enter__(self):
        self.prev, _STORAGE[0] = _STORAGE[0], self.dtype
        return self

    def __exit__(self, *exc):
        _STORAGE[0] = self.prev


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def _qa(t):
    """activation / gradient storage point"""
    return t if _STORAGE[0] is None else _RoundBoth.apply(t, _STORAGE[0])


def _qw(w):
    """packed weights: rounded value, gradient passed straight to the f32 master weight"""
    return w if _STORAGE[0] is None else w + (w.detach().to(_STORAGE[0]).to(w.dtype) - w.detach())


# ----------------------------------------------------------------------------------
# src/model.py:42-65
# ----------------------------------------------------------------------------------
class RefDownSampleConv(nn.Module):
    """Conv3d(k,s,p, bias) -> [BatchNorm3d] -> [LeakyReLU(0.2)]; defaults k4 s2 p1."""

    def __init__(self, in_channels, out_channels, kernel=4, strides=2, padding=1,
                 activation=True, batchnorm=True):
        super().__init__()
        self.has_act = bool(activation)
        self.has_bn = bool(batchnorm)
        self.conv = nn.Conv3d(in_channels, out_channels, kernel, strides, padding)
        if self.has_bn:
            self.bn = nn.BatchNorm3d(out_channels)

    def forward(self, x):
        z = _qa(F.conv3d(x, _qw(self.conv.weight), self.conv.bias,
                         self.conv.stride, self.conv.padding))
        if self.has_bn:
            z = F.batch_norm(z, self.bn.running_mean, self.bn.running_var,
                             self.bn.weight, self.bn.bias, self.training,
                             BN_MOMENTUM, NORM_EPS)
            if self.training:
                self.bn.num_batches_tracked += 1
        if self.has_act:
            z = F.leaky_relu(z, PATCHGAN_LRELU_SLOPE)
        return _qa(z) if (self.has_bn or self.has_act) else z


# ----------------------------------------------------------------------------------
# src/model.py:68-92
# ----------------------------------------------------------------------------------
class RefDiscriminator(nn.Module):
    def __init__(self, modality):
        super().__init__()
        self.modality = modality
        head24 = RefDownSampleConv(30, 32, batchnorm=False)      # :72
        head6 = RefDownSampleConv(12, 32, batchnorm=False)       # :73
        table = nn.ModuleDict({"dwi-tensor": head6, "pc-bssfp": head24,
                               "bssfp": head24, "t1w": head6})
        self.d1 = table                                          # :74 (registered twice)
        self.blocks = table
        self.d2 = RefDownSampleConv(32, 64)
        self.d3 = RefDownSampleConv(64, 128)
        self.d4 = RefDownSampleConv(128, 256)
        self.d5 = RefDownSampleConv(256, 512)
        self.final = nn.Conv3d(512, 1, kernel_size=1)            # :83

    def forward(self, x, y):
        h = _qa(torch.cat([x, y], dim=1))                        # :86
        h = self.d1[self.modality](h)
        for blk in (self.d2, self.d3, self.d4, self.d5):
            h = blk(h)
        return _qa(F.conv3d(h, _qw(self.final.weight), self.final.bias))   # raw logits


# ----------------------------------------------------------------------------------
# monai==1.3.0 BasicUNet, restated (UNPINNED -- see module docstring)
# ----------------------------------------------------------------------------------
class _RefADN(nn.Module):
    """MONAI ADN with ordering "NDA": InstanceNorm(affine) -> Dropout -> LeakyReLU."""

    def __init__(self, channels, dims, dropout):
        super().__init__()
        norm = {2: nn.InstanceNorm2d, 3: nn.InstanceNorm3d}[dims]
        self.N = norm(channels, eps=NORM_EPS, affine=True)
        self.p = float(dropout)

    def forward(self, z, drop_mask=None):
        z = F.instance_norm(z, None, None, self.N.weight, self.N.bias,
                            True, 0.0, NORM_EPS)
        if drop_mask is not None:                 # explicit mask (tests feed identical masks)
            z = z * drop_mask
        elif self.training and self.p > 0.0:
            z = F.dropout(z, self.p, True)
        return F.leaky_relu(z, UNET_LRELU_SLOPE)


class _RefConvolution(nn.Module):
    """MONAI ``Convolution(k=3, s=1, padding=1, adn_ordering="NDA")``."""

    def __init__(self, dims, cin, cout, dropout):
        super().__init__()
        self.conv = _conv_nd(dims)(cin, cout, kernel_size=3, stride=1, padding=1, bias=True)
        self.adn = _RefADN(cout, dims, dropout)

    def forward(self, x):
        c = self.conv
        return _qa(self.adn(_qa(c._conv_forward(x, _qw(c.weight), c.bias))))


class _RefTwoConv(nn.Module):
    def __init__(self, dims, cin, cout, dropout):
        super().__init__()
        self.conv_0 = _RefConvolution(dims, cin, cout, dropout)
        self.conv_1 = _RefConvolution(dims, cout, cout, dropout)

    def forward(self, x):
        return self.conv_1(self.conv_0(x))


class _RefDown(nn.Module):
    def __init__(self, dims, cin, cout, dropout):
        super().__init__()
        self.dims = dims
        self.convs = _RefTwoConv(dims, cin, cout, dropout)

    def forward(self, x):
        pool = F.max_pool3d if self.dims == 3 else F.max_pool2d
        return self.convs(pool(x, 2))


class _RefUpsample(nn.Module):
    def __init__(self, dims, cin, cout):
        super().__init__()
        ct = {2: nn.ConvTranspose2d, 3: nn.ConvTranspose3d}[dims]
        self.deconv = ct(cin, cout, kernel_size=2, stride=2, bias=True)

    def forward(self, x):
        d = self.deconv
        fn = F.conv_transpose3d if isinstance(d, nn.ConvTranspose3d) else F.conv_transpose2d
        return _qa(fn(x, _qw(d.weight), d.bias, stride=2))


class _RefUpCat(nn.Module):
    def __init__(self, dims, cin, ccat, cout, dropout, halves=True):
        super().__init__()
        cup = cin // 2 if halves else cin
        self.dims = dims
        self.upsample = _RefUpsample(dims, cin, cup)
        self.convs = _RefTwoConv(dims, ccat + cup, cout, dropout)

    def forward(self, x, x_e):
        x_0 = self.upsample(x)
        # MONAI: replicate-pad the upsampled map by one at the end of every spatial
        # dim where the skip is larger (odd skip sizes)
        pad = [0] * (2 * self.dims)
        for i in range(self.dims):
            if x_e.shape[-i - 1] != x_0.shape[-i - 1]:
                pad[i * 2 + 1] = 1
        if any(pad):
            x_0 = F.pad(x_0, pad, "replicate")
        return self.convs(torch.cat([x_e, x_0], dim=1))          # skip first


class RefBasicUNet(nn.Module):
    """``BasicUNet(spatial_dims, in_channels, out_channels, features, dropout)``."""

    def __init__(self, spatial_dims=3, in_channels=1, out_channels=2,
                 features: Sequence[int] = (32, 32, 64, 128, 256, 32),
                 act=None, norm=None, bias=True, dropout=0.0, upsample="deconv"):
        super().__init__()
        if upsample != "deconv" or not bias:
            raise NotImplementedError("oracle covers the reference's configuration only")
        f = tuple(features)
        if len(f) != 6:
            raise ValueError("features must have 6 entries")
        d = spatial_dims
        self.conv_0 = _RefTwoConv(d, in_channels, f[0], dropout)
        self.down_1 = _RefDown(d, f[0], f[1], dropout)
        self.down_2 = _RefDown(d, f[1], f[2], dropout)
        self.down_3 = _RefDown(d, f[2], f[3], dropout)
        self.down_4 = _RefDown(d, f[3], f[4], dropout)
        self.upcat_4 = _RefUpCat(d, f[4], f[3], f[3], dropout)
        self.upcat_3 = _RefUpCat(d, f[3], f[2], f[2], dropout)
        self.upcat_2 = _RefUpCat(d, f[2], f[1], f[1], dropout)
        self.upcat_1 = _RefUpCat(d, f[1], f[0], f[5], dropout, halves=False)
        self.final_conv = _conv_nd(d)(f[5], out_channels, kernel_size=1)

    def forward(self, x):
        x0 = self.conv_0(x)
        x1 = self.down_1(x0)
        x2 = self.down_2(x1)
        x3 = self.down_3(x2)
        x4 = self.down_4(x3)
        u4 = self.upcat_4(x4, x3)
        u3 = self.upcat_3(u4, x2)
        u2 = self.upcat_2(u3, x1)
        u1 = self.upcat_1(u2, x0)
        f = self.final_conv
        return _qa(f._conv_forward(u1, _qw(f.weight), f.bias))


# ----------------------------------------------------------------------------------
# src/model.py:15-39
# ----------------------------------------------------------------------------------
class RefGenerator(nn.Module):
    def __init__(self, input_modality, dropout=UNET_DROPOUT, unet_cls=RefBasicUNet):
        super().__init__()
        self.input_modality = input_modality
        head6 = RefDownSampleConv(6, 24, kernel=1, strides=1, padding=0)     # :19
        head24 = RefDownSampleConv(24, 24, kernel=1, strides=1, padding=0)   # :21
        unet = unet_cls(spatial_dims=3, in_channels=24, out_channels=6,
                        features=UNET_FEATURES, dropout=dropout)
        self.blocks = nn.ModuleDict({"dwi-tensor": head6, "pc-bssfp": head24,
                                     "bssfp": head24, "t1w": head6, "unet": unet})

    def forward(self, x):
        return self.blocks["unet"](self.blocks[self.input_modality](_qa(x)))


# ----------------------------------------------------------------------------------
# src/model.py:170-193, 201-213, 259-281  (training-step semantics, module-agnostic)
# ----------------------------------------------------------------------------------
def _set_requires_grad(module: nn.Module, flag: bool):
    for p in module.parameters():
        p.requires_grad_(flag)


def gan_training_step(gen: nn.Module, discr: nn.Module, gen_opt, discr_opt,
                      x: torch.Tensor, y: torch.Tensor, recon_factor: float = 1e2,
                      extra_recon_terms: Optional[Dict[str, Callable]] = None
                      ) -> Dict[str, torch.Tensor]:
    """One ``training_step`` (src/model.py:259-281) on a batch (x, y).

    ``toggle_optimizer`` (Lightning) == requires_grad False on the other network's
    parameters for the duration of the phase.  Returns the six logged scalars
    (detached tensors, no host sync).
    """
    logs: Dict[str, torch.Tensor] = {}
    bce = F.binary_cross_entropy_with_logits

    # ---- generator phase (:264-271, _gen_step :170-181)
    _set_requires_grad(discr, False)
    y_hat = gen(x)
    logits = discr(x, y_hat)
    adv = bce(logits, torch.ones_like(logits))
    terms = OrderedDict(L1=F.l1_loss(y_hat, y))                      # :136
    for name, fn in (extra_recon_terms or {}).items():               # Perceptual slot (:137)
        terms[name] = fn(y_hat, y)
    recon = sum(terms.values()) / len(terms) * recon_factor          # :209
    gen_loss = adv + recon                                           # :181
    gen_loss.backward()                                              # :268
    gen_opt.step()
    gen_opt.zero_grad()
    _set_requires_grad(discr, True)
    logs.update(gen_loss_adversarial=adv.detach(), gen_loss_recon=recon.detach(),
                gen_loss=gen_loss.detach())
    for k, v in terms.items():
        logs[f"gen_loss_recon_{k}"] = v.detach()

    # ---- discriminator phase (:274-281, _discr_step :183-193)
    _set_requires_grad(gen, False)
    y_fake = gen(x).detach()
    logits_fake = discr(x, y_fake)
    logits_real = discr(x, y)
    d_loss = (bce(logits_real, torch.ones_like(logits_real))
              + bce(logits_fake, torch.zeros_like(logits_fake))) / 2
    d_loss.backward()                                                # :278
    discr_opt.step()
    discr_opt.zero_grad()
    _set_requires_grad(gen, True)
    logs["discr_loss"] = d_loss.detach()
    return logs


def make_optimizers(gen: nn.Module, discr: nn.Module, lr: float = 1e-3):
    """src/model.py:359-361 -- two torch.optim.AdamW with torch defaults."""
    return (torch.optim.AdamW(gen.parameters(), lr=lr),
            torch.optim.AdamW(discr.parameters(), lr=lr))


def synthetic_batch(n: int, s, seed: int, cin: int = 24, cout: int = 6
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """U[0,1) volumes (every modality is min-max normalised: doc/thesis/03-methods.tex:670)."""
    if isinstance(s, int):
        s = (s, s, s)
    g = torch.Generator("cpu").manual_seed(seed)
    x = torch.rand(n, cin, *s, generator=g)
    y = torch.rand(n, cout, *s, generator=g)
    return x, y
