"""GPU-side intensity augmentations -- SURVEY.md 8(f) rank 4 (augmentation part).

``RandomBiasField``, ``RandomNoise`` and ``RandomGamma`` with TorchIO's constructor arguments and sampling rules
(the three cheap, image-space members of the reference's training transform, src/data_module.py:130-139), applied
to device tensors of shape (C, D, H, W) so that the input pipeline can keep up with a GPU that trains > 60 volumes
per second.  Random parameters are drawn with torch's CPU generator like TorchIO does; the voxel noise comes from a
counter-based hash on the device.  RandomMotion / RandomGhosting / RandomSpike (k-space) and RandomBlur are not built.
TorchIO is absent: behaviour restated from its published algorithm (oracle/augment_ref.py, parity unpinned).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib

Range = Union[float, Tuple[float, float]]


def _range(v: Range, symmetric: bool) -> Tuple[float, float]:
    if isinstance(v, (int, float)):
        return (-float(v), float(v)) if symmetric else (0.0, float(v))
    lo, hi = v
    return float(lo), float(hi)


def _check(x: torch.Tensor):
    if not x.is_cuda:
        raise _lib.Mi355Error("augmentations run on the GPU only (no CPU fallback)")
    if x.dim() != 4:
        raise ValueError(f"expected (C, D, H, W), got {tuple(x.shape)}")
    return x.float().contiguous()


class _Random:
    def __init__(self, p: float = 1.0):
        self.p = float(p)

    def __call__(self, subject):
        """subject: a (C, D, H, W) tensor or ``{name: {'data': tensor}}`` -- like TorchIO, ONE set of random
        parameters per call, applied to every image of the subject."""
        if torch.rand(1).item() >= self.p:
            return subject
        params = self.sample()
        if isinstance(subject, torch.Tensor):
            return self.apply(subject, params)
        return {k: ({**v, "data": self.apply(v["data"], params)} if isinstance(v, dict) and "data" in v else v)
                for k, v in subject.items()}


class RandomBiasField(_Random):
    def __init__(self, coefficients: Range = 0.5, order: int = 3, p: float = 1.0):
        super().__init__(p)
        if not 0 <= order <= 4:
            raise ValueError("order must be in 0..4")
        self.coefficients_range, self.order = _range(coefficients, True), int(order)

    def sample(self):
        n = (self.order + 1) * (self.order + 2) * (self.order + 3) // 6
        lo, hi = self.coefficients_range
        return (torch.rand(n) * (hi - lo) + lo).numpy().astype(np.float32)

    def apply(self, x, coefficients):
        x = _check(x)
        out = torch.empty_like(x)
        coefficients = np.ascontiguousarray(coefficients, dtype=np.float32)
        _lib.check(_lib.load().mi355_aug_bias_field(x.data_ptr(), out.data_ptr(), *x.shape, coefficients.ctypes.data, self.order,
                                                    torch.cuda.current_stream().cuda_stream), "aug_bias_field")
        return out


class RandomGamma(_Random):
    def __init__(self, log_gamma: Range = (-0.3, 0.3), p: float = 1.0):
        super().__init__(p)
        self.log_gamma_range = _range(log_gamma, True)

    def sample(self):
        lo, hi = self.log_gamma_range
        return math.exp(torch.rand(1).item() * (hi - lo) + lo)

    def apply(self, x, gamma):
        x = _check(x)
        out = torch.empty_like(x)
        _lib.check(_lib.load().mi355_aug_gamma(x.data_ptr(), out.data_ptr(), x.numel(), float(gamma),
                                               torch.cuda.current_stream().cuda_stream), "aug_gamma")
        return out


class RandomNoise(_Random):
    def __init__(self, mean: Range = 0.0, std: Range = (0, 0.25), p: float = 1.0):
        super().__init__(p)
        self.mean_range, self.std_range = _range(mean, True), _range(std, False)

    def sample(self):
        (ml, mh), (sl, sh) = self.mean_range, self.std_range
        return (torch.rand(1).item() * (mh - ml) + ml, torch.rand(1).item() * (sh - sl) + sl,
                int(torch.randint(0, 2 ** 62, (1,)).item()))

    def apply(self, x, params):
        mean, std, seed = params
        x = _check(x)
        out = torch.empty_like(x)
        _lib.check(_lib.load().mi355_aug_noise(x.data_ptr(), out.data_ptr(), x.numel(), float(mean), float(std), int(seed),
                                               torch.cuda.current_stream().cuda_stream), "aug_noise")
        return out


def crop_or_pad(x: torch.Tensor, target: Sequence[int], padding_value: float = 0.0) -> torch.Tensor:
    """``tio.CropOrPad(target, 0)`` (src/data_module.py:125-128): centred crop / constant pad of (C, D, H, W)."""
    out = x
    for ax, t in enumerate(target, start=1):
        n = out.shape[ax]
        if n > t:
            lo = (n - t) // 2
            out = out.narrow(ax, lo, t)
        elif n < t:
            lo = (t - n) // 2
            pad = [0, 0] * (out.dim() - 1 - ax) + [lo, t - n - lo]
            out = torch.nn.functional.pad(out, pad, value=padding_value)
    return out.contiguous()


def reference_augmentation() -> list:
    """the image-space members of src/data_module.py:131-139 that are built, with the reference's arguments"""
    return [RandomBiasField(p=0.1), RandomNoise(p=0.1, std=(0.01, 0.1)), RandomGamma(p=0.1)]
