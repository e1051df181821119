"""Validation metrics on the GPU -- SURVEY.md 8(f) rank 3.

Callables with the constructor signatures and (B, 1) results of the MONAI metrics the reference
instantiates at src/model.py:158-160 -- ``PSNRMetric(max_val)``, ``SSIMMetric(spatial_dims, data_range)``,
``MAEMetric()`` -- so that ``compute_metrics`` (:215-220) stays as it is: ``metric_fn(y_hat, y).mean()``.
MONAI is absent from this image; formulas restated in oracle/metrics_ref.py (parity unpinned).
The FID entry of the reference's list needs the remotely fetched MedicalNet weights and is not provided.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib


def _prep(y_pred: torch.Tensor, y: torch.Tensor):
    if not (y_pred.is_cuda and y.is_cuda):
        raise _lib.Mi355Error("metrics run on the GPU only (no CPU fallback)")
    if y_pred.shape != y.shape:
        raise ValueError(f"y_pred and y should have same shapes, got {tuple(y_pred.shape)} and {tuple(y.shape)}.")
    if y_pred.dim() < 2:
        raise ValueError("either channel or spatial dimensions required.")
    return y_pred.detach().float().contiguous(), y.detach().float().contiguous()


def error_sums(y_pred: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """(B, 2) f64: per item sum |y - y_pred| and sum (y - y_pred)^2."""
    a, b = _prep(y_pred, y)
    items, per = a.shape[0], a[0].numel()
    lib = _lib.load()
    part = torch.empty(items * lib.mi355_err_blocks(per) * 2, dtype=torch.float64, device=a.device)
    out = torch.empty((items, 2), dtype=torch.float64, device=a.device)
    _lib.check(lib.mi355_err_sums(a.data_ptr(), b.data_ptr(), per, items, part.data_ptr(), out.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "err_sums")
    return out


class MAEMetric:
    def __call__(self, y_pred, y):
        return (error_sums(y_pred, y)[:, :1] / y[0].numel()).float()


class MSEMetric:
    def __call__(self, y_pred, y):
        return (error_sums(y_pred, y)[:, 1:] / y[0].numel()).float()


class PSNRMetric:
    def __init__(self, max_val: float):
        self.max_val = float(max_val)

    def __call__(self, y_pred, y):
        mse = error_sums(y_pred, y)[:, 1:] / y[0].numel()
        return (20 * math.log10(self.max_val) - 10 * torch.log10(mse)).float()


class SSIMMetric:
    def __init__(self, spatial_dims: int, data_range: float = 1.0, kernel_type: str = "gaussian", win_size: int = 11,
                 kernel_sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03):
        if spatial_dims != 3:
            raise NotImplementedError("only the reference's 3-D configuration is built")
        if kernel_type != "gaussian":
            raise NotImplementedError("only the Gaussian window (MONAI's default) is built")
        if not 1 <= win_size <= 15:
            raise ValueError("win_size must be in 1..15")
        self.data_range, self.win_size, self.kernel_sigma, self.k1, self.k2 = data_range, win_size, kernel_sigma, k1, k2
        dist = torch.arange(start=(1 - win_size) / 2, end=(1 + win_size) / 2, step=1)
        g = torch.exp(-torch.pow(dist / kernel_sigma, 2) / 2)
        self._window = (g / g.sum()).float().contiguous()             # host side: travels by value

    def __call__(self, y_pred, y):
        a, b = _prep(y_pred, y)
        if a.dim() != 5:
            raise ValueError(f"y_pred should have 5 dimensions (batch, channel, D, H, W), got {a.dim()}.")
        n, c, d, h, w = a.shape
        lib = _lib.load()
        need = lib.mi355_ssim3d_workspace_bytes(n, c, d, h, w, self.win_size)
        if need < 0:
            raise ValueError(f"spatial size {(d, h, w)} is smaller than the {self.win_size}-wide window")
        work = torch.empty(need, dtype=torch.uint8, device=a.device)
        out = torch.empty((n, 1), dtype=torch.float64, device=a.device)
        c1, c2 = (self.k1 * self.data_range) ** 2, (self.k2 * self.data_range) ** 2
        _lib.check(lib.mi355_ssim3d(a.data_ptr(), b.data_ptr(), n, c, d, h, w, self.win_size,
                                    self._window.numpy().ctypes.data_as(C.c_void_p), c1, c2, work.data_ptr(), need,
                                    out.data_ptr(), torch.cuda.current_stream().cuda_stream), "ssim3d")
        return out.float()


def reference_metric_fns():
    """the list of src/model.py:158-160 without the FID entry"""
    return [[PSNRMetric(1), "PSNR"], [SSIMMetric(3, data_range=1), "SSIM"], [MAEMetric(), "L1"]]
