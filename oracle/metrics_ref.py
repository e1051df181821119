"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Validation metrics, SURVEY.md 8(f) rank 3.

The reference builds ``monai.metrics.PSNRMetric(1)``, ``SSIMMetric(3, data_range=1)`` and ``MAEMetric()``
(/root/reference/src/model.py:158-160) and logs ``metric_fn(y_hat, y).mean()`` (:215-220).  MONAI
(requirements.txt: monai==1.3.0) is NOT installed and not vendored: what follows restates its published
formulas with stock torch CPU ops -- PARITY UNPINNED; the tests add closed-form known answers.

* MAE / MSE: mean over everything but the batch axis, result (B, 1).
* PSNR = 20 log10(max_val) - 10 log10(MSE).
* SSIM: Gaussian window (win 11, sigma 1.5, normalised 1-D profile, outer product), "valid" grouped
  convolution of x, y, xx, yy, xy; c1 = (0.01 R)^2, c2 = (0.03 R)^2;
  ssim = (2 mx my + c1) / (mx^2 + my^2 + c1) * (2 sxy + c2) / (sx + sy + c2); mean per batch item, (B, 1).
"""
import math

import torch
import torch.nn.functional as F


def mae(y_pred, y):
    return (y - y_pred).abs().flatten(1).mean(-1, keepdim=True)


def mse(y_pred, y):
    return (y - y_pred).pow(2).flatten(1).mean(-1, keepdim=True)


def psnr(y_pred, y, max_val=1.0):
    return 20 * math.log10(max_val) - 10 * torch.log10(mse(y_pred, y))


def gaussian_1d(kernel_size=11, sigma=1.5):
    dist = torch.arange(start=(1 - kernel_size) / 2, end=(1 + kernel_size) / 2, step=1)
    g = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return g / g.sum()


def ssim3d(y_pred, y, data_range=1.0, win_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03):
    c = y_pred.shape[1]
    g = gaussian_1d(win_size, kernel_sigma).to(y_pred.dtype)
    k = (g[:, None, None] * g[None, :, None] * g[None, None, :]).expand(c, 1, win_size, win_size, win_size).contiguous()
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    conv = lambda t: F.conv3d(t, k, groups=c)
    mx, my = conv(y_pred), conv(y)
    sx = conv(y_pred * y_pred) - mx * mx
    sy = conv(y * y) - my * my
    sxy = conv(y_pred * y) - mx * my
    cs = (2 * sxy + c2) / (sx + sy + c2)
    full = ((2 * mx * my + c1) / (mx ** 2 + my ** 2 + c1)) * cs
    return full.reshape(full.shape[0], -1).mean(1, keepdim=True)
