"""Metric kernels on one 1x6x128^3 pair: time and algorithmic GB/s."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_bssfp_amd import metrics as M
y = torch.rand(1, 6, 128, 128, 128, device="cuda"); p = torch.rand_like(y)
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
nb = 2 * y.numel() * 4
for name, fn in (("mae", M.MAEMetric()), ("psnr", M.PSNRMetric(1)), ("ssim", M.SSIMMetric(3))):
    ms = timed(lambda: fn(p, y))
    print(json.dumps(dict(metric=name, ms=round(ms, 4), input_gbs=round(nb / ms / 1e6, 1))))
