"""Microbenchmark of the DTI scalar-map kernel: GB/s of algorithmic traffic (6 loads + 9 stores per voxel)
against the HBM roofline, and the oracle (numpy eigh, one core) on a bounded sample beside it."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unet_bssfp_amd import eval as E

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--cpu-voxels", type=int, default=200000)
a = ap.parse_args()
for dt in (torch.float32, torch.float64):
    for cf in (True, False):
        shape = (6,) + (a.size,) * 3 if cf else (a.size,) * 3 + (6,)
        x = torch.rand(shape, device="cuda", dtype=dt) * 1e-3
        for _ in range(3):
            E.calc_scalar_maps(x, channels_first=cf)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(a.iters):
            E.calc_scalar_maps(x, channels_first=cf)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.iters
        nvox = a.size ** 3
        gb = nvox * 15 * x.element_size() / 1e9
        print(json.dumps(dict(kernel="dti_scalar_maps", dtype=str(dt), channels_first=cf, voxels=nvox, ms=round(ms, 4),
                              gvox_per_s=round(nvox / ms / 1e6, 3), achieved_gbs=round(gb / ms * 1e3, 1),
                              hbm_frac=round(gb / ms * 1e3 / 8000, 4))))
from oracle import dti_ref
d = dti_ref.synthetic_tensor_field((a.cpu_voxels,), seed=1)
t = time.perf_counter(); dti_ref.scalar_maps(d); t = time.perf_counter() - t
print(json.dumps(dict(cpu_oracle_voxels=a.cpu_voxels, s=round(t, 3), mvox_per_s=round(a.cpu_voxels / t / 1e6, 4), kind="port (vectorised numpy eigh)")))
