"""Whole-volume grid inference: the reference's test volume (24, 96, 128, 128), 64^3 patches,
overlap 0 -> 8 patches.  Reports volumes/s and what share the patch gather / aggregation kernels take."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_bssfp_amd import inference as I, nn as N

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()


def timed(fn, iters):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


vol = torch.rand(24, 96, 128, 128, device="cuda")
for dt in ("bf16", "f32"):
    torch.manual_seed(0)
    gen = N.Generator("bssfp").cuda()
    N.set_compute_dtype(gen, torch.bfloat16 if dt == "bf16" else torch.float32)
    ms = timed(lambda: I.predict_volume(gen, vol, 64, 0, batch_size=a.batch), a.iters)
    sampler = I.GridSampler({"x": {"data": vol}}, 64)
    ms_g = timed(lambda: [b for b in sampler.batches(a.batch)], a.iters)
    pred = torch.rand(len(sampler), 6, 64, 64, 64, device="cuda")
    locs = sampler.locations

    def agg():
        g = I.GridAggregator(sampler)
        g.add_batch(pred, locs)
        return g.get_output_tensor()
    ms_a = timed(agg, a.iters)
    gather_bytes = 2 * len(sampler) * 24 * 64 ** 3 * 4
    agg_bytes = 2 * 6 * 96 * 128 * 128 * 4
    print(json.dumps(dict(workload="predict 24x96x128x128, 8 patches of 64^3", dtype=dt, ms_per_volume=round(ms, 3),
                          volumes_per_s=round(1e3 / ms, 2), gather_ms=round(ms_g, 4), gather_gbs=round(gather_bytes / ms_g / 1e6, 1),
                          aggregate_ms=round(ms_a, 4), aggregate_gbs=round(agg_bytes / ms_a / 1e6, 1))))
