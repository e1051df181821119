"""Host-side cost of the per-step input feed (bench.py --fresh-batch): where does the step time go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import unet_bssfp_amd as M
from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch
import bench

dev = "cuda:0"
torch.manual_seed(0)
model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp").to(dev), discr=M.Discriminator("bssfp").to(dev)).train()
M.set_compute_dtype(model, torch.bfloat16)
batch = synthetic_batch(1, 128, seed=1, device=dev)
g = GraphedTrainingStep(model, batch, warmup=2)
fresh = bench.FreshBatches(g)


def run(mode, n=60):
    torch.cuda.synchronize()
    ts = {"swap": 0.0, "launch": 0.0, "prefetch": 0.0}
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter()
        if mode != "resident":
            fresh.swap_in()
        b = time.perf_counter()
        g()
        c = time.perf_counter()
        if mode == "fresh":
            fresh.prefetch()
        elif mode == "d2d_only":
            fresh.ready.record(fresh.stream)
        d = time.perf_counter()
        ts["swap"] += b - a; ts["launch"] += c - b; ts["prefetch"] += d - c
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{mode:10s} {dt:7.3f} ms/step  host: " + " ".join(f"{k}={v / n * 1e3:.3f}ms" for k, v in ts.items()), flush=True)


for _ in range(2):
    run("resident"); run("fresh"); run("d2d_only")
