#!/usr/bin/env python3
"""Benchmark of the hot path: GAN training-step volumes/s on synthetic 128^3 bSSFP -> 6-ch DTI.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full ``training_step`` (src/model.py:259-281: generator phase + discriminator
phase, both AdamW updates) on one 1x24x128^3 volume per GPU, inputs resident in HBM.  Rank 0
prints ONE JSON line (metric/unit from BASELINE.json) carrying

* ``roofline``     -- the dominant kernel family (3x3x3 implicit-GEMM conv at full resolution with 32
  output channels: ``conv_ru_kernel<1>`` in bf16, ``conv_halo_kernel<float,3,2,4,32,1>`` in f32):
  algorithmic FLOPs of all its launches / their summed duration, measured live with HIP events on
  the launch stream; peak = dense MFMA peak of the dtype (MI355X_MICROARCH.md: bf16 ~2.5 PF,
  f32-matrix 157.3 TF); ``traffic`` = HBM-side bytes per launch from rocprofv3 PMC passes of the same
  step (profiles/r01_pmc_step_traffic.txt).
* ``cpu_baseline`` -- the CPU oracle (oracle/unet_ref.py, kind "port") timed on the host cores on
  a bounded sample (one step at the same 128^3 size), rank 0 at N=1 only.  A reported baseline,
  not the target.

The Perceptual loss term (src/model.py:127-129) needs remotely fetched weights and is absent.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}
# dominant kernel family = full-resolution 3x3x3 implicit GEMM with 32 output channels
# plan id = 10000*ks + 1000*halo + 100*tile_shape + 10*voxel_subtiles_per_wave + cout_subtiles_per_wave
DOMINANT_PLAN = {"bf16": (31941, "conv_ru_kernel<1>"), "f32": (31021, "conv_halo_kernel<float,3,2,4,32,1>")}
# HBM-side bytes per launch of the dominant kernel in THIS workload (1x24x128^3 GAN step), from rocprofv3 PMC passes
# of the same step (`tools/pmc_step.sh`, profiles/r01_pmc_step_traffic.txt): 2 x FETCH_SIZE (gfx950 counts the 128-B
# requests of 16-B/lane streams at 64 B) + WRITE_SIZE, averaged over the kernel's launches.  bench.py cannot collect
# counters itself; other sizes / dtypes report null.
PROFILED_TRAFFIC = {("bf16", 128, "gan_step"): 2 * 268627.3e3 + 153600.0e3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1, help="volumes per GPU per step")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--workload", choices=["gan_step", "gen_only"], default="gan_step")
    ap.add_argument("--dropout", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of as hipGraph replays")
    return ap.parse_args()


class KernelProbe:
    """Brackets every launch of one conv kernel family with HIP events on the launch stream."""

    def __init__(self, plan_id):
        self.plan_id = plan_id
        self.events = []
        self.flops = 0.0
        self.enabled = False

    def __call__(self, plan_id, d, real):
        if not self.enabled or plan_id != self.plan_id:
            return None
        # algorithmic FLOPs of this launch: 2 * Cin * Cout * taps * positions on the REAL channel counts
        cin, cout = real if real is not None else (d.c0 + d.c1, d.cstore)
        self.flops += 2.0 * cin * cout * (d.ks ** 3) * d.n * d.do_ * d.ho * d.wo
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.events.append((e0, e1))
        return e1.record

    def summary(self):
        if not self.events:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        return dict(launches=len(self.events), total_ms=ms, avg_ms=ms / len(self.events), flops=self.flops)


def host_cores() -> int:
    """CPU cores this process may actually use (affinity mask and cgroup quota), not the machine total:
    the GPU box exposes 256 logical CPUs but grants a 16-core share per GPU."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MI355_HOST_CORES", "16"))))


def cpu_baseline(size, workload):
    from oracle import unet_ref as R
    torch.set_num_threads(host_cores())
    torch.manual_seed(0)
    gen = R.RefGenerator("bssfp", dropout=0.05).train()
    discr = R.RefDiscriminator("bssfp").train()
    g_opt, d_opt = R.make_optimizers(gen, discr)
    xs, ys = R.synthetic_batch(1, 32, seed=1)
    x2, y2 = R.synthetic_batch(2, 32, seed=1)
    R.gan_training_step(gen, discr, g_opt, d_opt, x2, y2)       # warm-up (thread pools, allocators) at 32^3
    x, y = R.synthetic_batch(1, size, seed=1234)
    t0 = time.perf_counter()
    if workload == "gan_step":
        R.gan_training_step(gen, discr, g_opt, d_opt, x, y)
    else:
        loss = torch.nn.functional.l1_loss(gen(x), y)
        loss.backward()
        g_opt.step()
        g_opt.zero_grad()
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="volumes/s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 {workload} step on one 1x24x{size}^3 volume after a 32^3 warm-up step ({dt:.1f} s), "
                       "torch CPU f32, L1 + adversarial loss (no Perceptual term)")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import unet_bssfp_amd as M
    from unet_bssfp_amd import ddp, ops
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch

    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)                                   # identical init on every rank
    gen = M.Generator("bssfp", dropout=a.dropout)
    discr = M.Discriminator("bssfp")
    model = bSSFPToDWITensorModel("bssfp", gen=gen, discr=discr).to(dev).train()
    M.set_compute_dtype(model, dtype)
    batch = synthetic_batch(a.batch, a.size, seed=1234 + rank, device=dev)   # resident in HBM
    torch.manual_seed(1000 + rank)                         # dropout seeds differ per rank
    use_graph = not a.no_graph and a.workload == "gan_step"
    gen_opt = None

    def eager_step(i):
        nonlocal gen_opt
        if a.workload == "gan_step":
            model.training_step(batch, i)
        else:                                              # BASELINE.json configs[1]: generator only
            if gen_opt is None:
                gen_opt = model.optimizers()[0]
            x, y = model.unpack_batch(batch)
            loss = M.l1_loss(model.gen(x), y)
            loss.backward()
            if model.grad_sync_gen is not None:
                model.grad_sync_gen.finish()
            gen_opt.step()
            gen_opt.zero_grad()

    probe = KernelProbe(DOMINANT_PLAN[a.dtype][0])
    mode = "eager"
    if use_graph:
        from unet_bssfp_amd.gan import GraphedTrainingStep
        if world > 1:
            ddp.broadcast_module_state(model.gen, 0)
            ddp.broadcast_module_state(model.discr, 0)
        gstep = GraphedTrainingStep(model, batch, warmup=max(2, a.warmup))    # eager warm-up + capture
        step = lambda i: gstep()
        mode = "hipgraph"
    else:
        if world > 1:
            ddp.attach(model)
        step = eager_step

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if mode == "eager" and not a.no_probe:
        ops.CONV_PROBE = probe
        probe.enabled = True
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    probe.enabled = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    probe_mode = "events around every launch of the kernel family inside the timed region"
    if mode == "hipgraph" and not a.no_probe and rank == 0 and world == 1:
        # inside a graph replay single kernels cannot be bracketed with events: time the same launches
        # (same tensors, same kernels) in an eager replica pass right after the timed region
        ops.CONV_PROBE = probe
        probe.enabled = True
        for i in range(2):
            gstep._eager_step()
        torch.cuda.synchronize()
        probe.enabled = False
        probe_mode = "eager replica pass (2 steps) right after the hipGraph-timed region"

    if rank == 0:
        vols = a.steps * a.batch * world
        ms = dt / a.steps * 1e3
        flop_per_vol = {"gan_step": 5035.4e9, "gen_only": 3486.6e9}[a.workload] * (a.size / 128.0) ** 3
        out = {
            "metric": "train-step volumes/sec (128^3 bSSFP->6ch DTI)",
            "value": vols / dt, "unit": "volumes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.workload}: full GAN training step (G fwd x2, G bwd, D fwd x3, D bwd x3, 2x AdamW)"
                       if a.workload == "gan_step" else f"{a.workload}: generator fwd+bwd, L1 loss, AdamW",
                       "volume": f"{a.batch}x24x{a.size}^3 -> 6ch per GPU", "global_batch": a.batch * world,
                       "dropout": a.dropout, "perceptual_term": "absent (needs remote weights)",
                       "parallelism": f"dp{world}"},
            "step_tflops": flop_per_vol * vols / dt / 1e12,
            "launch_mode": mode,
        }
        s = probe.summary()
        if s:
            ach = s["flops"] / (s["total_ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                               "frac": ach / PEAK_TFLOPS[a.dtype],
                               "traffic": PROFILED_TRAFFIC.get((a.dtype, a.size, a.workload)) if a.batch == 1 else None,
                               "kernel": DOMINANT_PLAN[a.dtype][1],
                               "launches": s["launches"], "avg_launch_ms": s["avg_ms"], "measured": probe_mode}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.size, a.workload)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
