#!/usr/bin/env python3
"""Benchmark of the hot path: GAN training-step volumes/s on synthetic 128^3 bSSFP -> 6-ch DTI.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full ``training_step`` (src/model.py:259-281: generator phase + discriminator
phase, both AdamW updates) on one 1x24x128^3 volume per GPU, inputs resident in HBM.  W untimed
warm-up steps, then (untimed, reported as ``settle_steps``) further steps until the GPU has been
busy for ``--settle-s`` seconds so that the clock the chip holds under sustained MFMA load is reached,
then EXACTLY K timed steps between barrier + synchronize pairs; MAX over ranks.  Rank 0 prints ONE JSON
line (metric/unit from BASELINE.json) carrying

* ``roofline``     -- the dominant kernel family = the convolution plan with the largest summed duration
  (measured, not assumed): algorithmic FLOPs of all its launches / their summed duration, HIP events on
  the launch stream around every launch; peak = dense MFMA peak of the dtype (MI355X_MICROARCH.md:
  bf16 ~2.5 PF, f32-matrix 157.3 TF, fp8 ~5 PF).  ``traffic`` = HBM-side bytes per launch from rocprofv3 PMC
  passes (separate --pmc runs, 2 x FETCH_SIZE + WRITE_SIZE: gfx950 counts 16-B/lane read streams at half
  their bytes) recorded in profiles/*_traffic.json together with a hash of the kernel sources; it is
  reported only when that hash matches the sources this run was built from, else null.
* ``roofline_hbm`` -- the fused norm + dropout + LeakyReLU kernels (forward / backward): algorithmic bytes
  per SURVEY 8(d) (forward 2*C*V*b, backward 3*C*V*b) / event time, against 8 TB/s.
* ``cpu_baseline`` -- the CPU oracle (oracle/unet_ref.py, kind "port") timed on the host cores on
  a bounded sample (3 steps at the same 128^3 size after a warm-up), rank 0 at N=1 only.  A reported
  baseline, not the target.

``--fresh-batch`` feeds a NEW host batch to every step (pinned double buffer -> staging buffer on a side
stream under the previous step -> device-to-device copy into the graph's static inputs).

The Perceptual loss term (src/model.py:127-129) needs remotely fetched weights and is absent.
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}
HBM_PEAK_GBS = 8000.0
# plan id = 10000*ks + 1000*halo + 100*tile_shape + 10*voxel_subtiles_per_wave + cout_subtiles_per_wave
PLAN_NAMES = {22421: "conv_march2_kernel", 32041: "conv_march_kernel", 32141: "conv_marchg_kernel<4>", 32121: "conv_marchg_kernel<2>", 31941: "conv_ru_kernel<1>", 31942: "conv_ru_kernel<2>", 31021: "conv_halo_kernel<float,3,2,4,32,1>",
              31022: "conv_halo_kernel<float,3,2,4,32,2>", 31411: "conv_halo_kernel<bf16_t,3,2,4,16,1,4>"}
KERNEL_SOURCES = ("conv_march.h", "conv_marchg.h", "conv_march2.h", "conv_kernels.h", "conv_common.h", "conv_api.hip", "common.h")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-s", type=float, default=2.0, help="untimed extra steps until the GPU has run this long")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1, help="volumes per GPU per step")
    ap.add_argument("--dtype", choices=["bf16", "f32", "fp8"], default="bf16")
    ap.add_argument("--workload", choices=["gan_step", "gen_only"], default="gan_step")
    ap.add_argument("--dropout", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend at world size > 1 (nccl = RCCL; gloo: rehearsal of the multi-rank path, ranks may share a GPU)")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of as hipGraph replays")
    ap.add_argument("--fresh-batch", action="store_true", help="a new host batch every step (H2D hidden on a side stream)")
    ap.add_argument("--fresh-copy-after", type=int, default=None,
                    help="with --fresh-batch: run the step as its five graph segments and start the next batch's copy behind segment k (0..3)")
    ap.add_argument("--fresh-diag", default=None, choices=["nowait_ready", "nowait_done", "nowait_both", "x_only", "y_only", "gpu_wait"],
                    help="developer, timing only: --fresh-batch without one of its stream waits, or copying only x (201 MB) / only y (50 MB)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="one rank, real process group: run the FIVE-segment graph step and issue its four all-reduces "
                         "(rehearsal of process group + watchdog + thread-local capture on a single GPU)")
    ap.add_argument("--bn-broadcast-every", type=int, default=1,
                    help="world size > 1: broadcast rank 0's BatchNorm running statistics before every k-th step "
                         "(DDP's broadcast_buffers=True, src/train.py:30; 0 = never)")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch (--gpus N without torch.distributed.run): rendezvous port (0 = pick a free one)")
    ap.add_argument("--eager-final-dx", action="store_true", help="developer A/B: the final 1x1x1 convolution writes its data gradient (one launch, 134 MB) instead of leaving it to the norm backward kernels (Fn.LazyDx); reported")
    ap.add_argument("--separate-colsum", action="store_true", help="developer A/B: transposed-conv bias gradients by a separate pass over the gradient instead of the producing launch's statistics; reported")
    ap.add_argument("--immediate-reduce", action="store_true", help="developer A/B: every weight-gradient launch followed by its own slab reduction instead of the batched reduction at the end of a backward pass; reported")
    ap.add_argument("--eager-pool-bwd", action="store_true", help="developer A/B: MaxPool3d's backward as its own launch instead of inside the producing norm node's backward kernels; reported")
    ap.add_argument("--separate-pool", action="store_true", help="developer A/B: MaxPool3d's forward as its own launch instead of inside the producing norm + act launch; reported")
    ap.add_argument("--wgrad-first", action="store_true", help="developer A/B: a convolution's backward launches its weight gradient before its data gradient; reported")
    ap.add_argument("--composed-losses", action="store_true", help="developer A/B: BCE / L1 loss heads as composed torch ops instead of the one-launch kernels; reported")
    ap.add_argument("--side-stream", action="store_true", help="developer A/B: weight gradients of the small layers on a second stream (measured slower); reported")
    ap.add_argument("--small-norm-grouped", type=int, default=None, help="developer A/B: the same limit for BatchNorm tensors whose statistic groups one workgroup walks in order (forward_pair)")
    ap.add_argument("--small-norm-elements", type=int, default=None,
                    help="developer A/B: size limit of the one-launch norm kernels (0 = off); reported")
    ap.add_argument("--lib", default=None, help="developer A/B: another build of the C ABI (tools/diaglib.py); reported")
    return ap.parse_args()


class Probe:
    """Brackets launches with HIP events on the launch stream (torch's current stream = the stream the C ABI
    launches on).  conv launches are grouped by plan id, norm launches by direction."""

    def __init__(self):
        self.enabled = False
        self.overhead_ms = 0.0  # what an empty event bracket reads (subtracted per launch): see calibrate()
        self.conv = {}          # (plan id, operand dtype) -> [events, flops]
        self.conv_bytes = {}    # same key -> algorithmic bytes (input read once + output written once, stored widths)
        self.norm = {}          # "fwd" / "bwd" -> [events, algorithmic bytes]

    def _bracket(self, store, key, work):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ent = store.setdefault(key, [[], 0.0])
        ent[0].append((e0, e1))
        ent[1] += work
        e0.record()
        return e1.record

    def conv_probe(self, plan_id, d, real):
        if not self.enabled:
            return None
        cin, cout = real if real is not None else (d.c0 + d.c1, d.cstore)
        key = (plan_id, int(d.dtype))
        eb_in = 1 if int(d.dtype) == 3 else (4 if int(d.dtype) == 0 else 2)     # e4m3 operands, bf16 / f32 otherwise
        eb_out = 4 if int(d.dtype) == 0 else 2
        self.conv_bytes[key] = self.conv_bytes.get(key, 0.0) + float(d.n) * (
            (d.c0 + d.c1) * d.di * d.hi * d.wi * eb_in + d.cstore * d.do_ * d.ho * d.wo * eb_out)
        return self._bracket(self.conv, key, 2.0 * cin * cout * (d.ks ** 3) * d.n * d.do_ * d.ho * d.wo)

    def norm_probe(self, kind, c_real, rows, elem_bytes):
        if not self.enabled:
            return None
        passes = 2 if kind == "fwd" else 3           # fwd read z + write a; bwd read dy, read z, write dz
        return self._bracket(self.norm, kind, float(passes) * c_real * rows * elem_bytes)

    def calibrate(self, n=64):
        """An event pair with nothing between its records still reads a few microseconds; with 23 short launches in one
        family and 12 long ones in another that bias decides which family looks dominant.  Median of n empty brackets."""
        pairs = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        self.overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[n // 2]

    passes = 1      # identical step passes the brackets were collected over (3 for the eager replica after a graph run)

    def _sum(self, ent):
        """Per family: the MEAN over the probe passes of every launch's reading (what a profile's average duration shows:
        ``avg_ms`` / ``total_ms``), the per-launch MINIMUM separately (``best_ms``), and the number of readings set aside
        as one-off stalls (an allocator call or a page fault inside a bracket read 70 ms once): the i-th launch of a family
        is the same launch in every pass, so a reading above 3x that launch's own minimum is flagged and not averaged."""
        t = [max(0.0, a.elapsed_time(b) - self.overhead_ms) for a, b in ent[0]]
        n = len(t) // max(1, self.passes)
        flagged = 0
        if self.passes > 1 and n * self.passes == len(t):
            mean_t, best_t = [], []
            for i in range(n):
                r = [t[i + k * n] for k in range(self.passes)]
                lo = min(r)
                keep = [v for v in r if v <= 3.0 * lo + 1e-3]
                flagged += len(r) - len(keep)
                mean_t.append(sum(keep) / len(keep))
                best_t.append(lo)
            ms, best = sum(mean_t) * self.passes, sum(best_t) * self.passes
        else:
            ms = best = sum(t)
        nl = max(1, len(ent[0]))
        return dict(launches=len(ent[0]), total_ms=ms, avg_ms=ms / nl, best_ms=best / nl, work=ent[1], flagged=flagged)

    def dominant_conv(self, only_dtype=None):
        """(plan id, operand dtype code), summary of the convolution kernel family with the largest summed duration;
        ``only_dtype``: among the launches with that operand type (``--dtype fp8``: the e4m3 kernel is the one the
        configuration is about; the layers it does not take run on bf16 operands and are listed by share)."""
        best = None
        for pid, ent in self.conv.items():
            if only_dtype is not None and pid[1] != only_dtype:
                continue
            s = self._sum(ent)
            if best is None or s["total_ms"] > best[1]["total_ms"]:
                best = (pid, s)
        return best

    def norm_summary(self):
        return {k: self._sum(v) for k, v in self.norm.items()}


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


def cpu_baseline(size, workload, nsteps):
    from oracle import unet_ref as R
    torch.set_num_threads(host_cores())
    torch.manual_seed(0)
    gen = R.RefGenerator("bssfp", dropout=0.05).train()
    discr = R.RefDiscriminator("bssfp").train()
    g_opt, d_opt = R.make_optimizers(gen, discr)
    x2, y2 = R.synthetic_batch(2, 32, seed=1)
    R.gan_training_step(gen, discr, g_opt, d_opt, x2, y2)       # warm-up (thread pools, allocators) at 32^3
    x, y = R.synthetic_batch(1, size, seed=1234)
    times = []
    for _ in range(max(1, nsteps)):
        t0 = time.perf_counter()
        if workload == "gan_step":
            R.gan_training_step(gen, discr, g_opt, d_opt, x, y)
        else:
            loss = torch.nn.functional.l1_loss(gen(x), y)
            loss.backward()
            g_opt.step()
            g_opt.zero_grad()
        times.append(time.perf_counter() - t0)
    mean = sum(times) / len(times)
    return dict(value=1.0 / mean, unit="volumes/s", cores=torch.get_num_threads(), kind="port",
                n_steps=len(times), s_per_step_min=min(times), s_per_step_max=max(times),
                sample=f"{len(times)} {workload} steps on one 1x24x{size}^3 volume after a 32^3 warm-up step "
                       f"({mean:.1f} s per step, min {min(times):.1f} / max {max(times):.1f}), torch CPU f32, "
                       "L1 + adversarial loss (no Perceptual term)")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "unet_bssfp_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel, dtype, size, workload, batch):
    """HBM bytes per launch of `kernel` from the newest profiles/*_traffic.json entry whose recorded kernel-source hash
    equals the sources of this build; (None, reason) otherwise."""
    src = kernel_source_hash()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in rec.get("entries", []):
            if (e.get("kernel"), e.get("dtype"), e.get("size"), e.get("workload"), e.get("batch", 1)) != (kernel, dtype, size, workload, batch):
                continue
            if rec.get("kernel_source_sha16") == src:
                return e["hbm_bytes_per_launch"], dict(file=os.path.relpath(path, ROOT), git_head=rec.get("git_head"),
                                                       kernel_source_sha16=src, method=rec.get("method"))
            stale = os.path.relpath(path, ROOT)
    return None, dict(reason=("kernel sources changed since " + stale) if stale else "no PMC record for this kernel / workload",
                      kernel_source_sha16=src)


class FreshBatches:
    """A new host batch per step (src/data_module.py:185-188 delivers one every step) without a staging buffer: the step is
    captured over TWO static input sets (GraphedTrainingStep.add_instance) and runs on them alternately; while instance k
    replays, the next host batch crosses PCIe on a side stream straight into the inputs of instance 1 - k, which finished its
    last replay before instance k started.  No device-to-device copy at the step boundary, and the only cross-stream waits are
    on events that completed a whole step earlier."""

    def __init__(self, gstep, nbuf=2):
        self.gstep = gstep
        self.stream = torch.cuda.Stream()
        first = gstep.batch
        second = {}
        cache = {}
        for k, v in first.items():                               # 'dwi-tensor' and 'dwi-tensor_orig' share one tensor
            if isinstance(v, dict) and "data" in v:
                t = v["data"]
                if t.data_ptr() not in cache:
                    cache[t.data_ptr()] = t.clone()
                second[k] = {"data": cache[t.data_ptr()]}
            else:
                second[k] = v
        gstep.add_instance(second)
        self.sets = []
        for b in (first, second):
            uniq, seen = [], set()
            for k, v in b.items():
                if isinstance(v, dict) and "data" in v and v["data"].data_ptr() not in seen:
                    seen.add(v["data"].data_ptr())
                    uniq.append(v["data"])
            self.sets.append(uniq)
        g = torch.Generator().manual_seed(4321)
        self.host = [[torch.rand(t.shape, generator=g).pin_memory() for t in self.sets[0]] for _ in range(nbuf)]
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]    # the host batch has landed in input set k
        self.done = [torch.cuda.Event(), torch.cuda.Event()]     # the latest replay over input set k has finished
        cur = torch.cuda.current_stream()
        for k in (0, 1):
            self.done[k].record(cur)
        self.i = 0
        self.diag = None
        self.mark = torch.cuda.Event()
        self._copy_into(0)                                       # the first step's batch

    copy_after = None                                            # segment index behind which the next batch's copy starts

    def _copy_into(self, k, also_wait=None):
        with torch.cuda.stream(self.stream):
            if also_wait is not None:
                self.stream.wait_event(also_wait)
            # input set k must no longer be read: its last replay has FINISHED.  The host waits for that (it is one step ahead
            # of the GPU and has nothing else to do); a GPU-side wait of the copy stream on an event recorded between two
            # graph launches cost ~0.65 ms per step whatever the copy's size (measured: --fresh-diag gpu_wait)
            if self.diag == "gpu_wait":
                self.stream.wait_event(self.done[k])
            elif self.diag not in ("nowait_done", "nowait_both"):
                self.done[k].synchronize()
            for j, (t, h) in enumerate(zip(self.sets[k], self.host[self.i % len(self.host)])):
                if (self.diag == "x_only" and j != 0) or (self.diag == "y_only" and j == 0):
                    continue                                     # (timing-only: how does the cost scale with the bytes copied?)
                t.copy_(h, non_blocking=True)
            self.ready[k].record(self.stream)
        self.i += 1

    def step(self, i):
        k = i & 1
        cur = torch.cuda.current_stream()
        if self.diag not in ("nowait_ready", "nowait_both"):
            cur.wait_event(self.ready[k])
        if self.copy_after is None:
            self.gstep(k)
            self.done[k].record(cur)
            # (after the launch: enqueueing a 250 MB host-to-device copy can hold the CPU thread for milliseconds, which
            #  must pass while the GPU is busy with the step, not in front of it)
            self._copy_into(1 - k)
        else:
            # segmented step: the copy starts once segment `copy_after` of THIS step has run (an event on the main stream)
            def after(seg):
                if seg == self.copy_after:
                    self.mark.record(cur)
                    self._copy_into(1 - k, also_wait=self.mark)
            self.gstep(k, after)
            self.done[k].record(cur)


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(a) -> int:
    """``python bench.py --gpus N`` without an external launcher: N rank processes via ``python -m torch.distributed.run``
    started as a CHILD of this process, which has not initialised the GPU and does not (a process that has must never be
    replaced by exec, and is not here); rank 0's JSON line passes through on stdout."""
    import subprocess
    port = a.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required for RCCL between processes on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // max(1, a.gpus))))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        # self-launch: this parent has not touched the GPU (no torch.cuda call so far) and never does -- it starts the
        # ranks as fresh child processes through torch.distributed.run and relays rank 0's JSON line and the exit code
        sys.exit(self_launch(a))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if a.backend != "nccl":
        local = local % torch.cuda.device_count()        # rehearsal of the multi-rank path on fewer GPUs than ranks (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    real_stdout = None
    if world > 1 or a.force_collectives:
        # RCCL ("RCCL version : ...") and gloo ("[Gloo] Rank 0 is connected to ...") print banners on stdout: the contract is
        # ONE JSON line there, so file descriptor 1 points at stderr for the run and the line is written to the saved one
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(a.master_port or free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    # MI355_* plan knobs are read by DIAGNOSTIC builds only (tools/build_diag.sh: -DMI355_DIAG); the shipped library ignores
    # them, so they are recorded as non-default settings only when such a build is loaded (--lib)
    nondefault = ({k: v for k, v in os.environ.items() if k.startswith("MI355_") and k != "MI355_HOST_CORES"} if a.lib else {})
    if a.backend != "nccl":
        nondefault["backend"] = a.backend
    if a.force_collectives:
        nondefault["force_collectives"] = True
    if a.fresh_diag:
        nondefault["fresh_diag"] = a.fresh_diag
    if a.fresh_copy_after is not None:
        nondefault["fresh_copy_after"] = a.fresh_copy_after
    if a.lib:
        from tools import diaglib
        nondefault["lib"] = diaglib.use(a.lib)

    import unet_bssfp_amd as M
    from unet_bssfp_amd import ddp, ops
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch

    if a.side_stream:
        from unet_bssfp_amd import functional as _Fn
        _Fn.SideStream.allowed = True
        nondefault["side_stream"] = True
    if a.eager_final_dx:
        from unet_bssfp_amd import functional as _Fn3
        _Fn3.LazyDx.enabled = False
        nondefault["eager_final_dx"] = True
    if a.eager_pool_bwd:
        from unet_bssfp_amd import functional as _Fn5
        _Fn5.LazyPool.enabled = False
        nondefault["eager_pool_bwd"] = True
    if a.separate_pool:
        from unet_bssfp_amd import functional as _Fn6
        _Fn6.PoolSide.enabled = False
        nondefault["separate_pool"] = True
    if a.wgrad_first:
        from unet_bssfp_amd import functional as _Fn7
        _Fn7.ConvFn.wgrad_first = True
        nondefault["wgrad_first"] = True
    if a.immediate_reduce:
        from unet_bssfp_amd import functional as _Fn4
        _Fn4.DeferredReduce.allowed = False
        nondefault["immediate_reduce"] = True
    if a.separate_colsum:
        from unet_bssfp_amd import functional as _Fn2
        _Fn2.ColSumSide.enabled = False
        nondefault["separate_colsum"] = True
    if a.small_norm_elements is not None:
        ops.SMALL_NORM_ELEMENTS = a.small_norm_elements
        nondefault["small_norm_elements"] = a.small_norm_elements
    if a.small_norm_grouped is not None:
        ops.SMALL_NORM_ELEMENTS_GROUPED = a.small_norm_grouped
        nondefault["small_norm_grouped"] = a.small_norm_grouped
    dtype = M.compute_dtype_from_name(a.dtype)
    torch.manual_seed(0)                                   # identical init on every rank
    gen = M.Generator("bssfp", dropout=a.dropout)
    discr = M.Discriminator("bssfp")
    model = bSSFPToDWITensorModel("bssfp", gen=gen, discr=discr).to(dev).train()
    M.set_compute_dtype(model, dtype)
    if a.composed_losses:
        model.fused_loss_heads = False
        nondefault["composed_losses"] = True
    batch = synthetic_batch(a.batch, a.size, seed=1234 + rank, device=dev)   # resident in HBM
    torch.manual_seed(1000 + rank)                         # dropout seeds differ per rank
    use_graph = not a.no_graph and a.workload == "gan_step"

    def eager_step(i):
        if a.workload == "gan_step":
            model.training_step(batch, i)
        else:                                              # BASELINE.json configs[1]: generator only
            model.generator_only_step(batch, i)

    probe = Probe()
    mode = "eager"
    fresh = None
    if use_graph:
        from unet_bssfp_amd.gan import GraphedTrainingStep
        if world > 1:
            ddp.broadcast_module_state(model.gen, 0)
            ddp.broadcast_module_state(model.discr, 0)
        # 2 eager steps (allocations, caches, optimiser state) + capture
        gstep = GraphedTrainingStep(model, batch, warmup=2, force_collectives=a.force_collectives,
                                    force_segments=a.fresh_batch and a.fresh_copy_after is not None,
                                    broadcast_buffers_every=a.bn_broadcast_every)
        if a.fresh_batch:
            fresh = FreshBatches(gstep)
            fresh.diag = a.fresh_diag
            fresh.copy_after = a.fresh_copy_after

        def step(i):
            if fresh is not None:
                fresh.step(i)
            else:
                gstep()
        mode = "hipgraph"
    else:
        if world > 1:
            ddp.attach(model)

        def step(i):
            if world > 1:
                ddp.broadcast_buffers(model, every=a.bn_broadcast_every, step=i)
            eager_step(i)

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    # settle: keep the chip under the same load until it has been busy for --settle-s (clock / thermal steady state)
    settle_steps = 0
    t_s = time.perf_counter()
    while True:
        more = time.perf_counter() - t_s < a.settle_s
        if world > 1:
            # every rank must run the SAME number of steps (each step holds collectives): go on while any rank wants to
            flag = torch.tensor([1.0 if more else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            more = bool(flag.item() > 0.0)
        if not more:
            break
        for _ in range(5):
            step(a.warmup + settle_steps)
            settle_steps += 1
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if mode == "eager" and not a.no_probe:
        ops.CONV_PROBE, ops.NORM_PROBE = probe.conv_probe, probe.norm_probe
        probe.enabled = True
    half = a.steps // 2
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(a.steps):
        if i == half:
            ev[1].record()
        step(a.warmup + settle_steps + i)
    ev[2].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    probe.enabled = False
    first_ms = ev[0].elapsed_time(ev[1]) / max(1, half)
    last_ms = ev[1].elapsed_time(ev[2]) / max(1, a.steps - half)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    probe_mode = "events around every launch of the kernel family inside the timed region"
    if mode == "hipgraph" and not a.no_probe and rank == 0 and world == 1:
        # inside a graph replay single kernels cannot be bracketed with events: time the same launches
        # (same tensors, same kernels) in an eager replica pass right after the timed region
        gstep._eager_step()                 # unprobed: first eager pass after the replays (allocator, lazy state)
        torch.cuda.synchronize()
        ops.CONV_PROBE, ops.NORM_PROBE = probe.conv_probe, probe.norm_probe
        probe.enabled = True
        probe.passes = 3
        for i in range(probe.passes):
            gstep._eager_step()
        torch.cuda.synchronize()
        probe.enabled = False
        probe_mode = "eager replica passes (3 steps) right after the hipGraph-timed region; mean over the passes per launch"
    ops.CONV_PROBE = ops.NORM_PROBE = None
    if not a.no_probe:
        probe.calibrate()

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
                       "parallelism": f"dp{world}",
                       "bn_buffer_broadcast_every": (a.bn_broadcast_every if world > 1 else None),
                       "input_feed": "new host batch every step: host-to-device copy on a side stream into the idle one of two static input sets" if fresh is not None else "one batch resident in HBM"},
            "step_tflops": flop_per_vol * vols / dt / 1e12,
            "launch_mode": mode,
            "timed_region_s": dt, "settle_steps": settle_steps,
            "ms_per_step_first_half": first_ms, "ms_per_step_last_half": last_ms,
            "nondefault": nondefault or None,
        }
        dom = probe.dominant_conv(3 if a.dtype == "fp8" else None)
        if dom:
            (pid, dcode), s = dom
            name = PLAN_NAMES.get(pid, f"conv plan {pid}") + {3: "<e4m3>", 1: "", 2: "", 0: ""}.get(dcode, "")
            ach = s["work"] / (s["total_ms"] * 1e-3) / 1e12
            traffic, tinfo = profiled_traffic(name, a.dtype, a.size, a.workload, a.batch)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                               "frac": ach / PEAK_TFLOPS[a.dtype], "traffic": traffic, "traffic_source": tinfo,
                               "algorithmic_bytes": probe.conv_bytes.get((pid, dcode), 0.0) / max(1, s["launches"]),
                               "traffic_over_algorithmic": (traffic / (probe.conv_bytes.get((pid, dcode), 0.0) / max(1, s["launches"]))) if traffic else None,
                               "kernel": name, "launches": s["launches"], "avg_launch_ms": s["avg_ms"], "best_launch_ms": s["best_ms"],
                               "frac_best": s["work"] / (s["best_ms"] * s["launches"] * 1e-3) / 1e12 / PEAK_TFLOPS[a.dtype] if s["best_ms"] > 0 else None,
                               "flagged_brackets": s["flagged"],
                               "share_of_conv_time": s["total_ms"] / max(1e-9, sum(probe._sum(e)["total_ms"] for e in probe.conv.values())),
                               "conv_families": sorted(({"plan": PLAN_NAMES.get(k[0], str(k[0])), "operands": {0: "f32", 1: "bf16", 3: "e4m3"}.get(k[1], str(k[1])),
                                                          "launches_per_step": v["launches"] // max(1, probe.passes),
                                                          "ms_per_step": v["total_ms"] / max(1, probe.passes),
                                                          "tflops": v["work"] / max(1e-9, v["total_ms"] * 1e-3) / 1e12, "flagged": v["flagged"]}
                                                         for k, v in ((k, probe._sum(e)) for k, e in probe.conv.items())),
                                                        key=lambda r: -r["ms_per_step"]),
                               "measured": probe_mode, "event_bracket_overhead_us": probe.overhead_ms * 1e3}
        ns = probe.norm_summary()
        if ns:
            hb = {}
            for kind, s in ns.items():
                gbs = s["work"] / (s["total_ms"] * 1e-3) / 1e9
                hb[kind] = {"kernel": "normact_fwd_kernel" if kind == "fwd" else "normact_bwd (reduce + apply)",
                            "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "launches": s["launches"], "total_ms": s["total_ms"] / max(1, probe.passes),
                            "algorithmic_bytes": "2*C*V*b (read z, write a)" if kind == "fwd" else "3*C*V*b (read dy, read z, write dz; the two-kernel form executes 5 passes)"}
            out["roofline_hbm"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": hb, "measured": probe_mode}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.size, a.workload, a.cpu_steps)
        if real_stdout is not None:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
