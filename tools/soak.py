"""Soak run of the graphed GAN training step: N replays on a rotating set of synthetic batches, losses sampled every K steps,
finiteness of every parameter at the end, and (fp8) the state of the delayed-scaling slots.
usage: python tools/soak.py [--dtype bf16|fp8] [--size 128] [--steps 1500] [--every 100]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_bssfp_amd as M
from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--every", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.05), discr=M.Discriminator("bssfp")).to(dev).train()
M.set_compute_dtype(model, a.dtype)
batches = [synthetic_batch(1, a.size, seed=100 + i, device=dev) for i in range(4)]
gs = GraphedTrainingStep(model, batches[0], warmup=2)
t0 = time.perf_counter()
for i in range(a.steps):
    if i % 25 == 0:                                    # a new volume every 25 steps (copied into the static inputs)
        gs.load(batches[(i // 25) % 4])
    gs()
    if (i + 1) % a.every == 0:
        torch.cuda.synchronize()
        logs = {k: float(v) for k, v in model.last_logs.items()}
        print(f"step {i + 1:5d}  " + "  ".join(f"{k.replace('train_', '')}={v:.4f}" for k, v in logs.items()), flush=True)
        assert all(v == v and abs(v) < 1e6 for v in logs.values()), logs
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
print(f"{a.steps} steps in {dt:.1f} s ({dt / a.steps * 1e3:.2f} ms per step incl. the input copies); non-finite parameters: {bad or 'none'}")
if a.dtype == "fp8":
    from unet_bssfp_amd.functional import Fp8Scales
    for table, slots, sat in Fp8Scales._chunks.get(dev, []):
        t = table[: len(slots)].cpu()
        print("delayed-scaling slots (amax in use):", " ".join(f"{float(v):.3g}" for v in t[:, 0]))
        print("steps in which a slot saturated (a value clamped at +-448):", " ".join(str(int(v)) for v in sat[: len(slots)].cpu()))
assert not bad
