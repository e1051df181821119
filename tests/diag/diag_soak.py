"""Soak: several hundred replays of the graphed GAN step on one batch; the losses must stay finite and the L1 term must fall."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import unet_bssfp_amd as M
from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch

dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
torch.manual_seed(0)
model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp").to(dev), discr=M.Discriminator("bssfp").to(dev)).train()
M.set_compute_dtype(model, mode)
g = GraphedTrainingStep(model, synthetic_batch(1, size, seed=1, device=dev), warmup=2)
for i in range(steps):
    g()
    if i % 50 == 0 or i == steps - 1:
        logs = {k: float(v) for k, v in model.last_logs.items()}
        assert all(v == v and abs(v) < 1e6 for v in logs.values()), (i, logs)
        print(i, " ".join(f"{k.replace('train_', '')}={v:.4f}" for k, v in logs.items()), flush=True)
print("soak ok", mode, size, steps)
