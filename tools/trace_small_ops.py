"""Attribute the small torch-native launches of one eager training step to Python call sites (torch.profiler with stacks)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_bssfp_amd as M
from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = bSSFPToDWITensorModel("bssfp").to(dev).train()
M.set_compute_dtype(model, torch.bfloat16)
batch = synthetic_batch(1, 64, seed=1, device=dev)
for _ in range(2):
    model.training_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    model.training_step(batch)
torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::full", "aten::contiguous", "aten::clone")
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        st = [s for s in ev.stack if "unet_bssfp_amd" in s or "bench" in s or "autograd" in s.lower()]
        key = (ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"), str(ev.input_shapes)[:60])
        cnt[key] += 1
for (name, where, shp), n in cnt.most_common(60):
    print(f"{n:4d} {name:18s} {where[-90:]:90s} {shp}")
