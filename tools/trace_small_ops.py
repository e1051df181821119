"""Attribute the small torch launches of one eager training step to Python call sites: wraps a few Tensor methods /
torch functions and counts (op, caller) pairs (calls made inside the autograd engine's C++ side are not seen)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_bssfp_amd as M
from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = bSSFPToDWITensorModel("bssfp").to(dev).train()
M.set_compute_dtype(model, torch.bfloat16)
batch = synthetic_batch(1, 64, seed=1, device=dev)
for _ in range(2):
    model.training_step(batch)
torch.cuda.synchronize()
cnt = collections.Counter()
ACTIVE = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "unet_bssfp_amd" in fr.filename or "optim" in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ACTIVE[0]:
            t = a[0] if a and isinstance(a[0], torch.Tensor) else None
            shp = tuple(t.shape) if t is not None else (a[0] if a else None)
            cnt[(f"{getattr(owner, '__name__', owner)}.{name}", site(), str(shp)[:40])] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for nm in ("copy_", "add_", "zero_", "fill_", "contiguous", "clone", "mul_", "float", "to", "__add__", "__iadd__", "__mul__", "__truediv__", "sum", "mean"):
    wrap(torch.Tensor, nm)
for nm in ("zeros", "ones_like", "zeros_like", "full", "stack", "cat", "empty_like"):
    wrap(torch, nm)
ACTIVE[0] = True
model.training_step(batch)
ACTIVE[0] = False
torch.cuda.synchronize()
for (op, where, shp), n in cnt.most_common(70):
    print(f"{n:4d} {op:28s} {where:28s} {shp}")
