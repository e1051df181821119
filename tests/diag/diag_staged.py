"""Diagnostic: gradients of one eager training step, plain backward vs two-stage backward (single rank)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import unet_bssfp_amd as M
from unet_bssfp_amd.functional import DropoutState
from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
DEV = "cuda:0"
batch = synthetic_batch(2, 32, seed=9, device=DEV)
res = []
for staged in (False, True):
    torch.manual_seed(4); DropoutState.reset()
    m = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.05).to(DEV), discr=M.Discriminator("bssfp").to(DEV)).train()
    logs = {}
    m._phase_gen(batch, logs, staged=staged)
    if staged:
        print("boundary", [tuple(t.shape) for t in m._stage[0]], "early params", len(m._stage[2]))
        m._backward_early()
    torch.cuda.synchronize()
    g_gen = {n: p.grad.detach().clone() for n, p in m.gen.named_parameters() if p.grad is not None}
    m._update_gen()
    m._phase_discr(batch, logs, staged=staged)
    if staged:
        print("boundary", [tuple(t.shape) for t in m._stage[0]], "early params", len(m._stage[2]))
        m._backward_early()
    torch.cuda.synchronize()
    g_d = {n: p.grad.detach().clone() for n, p in m.discr.named_parameters() if p.grad is not None}
    res.append((g_gen, g_d, {k: float(v) for k, v in logs.items()}))
print(res[0][2]); print(res[1][2])
for which in (0, 1):
    a, b = res[0][which], res[1][which]
    for n in a:
        if n not in b: print("missing", n); continue
        if not torch.equal(a[n], b[n]):
            d = (a[n] - b[n]).abs().max().item(); print(("gen " if which == 0 else "discr ") + n, "max diff", d, "ref max", a[n].abs().max().item())
print("done")
