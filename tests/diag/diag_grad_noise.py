"""Diagnostic (GPU box): per-parameter gradient error of the f32 HIP path and of the CPU f32 oracle against the f64 oracle,
generator phase and discriminator phase of training step 0 at 64^3 (same weights, same batch).  Tells whether the HIP f32
gradients are noisier than the CPU's, layer by layer.  usage: python tests/diag/diag_grad_noise.py [size]"""
import copy
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import unet_ref as R  # noqa: E402
import unet_bssfp_amd as M  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
DEV = "cuda:0"
torch.set_num_threads(16)
torch.manual_seed(0)
gen, discr = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
rg, rd = R.RefGenerator("bssfp", dropout=0.0).train(), R.RefDiscriminator("bssfp").train()
rg.load_state_dict(gen.state_dict())
rd.load_state_dict(discr.state_dict())
dg, dd = copy.deepcopy(rg).double(), copy.deepcopy(rd).double()
gen, discr = gen.to(DEV).train(), discr.to(DEV).train()
x, y = R.synthetic_batch(1, S, seed=1234)


def phases(g, d, x, y, l1):
    out = {}
    for p in d.parameters():
        p.requires_grad_(False)
    y_hat = g(x)
    logits = d(x, y_hat)
    loss = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits)) + l1(y_hat, y) * 100.0
    loss.backward()
    out["gen"] = {n: p.grad.detach().double().cpu() for n, p in g.named_parameters() if p.grad is not None}
    for p in d.parameters():
        p.requires_grad_(True)
    for p in g.parameters():
        p.grad = None
    yh = y_hat.detach()
    lf, lr = d(x, yh), d(x, y)
    dl = (F.binary_cross_entropy_with_logits(lr, torch.ones_like(lr)) + F.binary_cross_entropy_with_logits(lf, torch.zeros_like(lf))) / 2
    dl.backward()
    out["discr"] = {n: p.grad.detach().double().cpu() for n, p in d.named_parameters() if p.grad is not None}
    return out, float(loss), float(dl)


hip, lh, dh = phases(gen, discr, x.to(DEV), y.to(DEV), M.l1_loss)
cpu, lc, dc = phases(rg, rd, x, y, F.l1_loss)
f64, l6, d6 = phases(dg, dd, x.double(), y.double(), F.l1_loss)
print(f"gen loss hip {lh:.8f} cpu {lc:.8f} f64 {l6:.8f} | discr loss hip {dh:.8f} cpu {dc:.8f} f64 {d6:.8f}")
for net in ("gen", "discr"):
    for n, g6 in f64[net].items():
        if n.startswith("blocks.") and net == "discr":
            continue
        nz = g6.norm().item()
        if nz == 0:
            continue
        eh = ((hip[net][n] - g6).norm() / nz).item() if n in hip[net] else float("nan")
        ec = ((cpu[net][n] - g6).norm() / nz).item()
        fh = (torch.sign(hip[net][n]) != torch.sign(g6)).float().mean().item() if n in hip[net] else float("nan")
        fc = (torch.sign(cpu[net][n]) != torch.sign(g6)).float().mean().item()
        flag = "  <<<" if eh > 5 * ec + 1e-7 else ""
        print(f"{net:5s} {n:52s} relerr hip {eh:.2e} cpu {ec:.2e} | sign flips hip {fh:.2e} cpu {fc:.2e}{flag}")
