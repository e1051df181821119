"""Diagnostic (not a test): element-wise gradient / update agreement GPU vs oracle for one GAN step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from oracle import unet_ref as R
import unet_bssfp_amd as M

DEV = "cuda:0"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.manual_seed(0)
gen, discr = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
rgen, rdiscr = R.RefGenerator("bssfp", dropout=0.0).train(), R.RefDiscriminator("bssfp").train()
rgen.load_state_dict(gen.state_dict()); rdiscr.load_state_dict(discr.state_dict())
gen, discr = gen.to(DEV).train(), discr.to(DEV).train()
x, y = R.synthetic_batch(N, S, seed=1234)
xd, yd = x.to(DEV), y.to(DEV)

def report(tag, net, rnet):
    worst = []
    for (n, p), (_, q) in zip(net.named_parameters(), rnet.named_parameters()):
        if p.grad is None or q.grad is None:
            continue
        a, b = p.grad.cpu().double(), q.grad.double()
        scale = b.abs().max().item() + 1e-30
        err = (a - b).abs().max().item() / scale
        rel_l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
        worst.append((rel_l2, err, n, scale))
    worst.sort(reverse=True)
    print(f"--- {tag}: worst rel-L2 grad errors")
    for w in worst[:12]:
        print(f"  relL2={w[0]:.2e} max/scale={w[1]:.2e} scale={w[3]:.2e} {w[2]}")

# gen phase grads
for p in discr.parameters(): p.requires_grad_(False)
for p in rdiscr.parameters(): p.requires_grad_(False)
yh = gen(xd); lg = discr(xd, yh)
loss = F.binary_cross_entropy_with_logits(lg, torch.ones_like(lg)) + M.l1_loss(yh, yd) * 100
loss.backward()
ryh = rgen(x); rlg = rdiscr(x, ryh)
rloss = F.binary_cross_entropy_with_logits(rlg, torch.ones_like(rlg)) + F.l1_loss(ryh, y) * 100
rloss.backward()
print("gen loss", loss.item(), rloss.item(), "yhat L1", (yh.detach().cpu() - ryh.detach()).abs().mean().item())
report("generator (gen phase)", gen, rgen)
for p in discr.parameters(): p.requires_grad_(True)
for p in rdiscr.parameters(): p.requires_grad_(True)
gen.zero_grad(); rgen.zero_grad()
# discr phase grads (same y_hat on both sides: oracle's)
yfake = ryh.detach()
lh, lr_ = discr(xd, yfake.to(DEV)), discr(xd, yd)
dl = (F.binary_cross_entropy_with_logits(lr_, torch.ones_like(lr_)) + F.binary_cross_entropy_with_logits(lh, torch.zeros_like(lh))) / 2
dl.backward()
rlh, rlr = rdiscr(x, yfake), rdiscr(x, y)
rdl = (F.binary_cross_entropy_with_logits(rlr, torch.ones_like(rlr)) + F.binary_cross_entropy_with_logits(rlh, torch.zeros_like(rlh))) / 2
rdl.backward()
print("discr loss", dl.item(), rdl.item())
report("discriminator (discr phase)", discr, rdiscr)
