"""Where do two graph instances diverge from eager steps?  python tests/diag/diag_two_instances.py [mode]
The eager model runs FIRST (snapshots per step): DropoutState.reset() in build() frees the previous model's counter tensor,
which a captured graph would keep using."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import unet_bssfp_amd as M
from unet_bssfp_amd.functional import DropoutState
from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch
DEV = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "default"

def build():
    torch.manual_seed(4)
    DropoutState.reset()
    gen, discr = M.Generator("bssfp", dropout=0.05), M.Discriminator("bssfp")
    return bSSFPToDWITensorModel("bssfp", gen=gen.to(DEV), discr=discr.to(DEV)).train()

a, b = synthetic_batch(2, 32, seed=9, device=DEV), synthetic_batch(2, 32, seed=(9 if mode == "same_batch" else 10), device=DEV)
seq = [(0, a)] * 4 if mode in ("one_instance", "replay_only_0") else [(0, a), (1, b), (0, a), (1, b)]
eager = build()
eager.training_step(a, 0); eager.training_step(a, 1)
snaps = []
for i, (k, batch) in enumerate(seq):
    eager.training_step(batch, 2 + i)
    torch.cuda.synchronize()
    snaps.append([p.detach().clone() for p in eager.parameters()])
graphed = build()
gs = GraphedTrainingStep(graphed, a, warmup=2)
if mode == "one_instance":
    pass
elif mode == "separate_pools":
    torch.cuda.synchronize()
    gs.instances.append((b, gs._capture(b, None)))
else:
    gs.add_instance(b)
for i, (k, batch) in enumerate(seq):
    gs(k)
    torch.cuda.synchronize()
    bad = [(n, float((p - q).abs().max())) for (n, q), p in zip(graphed.named_parameters(), snaps[i]) if not torch.equal(p, q)]
    print(f"{mode}: after replay {i} (instance {k}): {len(bad)} parameters differ", bad[:3])
