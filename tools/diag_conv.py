"""Diagnostic: relative-L2 error of conv fwd/dgrad/wgrad/bias-grad vs CPU at larger sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F
from unet_bssfp_amd import functional as Fn
from unet_bssfp_amd.nn import Conv3d
from test_gpu_ops import to_act, from_act, DEV

def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()

cases = [("k4s2_30_32", 1, 30, 32, (32, 32, 32), 4, 2, 1), ("k4s2_32_64", 1, 32, 64, (32, 32, 32), 4, 2, 1),
         ("k4s2_32_64_small", 2, 32, 64, (8, 8, 8), 4, 2, 1), ("k3_32_32", 1, 32, 32, (16, 32, 32), 3, 1, 1),
         ("k4s2_64_128", 1, 64, 128, (16, 16, 16), 4, 2, 1)]
for name, n, cin, cout, sp, ks, st, pad in cases:
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(1)
    layer = Conv3d(cin, cout, ks, st, pad)
    x = torch.rand(n, cin, *sp, generator=g) - 0.3
    w_cpu = layer.weight.detach().clone().requires_grad_(True)
    b_cpu = layer.bias.detach().clone().requires_grad_(True)
    x_cpu = x.clone().requires_grad_(True)
    z_ref = F.conv3d(x_cpu, w_cpu, b_cpu, st, pad)
    gz = torch.rand(z_ref.shape, generator=g) - 0.5
    z_ref.backward(gz)
    layer = layer.to(DEV)
    a = to_act(x, torch.float32).requires_grad_(True)
    z, _ = Fn.ConvFn.apply(a, None, layer.weight, layer.bias, layer.spec, False)
    z.backward(to_act(gz, torch.float32))
    print(f"{name}: fwd {rel(from_act(z, cout), z_ref.detach()):.2e} dgrad {rel(from_act(a.grad, cin), x_cpu.grad):.2e} "
          f"wgrad {rel(layer.weight.grad.cpu(), w_cpu.grad):.2e} bgrad {rel(layer.bias.grad.cpu(), b_cpu.grad):.2e}")
    # double-precision reference to see which side is off
    z64 = F.conv3d(x.double(), w_cpu.detach().double(), b_cpu.detach().double(), st, pad)
    xg = x.double().requires_grad_(True)
    F.conv3d(xg, w_cpu.detach().double(), None, st, pad).backward(gz.double())
    print(f"    vs f64: fwd gpu {rel(from_act(z, cout), z64):.2e} cpu {rel(z_ref.detach(), z64):.2e} | dgrad gpu {rel(from_act(a.grad, cin), xg.grad):.2e} cpu {rel(x_cpu.grad, xg.grad):.2e}")
