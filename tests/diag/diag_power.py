"""Is the marching convolution limited by its schedule or by the chip's power management?  Same launch, operands of
different bit activity (zeros / one constant / N(0,1) bf16): the instruction stream is identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unet_bssfp_amd import ops
from unet_bssfp_amd.nn import Conv3d
from tools.bench_kernels import timeit

DEV = "cuda:0"
s, c0, cout = 128, 32, 32
layer = Conv3d(c0, cout, 3, 1, 1).to(DEV)
for name, mk in (("zeros", lambda sh: torch.zeros(sh, device=DEV)), ("const 1.0", lambda sh: torch.ones(sh, device=DEV)),
                 ("randn", lambda sh: torch.randn(sh, device=DEV)), ("zeros again", lambda sh: torch.zeros(sh, device=DEV))):
    x0 = mk((1, s, s, s, c0)).to(torch.bfloat16)
    with torch.no_grad():
        layer.weight.copy_(mk(layer.weight.shape) if name != "randn" else torch.randn_like(layer.weight) * 0.05)
    wp, coutp, _ = layer.spec.w_fwd(layer.weight, torch.bfloat16, c0)
    out = ops.new_act(1, s, s, s, cout, torch.bfloat16, DEV)
    tiles, _ = ops.conv_num_tiles(x0, None, wp, coutp, 3, 1, (1, 1, 1), out, (s, s, s))
    part = torch.empty((tiles, 2, coutp), dtype=torch.float32, device=DEV)
    ms = timeit(lambda: ops.conv_fwd(x0, None, wp, coutp, layer.bias.detach(), 3, 1, (1, 1, 1), out, (s, s, s), stats=part), 100)
    print(f"{name:12s} {ms * 1e3:8.1f} us  {2.0 * c0 * cout * 27 * s ** 3 / ms / 1e9:8.1f} TFLOP/s", flush=True)
