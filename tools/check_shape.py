"""A/B helper for conv tile shapes (MI355_CONV_SHAPE is read once per process):
   python tools/check_shape.py run out.pt      -- run a set of wide bf16 3x3x3 convs, save outputs + statistics
   python tools/check_shape.py cmp a.pt b.pt   -- compare two such files (f32 reference inside each file)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(path):
    from unet_bssfp_amd import ops
    from unet_bssfp_amd.nn import Conv3d
    dev = "cuda:0"
    res, plans = {}, []
    ops.CONV_PROBE = lambda pid, d, real: plans.append(pid)
    cases = [("a", 32, 0, 32, (128, 64, 64)), ("b", 32, 64, 32, (126, 66, 70)), ("c", 16, 0, 64, (130, 64, 96)), ("d", 32, 0, 96, (128, 60, 64))]
    for name, c0, c1, cout, (d, h, w) in cases:
        torch.manual_seed(1)
        layer = Conv3d(c0 + c1, cout, 3, 1, 1).to(dev)
        x0 = torch.randn(1, d, h, w, c0, device=dev).bfloat16()
        x1 = torch.randn(1, d, h, w, c1, device=dev).bfloat16() if c1 else None
        wp, coutp, _ = layer.spec.w_fwd(layer.weight, torch.bfloat16, c0 + c1)
        out = ops.new_act(1, d, h, w, cout, torch.bfloat16, dev)
        out.fill_(7.0)
        tiles, _ = ops.conv_num_tiles(x0, x1, wp, coutp, 3, 1, (1, 1, 1), out, (d, h, w))
        part = torch.zeros((tiles, 2, coutp), dtype=torch.float32, device=dev)
        ops.conv_fwd(x0, x1, wp, coutp, layer.bias.detach(), 3, 1, (1, 1, 1), out, (d, h, w), stats=part)
        torch.cuda.synchronize()
        x = torch.cat([x0] + ([x1] if c1 else []), -1).float().permute(0, 4, 1, 2, 3)
        ref = torch.nn.functional.conv3d(x, layer.weight.detach().bfloat16().float(), layer.bias.detach(), padding=1)
        res[name] = dict(out=out.float().cpu(), stats=part.sum(0).cpu(), ref=ref.permute(0, 2, 3, 4, 1).cpu(), cout=cout,
                         plan=plans[-1])
        print(name, "plan", res[name]["plan"], flush=True)
    torch.save(res, path)


def cmp(pa, pb):
    a, b = torch.load(pa), torch.load(pb)
    for k in a:
        ra, rb, ref, cout = a[k]["out"], b[k]["out"], a[k]["ref"], a[k]["cout"]
        ea = (ra[..., :cout] - ref).abs().max().item(); eb = (rb[..., :cout] - ref).abs().max().item()
        sa, sb = a[k]["stats"], b[k]["stats"]
        srel = ((sa - sb).abs() / (sa.abs() + 1.0)).max().item()
        print(f"{k}: max|A-ref|={ea:.4f} max|B-ref|={eb:.4f} max|A-B|={(ra - rb).abs().max().item():.4f} stats rel diff={srel:.2e} pad equal={torch.equal(ra[..., cout:], rb[..., cout:])}")
        assert eb <= max(2 * ea, 0.05) and srel < 1e-3


if sys.argv[1] == "run":
    run(sys.argv[2])
else:
    cmp(sys.argv[2], sys.argv[3])
