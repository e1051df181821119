"""Micro-benchmarks of single kernels through the C ABI (HIP-event timing on the launch stream).
Usage: python tools/bench_kernels.py [conv|wgrad|norm|all] [--dtype bf16|f32|fp8 (conv only)] [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_bssfp_amd import functional as Fn, ops
from unet_bssfp_amd.nn import Conv3d

DEV = "cuda:0"


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bench_conv(dtype, reps, only=None):
    cases = [("32->32 @128^3", 32, 0, 32, 128), ("24(32)->32 @128^3", 32, 0, 32, 128), ("96->32 @128^3 (32|64)", 32, 64, 32, 128), ("32->96 @128^3", 32, 0, 96, 128),
             ("64->32 @64^3", 64, 0, 32, 64), ("32->64 @64^3", 32, 0, 64, 64), ("64->64 @64^3", 64, 0, 64, 64), ("128->64 @64^3 (64|64)", 64, 64, 64, 64), ("64->128 @32^3", 64, 0, 128, 32), ("128->128 @32^3", 128, 0, 128, 32), ("256->128 @32^3 (128|128)", 128, 128, 128, 32),
             ("128->64 @32^3", 128, 0, 64, 32), ("128->256 @32^3", 128, 0, 256, 32),
             ("128->256 @16^3", 128, 0, 256, 16), ("512->256 @16^3 (256|256)", 256, 256, 256, 16), ("256->128 @16^3", 256, 0, 128, 16),
             ("256->256 @16^3", 256, 0, 256, 16), ("512->512 @8^3", 512, 0, 512, 8)]
    for name, c0, c1, cout, s in cases:
        if only and only not in name:
            continue
        layer = Conv3d(c0 + c1, cout, 3, 1, 1).to(DEV)
        fp8 = dtype == "fp8"
        if fp8 and (c0 != 32 or c1):
            continue
        adt = torch.bfloat16 if fp8 else dtype
        x0 = torch.randn(1, s, s, s, c0, device=DEV).to(adt)
        x1 = torch.randn(1, s, s, s, c1, device=DEV).to(adt) if c1 else None
        q = None
        if fp8:
            wp, coutp, _, amax_w = layer.spec.w_fwd8(layer.weight, c0)
            amax_x = ops.amax_act(x0)
            x0 = ops.cast_fp8(x0, amax_x)
            q = (amax_x, amax_w)
        else:
            wp, coutp, _ = layer.spec.w_fwd(layer.weight, dtype, c0 + c1)
        out = ops.new_act(1, s, s, s, cout, adt, DEV)
        bias = layer.bias.detach()
        tiles, _ = ops.conv_num_tiles(x0, x1, wp, coutp, 3, 1, (1, 1, 1), out, (s, s, s), fp8=q)
        part = torch.empty((tiles, 2, coutp), dtype=torch.float32, device=DEV)
        plan = []
        ops.CONV_PROBE = lambda pid, d, real: plan.append(pid) and None
        ops.conv_fwd(x0, x1, wp, coutp, bias, 3, 1, (1, 1, 1), out, (s, s, s), stats=part, fp8=q)
        ops.CONV_PROBE = None
        ms = timeit(lambda: ops.conv_fwd(x0, x1, wp, coutp, bias, 3, 1, (1, 1, 1), out, (s, s, s), stats=part, fp8=q), reps)
        fl = 2.0 * (c0 + c1) * cout * 27 * s ** 3
        print(f"conv fwd  {name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s   plan {plan[0]}  stat rows {tiles}")


def bench_patchgan(dtype, reps, only=None):
    """PatchGAN k4 s2 p1 blocks as dense k2 s1 convolutions on space-to-depth tensors (forward)"""
    for name, cin, cout, s in [("d1 30->32 @128^3->64^3", 30, 32, 64), ("d2 32->64 @64^3->32^3", 32, 64, 32), ("d3 64->128 @32^3->16^3", 64, 128, 16)]:
        if only and only not in name:
            continue
        layer = Conv3d(cin, cout, 4, 2, 1).to(DEV)
        cp = ops.round_up(cin, 16)
        x0 = torch.randn(1, s + 1, s + 1, s + 1, 8 * cp, device=DEV).to(dtype)
        wp, coutp, _ = layer.spec.w_fwd_s2d(layer.weight, dtype, cp)
        out = ops.new_act(1, s, s, s, ops.round_up(cout, 16), dtype, DEV)
        bias = layer.bias.detach()
        plan = []
        ops.CONV_PROBE = lambda pid, d, real: plan.append(pid) and None
        ops.conv_fwd(x0, None, wp, coutp, bias, 2, 1, (0, 0, 0), out, (s, s, s))
        ops.CONV_PROBE = None
        ms = timeit(lambda: ops.conv_fwd(x0, None, wp, coutp, bias, 2, 1, (0, 0, 0), out, (s, s, s)), reps)
        fl = 2.0 * 8 * cp * 8 * cout * s ** 3
        nb = x0.numel() * 2 + out.numel() * 2
        print(f"conv k2/s2d {name:26s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s (dense)  {nb/ms/1e6:7.1f} GB/s  plan {plan[0]}")


def bench_upcat(dtype, reps, only=None):
    """MONAI UpCat's upsample + concat + first convolution as Fn.UpCatConvFn (no up-sampled tensor) against the unfused launches"""
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd.nn import ConvTranspose3d
    for name, ce, cl, cu, co, s in [("upcat_1 64->64 up, 96->32 @128^3", 32, 64, 64, 32, 64), ("upcat_2 128->64 up, 128->64 @64^3", 64, 128, 64, 64, 32)]:
        if only and only not in name:
            continue
        deconv, conv = ConvTranspose3d(cl, cu).to(DEV), Conv3d(ce + cu, co, 3, 1, 1).to(DEV)
        x_e = torch.randn(1, 2 * s, 2 * s, 2 * s, ce, device=DEV).to(dtype).requires_grad_(True)
        x_l = torch.randn(1, s, s, s, cl, device=DEV).to(dtype).requires_grad_(True)
        tables = Fn.UpCatTables()
        fused = lambda: Fn.UpCatConvFn.apply(x_e, x_l, deconv.weight, deconv.bias, conv.weight, conv.bias, conv.spec, tables, True)[0]

        def unfused():
            up = Fn.ConvFn.apply(x_l, None, deconv.weight, deconv.bias, deconv.spec, False)[0]
            return Fn.ConvFn.apply(x_e, up, conv.weight, conv.bias, conv.spec, True, True)[0]
        for label, fn in (("fused", fused), ("unfused", unfused)):
            with torch.no_grad():
                ms_f = timeit(fn, reps)
            z = fn()
            g = torch.randn_like(z)
            params = [deconv.weight, deconv.bias, conv.weight]
            ms_b = timeit(lambda: torch.autograd.grad(z, [x_e, x_l] + params, g, retain_graph=True), reps)
            print(f"upcat {label:8s} {name:36s} fwd {ms_f*1e3:8.1f} us   bwd {ms_b*1e3:8.1f} us")
        ms = timeit(lambda: ops.upcat_compose(deconv.weight.detach(), conv.weight.detach(), deconv.bias.detach(), conv.bias.detach(), ce, out=tables.bufs), reps)
        print(f"upcat compose  {name:36s}     {ms*1e3:8.1f} us")


def bench_deconv(dtype, reps, only=None):
    """transposed conv k2 s2 (forward = one 1x1x1 GEMM with 8*Cout columns, data gradient = k2 s2 gather)"""
    from unet_bssfp_amd.nn import ConvTranspose3d
    for name, cin, cout, s in [("64->64 @64^3->128^3", 64, 64, 64), ("128->64 @32^3->64^3", 128, 64, 32), ("512->256 @8^3->16^3", 512, 256, 8),
                               ("64->64 @32^3->64^3 (upcat_2)", 64, 64, 32), ("128->128 @16^3->32^3 (upcat_3)", 128, 128, 16),
                               ("256->256 @8^3->16^3 (upcat_4)", 256, 256, 8)]:
        if only and only not in name:
            continue
        layer = ConvTranspose3d(cin, cout).to(DEV)
        from unet_bssfp_amd import nn as N
        N.set_compute_dtype(layer, dtype)
        x = torch.randn(1, s, s, s, cin, device=DEV).to(dtype).requires_grad_(True)
        with torch.no_grad():
            ms = timeit(lambda: layer.forward_act(x), reps)
        es = 2 if dtype == torch.bfloat16 else 4
        ob = 8 * s ** 3 * cout * es
        print(f"deconv fwd {name:26s} {ms*1e3:9.1f} us  {ob/ms/1e6:8.1f} GB/s of output")
        y = layer.forward_act(x)
        g = torch.randn_like(y)
        layer.weight.requires_grad_(False); layer.bias.requires_grad_(False)      # data gradient only
        y = layer.forward_act(x)
        ms = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True), reps)
        print(f"deconv dgrad {name:24s} {ms*1e3:9.1f} us  {ob/ms/1e6:8.1f} GB/s of dy")


def bench_wgrad(dtype, reps, only=None):
    cases = [("32->32 @128^3", 32, 32, 128), ("96->32 @128^3", 96, 32, 128), ("64->64 @64^3", 64, 64, 64),
             ("128->128 @32^3", 128, 128, 32), ("256->256 @16^3", 256, 256, 16), ("512->512 @8^3", 512, 512, 8)]
    for name, cin, cout, s in cases:
        if only and only not in name:
            continue
        x = torch.randn(1, s, s, s, cin, device=DEV).to(dtype)
        g = torch.randn(1, s, s, s, cout, device=DEV).to(dtype)
        dw = torch.empty(cout, cin, 3, 3, 3, device=DEV)
        ms = timeit(lambda: ops.conv_wgrad(x, None, g, (s, s, s), 1, (0, 0, 0), 3, 1, (1, 1, 1), dw, cout, cin,
                                           cin * 27, 27, (9, 3, 1), (0, 0, 0), (1, 1, 1)), reps)
        fl = 2.0 * cin * cout * 27 * s ** 3
        print(f"wgrad     {name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s")
    # the layers outside the 3x3x3 families: 1x1x1 head / final at full resolution (HBM streams), the PatchGAN's first blocks
    # as dense k2 on space-to-depth operands (first block: its x- and y-parts, Fn.SplitS2dConvFn)
    for name, cin, cout, s in [("k1 head 24->24 @128^3", 24, 24, 128), ("k1 final 32->6 @128^3", 32, 6, 128)]:
        if only and only not in name:
            continue
        x = torch.randn(1, s, s, s, ops.round_up(cin, 16), device=DEV).to(dtype)
        g = torch.randn(1, s, s, s, ops.round_up(cout, 16), device=DEV).to(dtype)
        dw = torch.empty(cout, cin, 1, 1, 1, device=DEV)
        kinds = []
        ops.WGRAD_PROBE = lambda k, d: kinds.append(k)
        ms = timeit(lambda: ops.conv_wgrad(x, None, g, (s, s, s), 1, (0, 0, 0), 1, 1, (0, 0, 0), dw, cout, cin, cin, 1, (1, 1, 1), (0, 0, 0), (1, 1, 1)), reps)
        ops.WGRAD_PROBE = None
        nb = (x.numel() + g.numel()) * x.element_size()
        print(f"wgrad     {name:28s} {ms*1e3:9.1f} us  {nb/ms/1e6:8.1f} GB/s of operands   kind {kinds[0]}")
    for name, n, cin, cp, cout, s in [("k2/s2d d1 x-part 24->32 @64^3", 1, 24, 32, 32, 64), ("k2/s2d d1 y-part 6->32 N=2", 2, 6, 8, 32, 64),
                                      ("k2/s2d d2 32->64 @32^3 N=2", 2, 32, 32, 64, 32)]:
        if only and only not in name:
            continue
        x = torch.randn(n, s + 1, s + 1, s + 1, 8 * cp, device=DEV).to(dtype)
        g = torch.randn(n, s, s, s, ops.round_up(cout, 16), device=DEV).to(dtype)
        dw = torch.empty(cout, cin, 4, 4, 4, device=DEV)
        ms = timeit(lambda: ops.conv_wgrad(x, None, g, (s, s, s), 1, (0, 0, 0), 2, 1, (0, 0, 0), dw, cout, cin, cin * 64, 64, (16, 4, 1),
                                           (0, 0, 0), (2, 2, 2), s2d_cp=cp), reps)
        fl = 2.0 * 8 * cp * 8 * cout * n * s ** 3
        nb = (x.numel() + g.numel()) * x.element_size()
        print(f"wgrad     {name:30s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s (dense)  {nb/ms/1e6:8.1f} GB/s of operands")


def bench_norm(dtype, reps, only=None):
    es = 2 if dtype == torch.bfloat16 else 4
    for name, c, s in [("C=32 @128^3", 32, 128), ("C=64 @64^3", 64, 64), ("C=128 @32^3", 128, 32)]:
        if only and only not in name:
            continue
        z = torch.randn(1, s, s, s, c, device=DEV).to(dtype)
        da = torch.randn(1, s, s, s, c, device=DEV).to(dtype)
        gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        part, bpg = ops.channel_stats(z, 1)
        mean, rstd = ops.norm_finalize(part, bpg, 1, c, s ** 3, None, 1e-5)
        out = torch.empty_like(z)
        nb = z.numel() * es
        ms = timeit(lambda: ops.normact_fwd(z, 1, mean, rstd, gamma, beta, 0.1, 0.05, 1234, out=out), reps)
        print(f"normact fwd (p=.05) {name:16s} {ms*1e3:9.1f} us  {2*nb/ms/1e6:8.1f} GB/s (2 passes)")
        ms = timeit(lambda: ops.normact_fwd(z, 1, mean, rstd, gamma, beta, 0.1, 0.0, 0, out=out), reps)
        print(f"normact fwd (p=0)   {name:16s} {ms*1e3:9.1f} us  {2*nb/ms/1e6:8.1f} GB/s (2 passes)")
        if c == 32 and dtype == torch.bfloat16:
            tab = torch.tensor([[3.0, 0.0]], device=DEV)
            a8 = torch.empty(z.shape, dtype=torch.uint8, device=DEV)
            for p_, sd in ((0.05, 1234), (0.0, 0)):
                ms = timeit(lambda: ops.normact_fwd(z, 1, mean, rstd, gamma, beta, 0.1, p_, sd, out=out, q8=(a8, tab[0, 0:1], tab[0, 1:2])), reps)
                print(f"normact fwd (p={p_}) + e4m3 copy {name:8s} {ms*1e3:9.1f} us  {2.5*nb/ms/1e6:8.1f} GB/s (2.5 passes)")
            ms = timeit(lambda: ops.normact_bwd(z, da, 1, mean, rstd, gamma, beta, 0.1, 0.05, 1234, True, True, q8=(a8, tab[0, 0:1], tab[0, 1:2])), reps)
            print(f"normact bwd (all) + e4m3 copy {name:10s} {ms*1e3:9.1f} us  {5.5*nb/ms/1e6:8.1f} GB/s (5.5 passes)")
            ms = timeit(lambda: ops.cast_fp8(z, tab[0, 0:1], tab[0, 1:2]), reps)
            print(f"cast_fp8 (one pass, gathers) {name:11s} {ms*1e3:9.1f} us  {1.5*nb/ms/1e6:8.1f} GB/s (1.5 passes)")
        ms = timeit(lambda: ops.channel_stats(z, 1), reps)
        print(f"channel_stats       {name:16s} {ms*1e3:9.1f} us  {nb/ms/1e6:8.1f} GB/s (1 pass)")
        ms = timeit(lambda: ops.normact_bwd(z, da, 1, mean, rstd, gamma, beta, 0.1, 0.05, 1234, True, True), reps)
        print(f"normact bwd (all)   {name:16s} {ms*1e3:9.1f} us  {5*nb/ms/1e6:8.1f} GB/s (5 passes)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--lib", default=None, help="another build of the C ABI (tools/build_diag.sh)")
    a = ap.parse_args()
    if a.lib:
        import tools.diaglib as D
        D.use(a.lib)
    dt = torch.bfloat16 if a.dtype == "bf16" else ("fp8" if a.dtype == "fp8" else torch.float32)
    torch.manual_seed(0)
    if a.what in ("conv", "all"):
        bench_conv(dt, a.reps, a.only)
    if a.what in ("patchgan", "all"):
        bench_patchgan(dt, a.reps, a.only)
    if a.what in ("upcat", "all"):
        bench_upcat(dt, a.reps, a.only)
    if a.what in ("deconv", "all"):
        bench_deconv(dt, a.reps, a.only)
    if a.what in ("wgrad", "all"):
        bench_wgrad(dt, a.reps, a.only)
    if a.what in ("norm", "all"):
        bench_norm(dt, a.reps, a.only)
