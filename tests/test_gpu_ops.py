"""GPU parity of every HIP kernel (through the C ABI) against plain PyTorch CPU f32 ops.

Tolerances, derived:
* f32 path: the f32 MFMA is an exact fmaf chain, so differences come only from the summation order:
  rtol 2e-4 / atol 2e-5 on O(1) data.
* bf16 path: the test feeds bf16-representable inputs and weights, products of two bf16 values are exact in f32 and
  the kernels accumulate in f32, so the ONLY legitimate error of a bf16 OUTPUT is its final rounding (half an ulp =
  2^-9 = 2e-3 relative): `close()` allows rtol 1e-2 plus atol 2e-3 * sigma(reference) for values near zero.
* weight / bias gradients are f32 OUTPUTS in both modes: sums over all output positions whose only error is the f32
  summation order, bounded relative to the largest element: |err| <= 2e-3 * max|ref| (measured ~1e-5 .. 1e-4); an
  all-zero or mis-scaled gradient fails by a factor of 500.
"""
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: dict(rtol=2e-4, atol=2e-5)}


def close(got, ref, dtype, what=""):
    """Output stored in `dtype`: f32 -> summation-order noise only; bf16 -> final rounding (see the module docstring)."""
    if dtype == torch.float32:
        torch.testing.assert_close(got, ref, **TOL[dtype], msg=lambda m: f"{what}: {m}")
    else:
        sigma = float(ref.float().std()) if ref.numel() > 1 else float(ref.abs().max())
        torch.testing.assert_close(got, ref, rtol=1e-2, atol=2e-3 * max(sigma, 1e-30), msg=lambda m: f"{what}: {m}")


def close_f32_sum(got, ref, what=""):
    """f32 reductions over many positions (weight / bias / affine gradients): error relative to the largest element."""
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= 2e-3 * scale + 1e-12, f"{what}: max |err| {err:.3e} > 2e-3 * max|ref| ({scale:.3e})"


def _ops():
    from unet_bssfp_amd import ops
    return ops


def to_act(x, dtype, cp=None):
    """NCDHW f32 CPU tensor -> NDHWC activation on the GPU (channels padded to cp)."""
    ops = _ops()
    n, c, d, h, w = x.shape
    cp = ops.round_up(c, 16) if cp is None else cp
    out = ops.new_act(n, d, h, w, cp, dtype, DEV)
    ops.pack_ncdhw(x.to(DEV).float().contiguous(), out, 0, cp)
    return out


def from_act(a, c):
    return _ops().unpack_ncdhw(a, c, 0).cpu()


def q(x, dtype):
    """Round a CPU f32 tensor through `dtype` (so that the CPU reference sees the same inputs)."""
    return x.to(dtype).to(torch.float32)


def test_mfma_fragment_layout(hip):
    a, b = _ops().mfma_selftest(DEV)
    i = torch.arange(1, 33, dtype=torch.float32).view(32, 1)
    j = torch.arange(1, 33, dtype=torch.float32).view(1, 32)
    assert torch.equal(a.cpu(), i * 100.0 * j)           # D[i][j] = (i+1)*100*(j+1): asymmetric
    assert torch.equal(b.cpu(), i * j + 0.5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_unpack_roundtrip(hip, dtype):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 24, 4, 6, 8, generator=g)
    y = torch.rand(2, 6, 4, 6, 8, generator=g)
    ops = _ops()
    buf = ops.new_act(2, 4, 6, 8, 32, dtype, DEV)
    buf.fill_(float("nan"))
    ops.pack_ncdhw(x.to(DEV), buf, 0, 24)
    ops.pack_ncdhw(y.to(DEV), buf, 24, 32)
    full = buf.float().cpu()                              # (N,D,H,W,32)
    ref = torch.cat([q(x, dtype), q(y, dtype), torch.zeros(2, 2, 4, 6, 8)], 1).permute(0, 2, 3, 4, 1)
    assert torch.equal(full, ref)
    assert torch.equal(ops.unpack_ncdhw(buf, 6, 24).cpu(), q(y, dtype))


CONV_CASES = [
    # name, N, Cin(s), Cout, spatial, ks, stride, pad
    ("k3_wide", 1, (32,), 32, (4, 8, 32), 3, 1, 1),
    ("k3_wide_ragged", 2, (16,), 32, (3, 5, 40), 3, 1, 1),
    ("k3_mid_ct2", 1, (32,), 64, (4, 16, 16), 3, 1, 1),
    ("k3_small", 2, (64,), 64, (8, 8, 8), 3, 1, 1),
    ("k3_tiny", 1, (32,), 96, (2, 2, 2), 3, 1, 1),
    ("k3_concat", 1, (32, 64), 32, (4, 4, 32), 3, 1, 1),
    ("k3_cin24", 1, (24,), 32, (4, 4, 32), 3, 1, 1),
    ("k3_splitk", 2, (256,), 64, (4, 8, 8), 3, 1, 1),            # few positions, long contraction: split-K path
    ("k3_splitk_concat", 1, (128, 128), 96, (8, 8, 16), 3, 1, 1),
    # wide layers that take the row-reuse / LDS-DMA kernel in bf16 (>= 1024 tiles of 4x4x32, or >= 512 with 64 channels)
    ("k3_ru_ct1", 1, (32,), 32, (64, 64, 128), 3, 1, 1),
    ("k3_ru_ragged_concat", 1, (32, 64), 32, (62, 66, 130), 3, 1, 1),
    ("k3_ru_ct2_n2", 2, (16,), 64, (32, 64, 64), 3, 1, 1),
    ("k3_ru_cout96", 1, (32,), 96, (64, 64, 128), 3, 1, 1),
    # <= 32 input channels in one source, plain output grid, >= 128 footprint-segments: the marching kernel in bf16
    ("k3_march_ragged", 1, (32,), 32, (30, 100, 140), 3, 1, 1),     # border footprints (generic march), 5-plane segments
    ("k3_march_n2_short", 2, (32,), 32, (16, 64, 128), 3, 1, 1),    # two samples; last segment is a single plane
    ("k3_march_cin24_cout64", 1, (24,), 64, (32, 64, 128), 3, 1, 1),  # padded input channels, two output-channel blocks
    ("k3_march_long", 1, (32,), 32, (40, 128, 256), 3, 1, 1),       # 11-plane segments: three triples of straight-line steps
    # more than 32 input channels in whole 32-channel groups, plain output grid, enough footprints x d-segments: the
    # group-marching kernel (input planes per 32-channel group, weights streamed through an LDS ring) in bf16
    ("k3_mg_96to32", 1, (32, 64), 32, (40, 64, 128), 3, 1, 1),       # upcat_1.conv_0's shape: skip + up-sampled source
    ("k3_mg_ragged", 1, (32, 64), 32, (21, 70, 100), 3, 1, 1),       # border footprints in h and w, odd depth
    ("k3_mg_64to64", 1, (64,), 64, (32, 64, 64), 3, 1, 1),           # 64^3-level shape: two output-channel blocks
    ("k3_mg_n2_cout96", 2, (64,), 96, (16, 32, 64), 3, 1, 1),        # two samples; 32 -> 96-style data gradient widths
    ("k3_mg_128to64", 1, (64, 64), 64, (16, 64, 64), 3, 1, 1),       # upcat_2.conv_0: four groups from two sources
    ("k3_mg_rows2", 1, (64,), 64, (32, 32, 32), 3, 1, 1),            # too few 16-row footprints: the 8-row variant
    ("k3_mg_256to128", 1, (128, 128), 128, (32, 32, 32), 3, 1, 1),   # upcat_3.conv_0: eight groups, few workgroups, no split-K
    # low levels (extents <= 16, >= 64 channels either side, 64-channel output blocks) in bf16: conv_lowg_kernel -- 512-voxel
    # tiles, weights through LDS once per workgroup, split-K down to one 16-channel chunk per workgroup
    ("k3_lowg_16", 1, (128, 128), 128, (16, 16, 16), 3, 1, 1),       # upcat_4.conv_0's shape at 16^3: 4x8x16 tiles, two sources
    ("k3_lowg_8", 1, (256,), 128, (8, 8, 8), 3, 1, 1),               # 8^3: one 8x8x8 tile
    ("k3_lowg_ragged", 2, (64, 64), 64, (10, 10, 10), 3, 1, 1),      # 160^3 configuration's 10^3 level: partial tiles, two samples
    ("k3_lowg_w12", 1, (64,), 192, (5, 9, 12), 3, 1, 1),             # ragged 4x8x16 tiles, three output-channel blocks of 64
    ("k3_lowg_20", 1, (128, 128), 128, (20, 20, 20), 3, 1, 1),       # 160^3 configuration's 20^3 level (upcat_4.conv_0): ragged 4x8x16 tiles (rows of 20)
    ("k4s2", 1, (30,), 32, (8, 8, 16), 4, 2, 1),
    ("k4s2_ct2", 2, (32,), 64, (8, 8, 8), 4, 2, 1),
    ("k1_head", 2, (24,), 24, (4, 4, 8), 1, 1, 0),
    ("k1_final6", 1, (32,), 6, (4, 4, 8), 1, 1, 0),
    ("k1_final1", 2, (512,), 1, (2, 2, 2), 1, 1, 0),
    # >= 131072 voxels, <= 32 channels either side: the persistent pointwise kernel in bf16
    ("k1_pointwise_head", 1, (24,), 24, (64, 64, 32), 1, 1, 0),
    ("k1_pointwise_final", 2, (32,), 6, (32, 64, 32), 1, 1, 0),
    ("k1_pointwise_ragged", 1, (16,), 32, (64, 64, 33), 1, 1, 0),
    ("k1_pointwise_wgstats", 1, (24,), 24, (95, 96, 65), 1, 1, 0),   # > 2048 tiles, one sample: one statistics row per workgroup
]


@pytest.mark.parametrize("case", ["march", "marchg", "halo"])
def test_conv_fused_statistics_with_a_bias_far_from_zero(hip, case):
    """A pre-norm conv bias that has random-walked far from zero (a loaded checkpoint; default init keeps |b| <= 0.04):
    bias = 10 sigma(z - b).  The fused statistics must still be those of (z - b) to f32 summation accuracy -- a kernel that
    accumulates sum z, sum z^2 and subtracts the bias terms afterwards loses ~2 decimal digits of the variance here
    (cancellation of cnt * b^2 against sum z^2); every conv kernel accumulates (z - b) itself."""
    from unet_bssfp_amd import functional as Fn
    cins, sp, want = {"march": ((32,), (40, 64, 128), 32041), "marchg": ((32, 64), (16, 64, 128), 32141),
                      "halo": ((64,), (8, 16, 16), None)}[case]
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    layer = _conv_layer(cins, 32, 3, 1, 1, 3)
    xs = [q(torch.rand(1, c, *sp, generator=g) - 0.5, dtype) for c in cins]
    with torch.no_grad():
        layer.weight.copy_(q(layer.weight, dtype))
        zc = F.conv3d(torch.cat(xs, 1), layer.weight, None, 1, 1)
        sigma = zc.std((0, 2, 3, 4))
        layer.bias.copy_(10.0 * sigma * torch.where(torch.arange(32) % 2 == 0, 1.0, -1.0))
    layer = layer.to(DEV)
    acts = [to_act(x, dtype) for x in xs]
    plans = []
    _ops().CONV_PROBE = lambda pid, d, real: plans.append(pid)
    try:
        z, part = Fn.ConvFn.apply(acts[0], acts[1] if len(acts) > 1 else None, layer.weight, layer.bias, layer.spec, True)
    finally:
        _ops().CONV_PROBE = None
    if want is not None:
        assert plans == [want], plans
    s = part.sum(0).double().cpu()
    npos = zc.numel() / 32
    ref1, ref2 = zc.double().sum((0, 2, 3, 4)), (zc.double() ** 2).sum((0, 2, 3, 4))
    var_got = s[1, :32] / npos - (s[0, :32] / npos) ** 2
    var_ref = ref2 / npos - (ref1 / npos) ** 2
    assert float(((var_got - var_ref).abs() / var_ref).max()) <= 1e-4, ((var_got - var_ref).abs() / var_ref).max()
    assert float(((s[1, :32] - ref2).abs() / ref2).max()) <= 1e-4
    close(from_act(z, 32), zc + layer.bias.detach().cpu().view(1, -1, 1, 1, 1), dtype, "z")


# which kernel a bf16 case must take (plan id = 10000 ks + 1000 halo + 100 shape + 10 vt + ct; mi355_conv_plan_id):
# 32041 conv_march_kernel, 32141 / 32121 conv_marchg_kernel<4> / <2>, 31941 / 31942 conv_ru_kernel<1> / <2>
BF16_PLANS = {
    "k3_march_ragged": 32041, "k3_march_n2_short": 32041, "k3_march_cin24_cout64": 32041, "k3_march_long": 32041,
    "k3_ru_ct1": 32041, "k3_ru_cout96": 32041, "k3_ru_ragged_concat": 32141, "k3_ru_ct2_n2": 31942,
    "k3_mg_96to32": 32141, "k3_mg_ragged": 32141, "k3_mg_64to64": 32141, "k3_mg_n2_cout96": 32141, "k3_mg_128to64": 32141,
    "k3_mg_rows2": 32121, "k3_mg_256to128": 32141,
    "k3_small": 32342, "k3_splitk": 32342, "k3_lowg_16": 32242, "k3_lowg_8": 32342, "k3_lowg_ragged": 32242, "k3_lowg_w12": 32242, "k3_lowg_20": 32242,
}


def _conv_layer(cins, cout, ks, stride, pad, seed):
    from unet_bssfp_amd.nn import Conv3d
    torch.manual_seed(seed)
    return Conv3d(sum(cins), cout, ks, stride, pad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(hip, case, dtype):
    from unet_bssfp_amd import functional as Fn
    name, n, cins, cout, sp, ks, stride, pad = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)     # (str hash is salted per process)
    layer = _conv_layer(cins, cout, ks, stride, pad, 1)
    with torch.no_grad():
        layer.weight.copy_(q(layer.weight, dtype))       # weights representable in the compute dtype
    xs = [q(torch.rand(n, c, *sp, generator=g) - 0.3, dtype) for c in cins]
    w_cpu = layer.weight.detach().clone().requires_grad_(True)
    b_cpu = layer.bias.detach().clone().requires_grad_(True)
    xcat = torch.cat(xs, 1).requires_grad_(True)
    z_ref = F.conv3d(xcat, w_cpu, b_cpu, stride, pad)
    gz = q(torch.rand(z_ref.shape, generator=g) - 0.5, dtype)
    z_ref.backward(gz)

    layer = layer.to(DEV)
    acts = [to_act(x, dtype).requires_grad_(True) for x in xs]
    plans = []
    _ops().CONV_PROBE = lambda pid, d, real: plans.append(pid)
    try:
        z, part = Fn.ConvFn.apply(acts[0], acts[1] if len(acts) > 1 else None, layer.weight, layer.bias, layer.spec, True)
    finally:
        _ops().CONV_PROBE = None
    if dtype == torch.bfloat16 and name in BF16_PLANS:
        assert plans and plans[0] == BF16_PLANS[name], (plans, BF16_PLANS[name])
    close(from_act(z, cout), z_ref.detach(), dtype, "z")
    cp = z.shape[4]
    if cp > cout:                                        # pad channels must hold zeros
        assert float(z[..., cout:].abs().max()) == 0.0
    # fused statistics of (z - bias): per-tile partial sums
    s = part.sum(0).cpu()                                # [2][coutp]
    zc = (z_ref.detach() - b_cpu.detach().view(1, -1, 1, 1, 1))
    # (computed from the f32 accumulators, before the output is rounded: f32 summation-order noise only).  The signed sum
    # cancels, so it is bounded on the scale of sum |z - b| <= sqrt(N * sum (z - b)^2)
    n_pos = zc.numel() / cout
    e0 = (s[0, :cout] - zc.sum((0, 2, 3, 4))).abs()
    assert bool((e0 <= 2e-3 * (n_pos * (zc * zc).sum((0, 2, 3, 4))).sqrt() + 1e-6).all()), e0.max()
    close_f32_sum(s[1, :cout], (zc * zc).sum((0, 2, 3, 4)), "sum (z-b)^2")
    # backward
    kinds = []
    _ops().WGRAD_PROBE = lambda kind, d: kinds.append(kind)
    try:
        z.backward(to_act(gz, dtype))
    finally:
        _ops().WGRAD_PROBE = None
    if (name.startswith("k3_ru") or name.startswith("k3_march")) and dtype == torch.bfloat16:
        assert kinds == [2], kinds                                       # wgrad_march_kernel (3x3x3, W >= 32, D >= 8)
    if name.startswith("k1_pointwise") and dtype == torch.bfloat16:
        assert kinds == [3], kinds                                       # wgrad_pw_kernel (1x1x1, <= 32 channels either side, >= 128 K voxels)
    off = 0
    for a, c in zip(acts, cins):
        close(from_act(a.grad, c), xcat.grad[:, off:off + c], dtype, "dx")
        off += c
    close_f32_sum(layer.weight.grad.cpu(), w_cpu.grad, "dw")
    close_f32_sum(layer.bias.grad.cpu(), b_cpu.grad, "db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,sp", [(64, 64, (2, 4, 8)), (128, 64, (4, 4, 4)),
                                         (64, 64, (64, 64, 33)), (128, 64, (32, 64, 65)),    # >= 131072 voxels: deconv_fwd_kernel in bf16
                                         (64, 64, (8, 8, 32)), (128, 64, (4, 8, 64)),        # W % 32 == 0: wgrad_deconv_kernel in bf16
                                         (256, 128, (8, 8, 8))])       # upcat_4's shape class: the data gradient's (tap, chunk) pairs split over blockIdx.z
def test_deconv_fwd_bwd(hip, dtype, cin, cout, sp):
    from unet_bssfp_amd.nn import ConvTranspose3d
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(2)
    layer = ConvTranspose3d(cin, cout, 2, 2)
    with torch.no_grad():
        layer.weight.copy_(q(layer.weight, dtype))
    x = q(torch.rand(2, cin, *sp, generator=g) - 0.4, dtype)
    w_cpu = layer.weight.detach().clone().requires_grad_(True)
    b_cpu = layer.bias.detach().clone().requires_grad_(True)
    x_cpu = x.clone().requires_grad_(True)
    z_ref = F.conv_transpose3d(x_cpu, w_cpu, b_cpu, stride=2)
    gz = q(torch.rand(z_ref.shape, generator=g) - 0.5, dtype)
    z_ref.backward(gz)
    layer = layer.to(DEV)
    a = to_act(x, dtype).requires_grad_(True)
    z = layer.forward_act(a)
    close(from_act(z, cout), z_ref.detach(), dtype, "z")
    z.backward(to_act(gz, dtype))
    close(from_act(a.grad, cin), x_cpu.grad, dtype, "dx")
    close_f32_sum(layer.weight.grad.cpu(), w_cpu.grad, "dw")
    close_f32_sum(layer.bias.grad.cpu(), b_cpu.grad, "db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,n,c,sp", [("instance", 2, 32, (4, 8, 8)), ("instance", 1, 64, (16, 16, 32)),
                                         ("batch", 2, 24, (4, 4, 8)), ("batch", 3, 512, (2, 2, 2)),
                                         ("none", 2, 32, (4, 4, 4))])
def test_normact_fwd_bwd(hip, dtype, kind, n, c, sp):
    from unet_bssfp_amd import functional as Fn
    g = torch.Generator().manual_seed(7)
    slope = 0.1 if kind == "instance" else 0.2
    z = q(torch.randn(n, c, *sp, generator=g) * 1.5 + 0.7, dtype)
    gamma = (torch.rand(c, generator=g) + 0.5)
    beta = (torch.rand(c, generator=g) - 0.5)
    z_cpu = z.clone().requires_grad_(True)
    g_cpu, b_cpu = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    if kind == "instance":
        y = F.instance_norm(z_cpu, None, None, g_cpu, b_cpu, True, 0.0, 1e-5)
    elif kind == "batch":
        y = F.batch_norm(z_cpu, rm, rv, g_cpu, b_cpu, True, 0.1, 1e-5)
    else:
        y = z_cpu
    a_ref = F.leaky_relu(y, slope)
    ga = q(torch.rand(a_ref.shape, generator=g) - 0.5, dtype)
    a_ref.backward(ga)

    cfg = Fn.NormCfg(kind, c, slope=slope)
    zd = to_act(z, dtype).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    if kind == "none":
        a = Fn.NormActFn.apply(zd, None, None, None, None, cfg, True, None, None)
    else:
        a = Fn.NormActFn.apply(zd, None, gd, bd, None, cfg, True, rmd if kind == "batch" else None,
                               rvd if kind == "batch" else None)
    close(from_act(a, c), a_ref.detach(), dtype, "a")
    a.backward(to_act(ga, dtype))
    if dtype == torch.float32:
        torch.testing.assert_close(from_act(zd.grad, c), z_cpu.grad, rtol=1e-3, atol=1e-5)
    else:
        close(from_act(zd.grad, c), z_cpu.grad, dtype, "dz")
    if kind != "none":
        close_f32_sum(gd.grad.cpu(), g_cpu.grad, "dgamma")     # f32 outputs in both modes
        close_f32_sum(bd.grad.cpu(), b_cpu.grad, "dbeta")
    if kind == "batch":
        torch.testing.assert_close(rmd.cpu(), rm, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rvd.cpu(), rv, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,n,c,sp,s2d", [("instance", 2, 128, (4, 6, 8), False), ("batch", 1, 256, (8, 8, 8), False),
                                             ("batch", 2, 64, (4, 4, 8), True), ("instance", 1, 512, (2, 2, 2), False),
                                             # register-resident forms: 8 row slots per thread (full / predicated, two groups one
                                             # after the other), an odd group count; the streaming forms behind them: more than
                                             # 4096 rows per group, more than 16 groups
                                             ("instance", 1, 128, (16, 16, 16), False), ("instance", 2, 64, (8, 12, 16), False),
                                             ("instance", 3, 64, (4, 4, 4), False), ("batch", 1, 64, (16, 16, 32), False),
                                             ("instance", 32, 64, (2, 2, 2), False)])
def test_normact_small_tensor_kernels(hip, dtype, kind, n, c, sp, s2d):
    """mi355_normact_small_fwd / _bwd (one launch each way for tensors of up to 1 M elements: the workgroup computes the
    statistics itself) against torch, and against the three-launch path on the same input: activations, input gradient,
    affine gradients, BatchNorm running statistics and counter; space-to-depth output / gradient layout; dropout masks of
    the two paths are the same function of (seed, element index)."""
    from unet_bssfp_amd import functional as Fn, ops
    assert ops.norm_is_small(n, *sp, c)
    g = torch.Generator().manual_seed(11)
    slope = 0.1 if kind == "instance" else 0.2
    z = q(torch.randn(n, c, *sp, generator=g) * 1.5 + 0.7, dtype)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    z_cpu = z.clone().requires_grad_(True)
    g_cpu, b_cpu = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    y = F.instance_norm(z_cpu, None, None, g_cpu, b_cpu, True, 0.0, 1e-5) if kind == "instance" else \
        F.batch_norm(z_cpu, rm, rv, g_cpu, b_cpu, True, 0.1, 1e-5)
    a_ref = F.leaky_relu(y, slope)
    ga = q(torch.rand(a_ref.shape, generator=g) - 0.5, dtype)
    a_ref.backward(ga)
    outs = {}
    for small in (True, False):
        cfg = Fn.NormCfg(kind, c, slope=slope)
        zd = to_act(z, dtype).requires_grad_(True)
        gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
        rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        bn = kind == "batch"
        a = Fn.NormActFn.apply(zd, None, gd, bd, None, cfg, True, rmd if bn else None, rvd if bn else None, s2d,
                               nbt if bn else None, small)
        if s2d:                                              # S(a) back to the plain layout; the gradient arrives in S layout
            a_plain = ops.unpack_ncdhw_s2d(a.detach(), c, sp, c, 0).cpu()
            gs = Fn._new_s2d(ops.s2d_shape(n, *sp, c), dtype, DEV)
            ops.pack_ncdhw_s2d(ga.to(DEV).contiguous(), gs, c, 0, c)
            a.backward(gs)
        else:
            a_plain = from_act(a.detach(), c)
            a.backward(to_act(ga, dtype))
        outs[small] = (a_plain, from_act(zd.grad, c), gd.grad.cpu(), bd.grad.cpu(), rmd.cpu(), rvd.cpu(), int(nbt))
    sm, big = outs[True], outs[False]
    close(sm[0], a_ref.detach(), dtype, "a")
    if dtype == torch.float32:
        torch.testing.assert_close(sm[1], z_cpu.grad, rtol=1e-3, atol=1e-5)
    else:
        close(sm[1], z_cpu.grad, dtype, "dz")
    close_f32_sum(sm[2], g_cpu.grad, "dgamma")
    close_f32_sum(sm[3], b_cpu.grad, "dbeta")
    if kind == "batch":
        torch.testing.assert_close(sm[4], rm, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sm[5], rv, rtol=1e-4, atol=1e-6)
        assert sm[6] == 1 and big[6] == 1
    # the two paths agree with each other at least as well as either agrees with torch
    close(sm[0], big[0], dtype, "a: small vs three-launch path")
    close_f32_sum(sm[2], big[2], "dgamma: small vs three-launch path")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,n,c,sp,bn_groups", [("instance", 1, 128, (16, 16, 16), 1), ("instance", 2, 128, (8, 8, 8), 1),
                                                   ("batch", 2, 64, (16, 16, 16), 2), ("batch", 2, 512, (4, 4, 4), 2),
                                                   ("batch", 4, 64, (8, 8, 8), 2)])
def test_normact_small_resident_forms_equal_the_three_launch_path_with_dropout(hip, dtype, kind, n, c, sp, bn_groups):
    """The one-launch kernels keep their rows in registers (normact_small_res_*: S = 1 / 8 slots, one or two statistic groups
    per chunk).  With dropout and with BatchNorm over two statistic groups (the stacked PatchGAN pair) they must reproduce
    the three-launch path on the same input and the same seed: same mask (a is zero at the same elements), activations and
    input gradient to rounding, affine gradients, running statistics after the two groups' updates in order."""
    from unet_bssfp_amd import functional as Fn, ops
    assert ops.norm_is_small(n, *sp, c, bn_groups)
    g = torch.Generator().manual_seed(23)
    z = q(torch.randn(n, c, *sp, generator=g) * 1.3 - 0.4, dtype)
    ga = q(torch.rand(n, c, *sp, generator=g) - 0.5, dtype)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    bn = kind == "batch"
    outs = {}
    for small in (True, False):
        Fn.DropoutState._salt = 0                              # the same salt, hence the same mask, for both paths
        cfg = Fn.NormCfg(kind, c, slope=0.2, p=0.0 if bn else 0.1)
        zd = to_act(z, dtype).requires_grad_(True)
        gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
        rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        nbt = torch.zeros((), dtype=torch.long, device=DEV)
        a = Fn.NormActFn.apply(zd, None, gd, bd, None, cfg, True, rmd if bn else None, rvd if bn else None, False,
                               nbt if bn else None, small, bn_groups)
        a.backward(to_act(ga, dtype))
        outs[small] = (from_act(a.detach(), c), from_act(zd.grad, c), gd.grad.cpu(), bd.grad.cpu(), rmd.cpu(), rvd.cpu(), int(nbt))
    sm, big = outs[True], outs[False]
    if not bn:
        assert torch.equal(sm[0] == 0, big[0] == 0)
        assert 0.08 < (sm[0] == 0).float().mean().item() < 0.12
    close(sm[0], big[0], dtype, "a")
    if dtype == torch.float32:
        torch.testing.assert_close(sm[1], big[1], rtol=1e-3, atol=1e-5)
    else:
        close(sm[1], big[1], dtype, "dz")
    close_f32_sum(sm[2], big[2], "dgamma")
    close_f32_sum(sm[3], big[3], "dbeta")
    if bn:
        torch.testing.assert_close(sm[4], big[4], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sm[5], big[5], rtol=1e-4, atol=1e-6)
        assert sm[6] == big[6] == bn_groups


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,c,sp,drop_p", [(8, 128, (8, 8, 8), 0.0), (8, 256, (4, 4, 4), 0.1), (3, 64, (8, 12, 16), 0.0), (4, 64, (4, 4, 8), 0.1)])
def test_normact_small_bwd_group_chunks_as_workgroups_bit_identical(hip, dtype, n, c, sp, drop_p):
    """mi355_normact_small_bwd with scratch for the per-group sums (base.part: the chunks of statistic groups run as independent
    workgroups, normact_small_affine_kernel sums the affine gradients over the groups) against the form without (one workgroup
    per 16 bytes of channels walks the groups in order): dz and the affine gradients bit for bit, overwrite and accumulate."""
    from unet_bssfp_amd import ops
    assert ops.norm_is_small(n, *sp, c)
    g = torch.Generator().manual_seed(31)
    z = to_act(q(torch.randn(n, c, *sp, generator=g) * 1.2 + 0.3, dtype), dtype)
    da = to_act(q(torch.rand(n, c, *sp, generator=g) - 0.5, dtype), dtype)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.rand(c, generator=g) - 0.5).to(DEV)
    a, mean, rstd = ops.normact_small_fwd(z, n, gamma, beta, 1e-5, 0.1, drop_p, 9)
    outs = []
    for scratch in (False, True):
        dz, dg, db = ops.normact_small_bwd(z, da, n, mean, rstd, gamma, beta, 0.1, drop_p, 9, True, group_scratch=scratch)
        acc_g, acc_b = torch.full((c,), 0.25, device=DEV), torch.full((c,), -0.5, device=DEV)
        ops.normact_small_bwd(z, da, n, mean, rstd, gamma, beta, 0.1, drop_p, 9, True, affine_into=(acc_g, acc_b), accumulate=True,
                              group_scratch=scratch)
        outs.append((dz, dg, db, acc_g, acc_b))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert outs[0][1].abs().max() > 0 and not torch.equal(outs[0][3], outs[0][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_eval_mode(hip, dtype):
    from unet_bssfp_amd import functional as Fn
    g = torch.Generator().manual_seed(9)
    c = 32
    z = q(torch.randn(2, c, 4, 4, 4, generator=g), dtype)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    rm, rv = torch.rand(c, generator=g) - 0.5, torch.rand(c, generator=g) + 0.5
    z_cpu = z.clone().requires_grad_(True)
    a_ref = F.leaky_relu(F.batch_norm(z_cpu, rm.clone(), rv.clone(), gamma, beta, False, 0.1, 1e-5), 0.2)
    ga = q(torch.rand(a_ref.shape, generator=g) - 0.5, dtype)
    a_ref.backward(ga)
    cfg = Fn.NormCfg("batch", c, slope=0.2)
    zd = to_act(z, dtype).requires_grad_(True)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    a = Fn.NormActFn.apply(zd, None, gamma.to(DEV), beta.to(DEV), None, cfg, False, rmd, rvd)
    close(from_act(a, c), a_ref.detach(), dtype, "a")
    a.backward(to_act(ga, dtype))
    close(from_act(zd.grad, c), z_cpu.grad, dtype, "dz")
    assert torch.equal(rmd.cpu(), rm) and torch.equal(rvd.cpu(), rv)     # eval: buffers untouched


def test_norm_single_value_raises(hip):
    from unet_bssfp_amd import functional as Fn
    z = to_act(torch.rand(1, 32, 1, 1, 1), torch.float32)
    cfg = Fn.NormCfg("batch", 32, slope=0.2)
    with pytest.raises(ValueError):
        Fn.NormActFn.apply(z, None, torch.ones(32, device=DEV), torch.zeros(32, device=DEV), None, cfg, True,
                           torch.zeros(32, device=DEV), torch.ones(32, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout_statistics_and_backward_mask(hip, dtype):
    from unet_bssfp_amd import functional as Fn
    torch.manual_seed(0)
    c, p = 32, 0.05
    z = q(torch.randn(1, c, 16, 16, 16), dtype)
    cfg = Fn.NormCfg("instance", c, slope=0.1, p=p)
    zd = to_act(z, dtype).requires_grad_(True)
    ones, zeros = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    a = Fn.NormActFn.apply(zd, None, ones, zeros, None, cfg, True, None, None)
    av = from_act(a, c)
    dropped = (av == 0).float().mean().item()
    assert abs(dropped - p) < 0.004                       # ~131k elements: sigma ~ 6e-4
    ref = F.leaky_relu(F.instance_norm(z, eps=1e-5), 0.1) / (1 - p)
    keep = av != 0
    close(av[keep], ref[keep], dtype, "kept")
    # eval mode: no dropout
    a_eval = Fn.NormActFn.apply(zd, None, ones, zeros, None, cfg, False, None, None)
    close(from_act(a_eval, c), ref * (1 - p), dtype, "eval")
    # backward uses the same mask: dz is exactly 0-contribution where dropped => check via linearity
    a.backward(to_act(torch.ones(1, c, 16, 16, 16), dtype))
    assert torch.isfinite(zd.grad.float()).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_fwd_bwd(hip, dtype):
    from unet_bssfp_amd import functional as Fn
    g = torch.Generator().manual_seed(11)
    x = q(torch.randn(2, 32, 4, 8, 8, generator=g), dtype)
    x[:, :, :2, :2, :2] = 0.0                             # ties: first max in scan order takes the gradient
    x_cpu = x.clone().requires_grad_(True)
    y_ref = F.max_pool3d(x_cpu, 2)
    gy = q(torch.rand(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = to_act(x, dtype).requires_grad_(True)
    y = Fn.MaxPoolFn.apply(xd)
    assert torch.equal(from_act(y, 32), y_ref.detach())
    y.backward(to_act(gy, dtype))
    assert torch.equal(from_act(xd.grad, 32), x_cpu.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_odd_extents_floor(hip, dtype):
    """MaxPool3d(2) floors odd extents (the levels MONAI's replicate-pad branch exists for): the last plane / row / column is
    in no window, its gradient is zero -- and dx is still fully written (no stale memory)."""
    from unet_bssfp_amd import functional as Fn
    g = torch.Generator().manual_seed(12)
    x = q(torch.randn(2, 16, 5, 7, 9, generator=g), dtype)
    x_cpu = x.clone().requires_grad_(True)
    y_ref = F.max_pool3d(x_cpu, 2)
    gy = q(torch.rand(y_ref.shape, generator=g), dtype)
    y_ref.backward(gy)
    xd = to_act(x, dtype).requires_grad_(True)
    y = Fn.MaxPoolFn.apply(xd)
    assert torch.equal(from_act(y, 16), y_ref.detach())
    y.backward(to_act(gy, dtype))
    assert torch.equal(from_act(xd.grad, 16), x_cpu.grad)
    from unet_bssfp_amd import _lib
    with pytest.raises(_lib.Mi355Error):
        _ops().maxpool2_fwd(to_act(torch.rand(1, 16, 1, 4, 4), torch.float32))      # an extent below the window


def test_l1_loss_fwd_bwd(hip):
    from unet_bssfp_amd import l1_loss
    g = torch.Generator().manual_seed(13)
    a = torch.rand(2, 6, 9, 10, 11, generator=g)
    b = torch.rand(2, 6, 9, 10, 11, generator=g)
    a_cpu = a.clone().requires_grad_(True)
    (F.l1_loss(a_cpu, b) * 50.0).backward()
    ad = a.to(DEV).requires_grad_(True)
    loss = l1_loss(ad, b.to(DEV))
    (loss * 50.0).backward()
    torch.testing.assert_close(loss.cpu(), F.l1_loss(a, b), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ad.grad.cpu(), a_cpu.grad, rtol=1e-6, atol=1e-10)


def test_cpu_tensor_is_rejected_loudly(hip):
    from unet_bssfp_amd import Generator, _lib
    gen = Generator("bssfp")
    with pytest.raises(_lib.Mi355Error):
        gen(torch.rand(1, 24, 32, 32, 32))


# ------------------------------------------------------------------ space-to-depth PatchGAN path
def to_s2d(x, dtype, cp):
    """NCDHW f32 CPU tensor -> S(x) on the GPU (channels padded to cp per block)."""
    ops = _ops()
    n, c, d, h, w = x.shape
    out = torch.zeros(ops.s2d_shape(n, d, h, w, cp), dtype=dtype, device=DEV)
    ops.pack_ncdhw_s2d(x.to(DEV).float().contiguous(), out, cp, 0, cp)
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_s2d_layout_definition(hip, dtype):
    g = torch.Generator().manual_seed(3)
    x = q(torch.rand(2, 5, 4, 6, 8, generator=g), dtype)
    s = to_s2d(x, dtype, 16).float().cpu()                       # (N, 3, 4, 5, 128)
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1))           # x[-1 .. D]
    for blk in range(8):
        bd, bh, bw = blk >> 2, (blk >> 1) & 1, blk & 1
        ref = xp[:, :, bd::2, bh::2, bw::2][:, :, :3, :4, :5].permute(0, 2, 3, 4, 1)   # a[2j+b-1]
        assert torch.equal(s[..., blk * 16: blk * 16 + 5], ref), blk
        assert float(s[..., blk * 16 + 5: (blk + 1) * 16].abs().max()) == 0.0
    back = _ops().unpack_ncdhw_s2d(s.to(DEV).to(dtype), 5, (4, 6, 8), 16, 0).cpu()
    assert torch.equal(back, x)


# bf16 plans of the wide cases: 22421 = conv_march2_kernel<2> (the marching k2 kernel: 8-row footprints, two workgroups per CU),
# forward and -- where the (extent + 1)-wide space-to-depth gradient tiles well -- data gradient
S2D_PLANS = {
    (1, 30, 32, (32, 64, 128)): [22421, 21022],     # d1's widths: 8 groups of 32 S-channels, one-plane segments
    (2, 30, 32, (64, 64, 128)): [22421, 21022],     # two samples (Discriminator.forward_pair)
    (1, 30, 32, (40, 72, 96)): [22421, 22421],      # ragged in h (36 = 4.5 x 8) and w (48 = 1.5 x 32); gradient 21 x 37 x 49 marches
    (1, 32, 64, (32, 64, 64)): [22421, 21022],      # d2's widths: two output-channel blocks
    (1, 32, 32, (32, 62, 126)): [22421, 22421],     # gradient extents 17 x 32 x 64: the data gradient marches too (padding 1)
}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,cin,cout,sp", [(1, 30, 32, (8, 8, 32)), (2, 32, 64, (8, 8, 8)), (1, 64, 128, (4, 4, 4)),
                                             (2, 256, 32, (4, 4, 4))] + list(S2D_PLANS))      # dS has 2048 channels: split-K, 2 channel blocks
def test_k4s2_conv_as_dense_k2_on_s2d(hip, dtype, n, cin, cout, sp):
    from unet_bssfp_amd import functional as Fn
    wide = (n, cin, cout, sp) in S2D_PLANS
    if wide and dtype == torch.float32:
        pytest.skip("the wide cases pin the bf16 marching k2 kernel; f32 takes the same halo kernel as the small cases")
    plans, kinds = [], []
    _ops().CONV_PROBE = lambda pid, d, real: plans.append(pid)
    _ops().WGRAD_PROBE = lambda kind, d: kinds.append(kind)
    g = torch.Generator().manual_seed(17)
    layer = _conv_layer((cin,), cout, 4, 2, 1, 4)
    with torch.no_grad():
        layer.weight.copy_(q(layer.weight, dtype))
    x = q(torch.rand(n, cin, *sp, generator=g) - 0.3, dtype)
    w_cpu = layer.weight.detach().clone().requires_grad_(True)
    b_cpu = layer.bias.detach().clone().requires_grad_(True)
    x_cpu = x.clone().requires_grad_(True)
    z_ref = F.conv3d(x_cpu, w_cpu, b_cpu, 2, 1)
    gz = q(torch.rand(z_ref.shape, generator=g) - 0.5, dtype)
    z_ref.backward(gz)
    layer = layer.to(DEV)
    cp = _ops().round_up(cin, 16)
    s = to_s2d(x, dtype, cp).requires_grad_(True)
    z, part = Fn.ConvFn.apply(s, None, layer.weight, layer.bias, layer.spec, True, False, cp)
    close(from_act(z, cout), z_ref.detach(), dtype, "z")
    zc = z_ref.detach() - b_cpu.detach().view(1, -1, 1, 1, 1)
    e0 = (part.sum(0).cpu()[0, :cout] - zc.sum((0, 2, 3, 4))).abs()
    assert bool((e0 <= 2e-3 * ((zc.numel() / cout) * (zc * zc).sum((0, 2, 3, 4))).sqrt() + 1e-6).all()), e0.max()
    try:
        z.backward(to_act(gz, dtype))
    finally:
        _ops().CONV_PROBE = None
        _ops().WGRAD_PROBE = None
    if wide:
        assert plans == S2D_PLANS[(n, cin, cout, sp)], plans
        assert kinds == [4], kinds                                       # wgrad_march2_kernel (dense 2x2x2, W >= 32)
    dx = _ops().unpack_ncdhw_s2d(s.grad, cin, sp, cp, 0).cpu()
    close(dx, x_cpu.grad, dtype, "dx")
    close_f32_sum(layer.weight.grad.cpu(), w_cpu.grad, "dw")
    close_f32_sum(layer.bias.grad.cpu(), b_cpu.grad, "db")


@pytest.mark.parametrize("ny,summed", [(1, True), (2, True), (2, False)])
def test_split_s2d_conv_equals_the_conv_of_the_concatenation(hip, ny, summed):
    """Fn.SplitS2dConvFn -- the PatchGAN's first block (src/model.py:72-73, 86-87: Conv3d(30, 32, k4, s2, p1) of cat([x, y], 1))
    as x-part (f32, computed once, the accumulators' start value of the second launch) + y-part -- against torch's convolution
    of the concatenation on the CPU: z, fused statistics, the gradient of the y part, the whole weight gradient (both channel
    slices) and the bias gradient.  ny = 2: two stacked y batches over ONE x (Discriminator.forward_pair: add_n); the x-part of
    the weight gradient then runs on the sum of the two gradients (default) or over x once per half (xn)."""
    from unet_bssfp_amd import functional as Fn
    Fn.SplitS2dConvFn.sum_pair_gradients = summed
    dtype, cx, cy, cout, sp = torch.bfloat16, 24, 6, 32, (64, 64, 64)
    g = torch.Generator().manual_seed(23)
    layer = _conv_layer((cx + cy,), cout, 4, 2, 1, 6)
    with torch.no_grad():
        layer.weight.copy_(q(layer.weight, dtype))
    x = q(torch.rand(1, cx, *sp, generator=g) - 0.3, dtype)
    y = q(torch.rand(ny, cy, *sp, generator=g) - 0.3, dtype)
    w_cpu = layer.weight.detach().clone().requires_grad_(True)
    b_cpu = layer.bias.detach().clone().requires_grad_(True)
    y_cpu = y.clone().requires_grad_(True)
    z_ref = F.conv3d(torch.cat([x.expand(ny, -1, -1, -1, -1), y_cpu], 1), w_cpu, b_cpu, 2, 1)
    gz = q(torch.rand(z_ref.shape, generator=g) - 0.5, dtype)
    z_ref.backward(gz)
    layer = layer.to(DEV)
    sx, sy = to_s2d(x, dtype, 32), to_s2d(y, dtype, 8).requires_grad_(True)
    plans = []
    _ops().CONV_PROBE = lambda pid, d, real: plans.append((pid, bool(d.addend), d.y_f32, d.add_n))
    try:
        Fn.StepMemo.clear()
        z, part = Fn.SplitS2dConvFn.apply(sx, sy, layer.weight, layer.bias, layer.spec, cx, cy, True)
        z2, _ = Fn.SplitS2dConvFn.apply(sx, sy, layer.weight, layer.bias, layer.spec, cx, cy, False)     # x-part from the memo
    finally:
        _ops().CONV_PROBE = None
    # x-part once (f32 out), y-part twice with the addend; all on the marching k2 kernel
    assert [p[1:] for p in plans] == [(False, 1, 0), (True, 0, 1 if ny > 1 else 0), (True, 0, 1 if ny > 1 else 0)], plans
    assert all((p[0] % 10000) // 100 == 24 for p in plans), plans
    assert torch.equal(z, z2)
    close(from_act(z, cout), z_ref.detach(), dtype, "z")
    zc = z_ref.detach() - b_cpu.detach().view(1, -1, 1, 1, 1)
    e0 = (part.sum(0).cpu()[0, :cout] - zc.sum((0, 2, 3, 4))).abs()
    assert bool((e0 <= 2e-3 * ((zc.numel() / cout) * (zc * zc).sum((0, 2, 3, 4))).sqrt() + 1e-6).all()), e0.max()
    close_f32_sum(part.sum(0).cpu()[1, :cout], (zc * zc).sum((0, 2, 3, 4)), "sum (z-b)^2")
    try:
        z.backward(to_act(gz, dtype))
    finally:
        Fn.SplitS2dConvFn.sum_pair_gradients = True
    dy = _ops().unpack_ncdhw_s2d(sy.grad, cy, sp, 8, 0).cpu()
    close(dy, y_cpu.grad, dtype, "dy")
    close_f32_sum(layer.weight.grad.cpu(), w_cpu.grad, "dw")
    close_f32_sum(layer.bias.grad.cpu(), b_cpu.grad, "db")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_normact_writes_and_reads_s2d(hip, dtype):
    from unet_bssfp_amd import functional as Fn
    g = torch.Generator().manual_seed(23)
    n, c, sp = 2, 32, (4, 8, 8)
    z = q(torch.randn(n, c, *sp, generator=g), dtype)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5
    cfg = Fn.NormCfg("batch", c, slope=0.2)
    zd1 = to_act(z, dtype).requires_grad_(True)
    zd2 = to_act(z, dtype).requires_grad_(True)
    mk = lambda: (torch.zeros(c, device=DEV), torch.ones(c, device=DEV))
    rm1, rv1 = mk()
    rm2, rv2 = mk()
    a_plain = Fn.NormActFn.apply(zd1, None, gamma.to(DEV), beta.to(DEV), None, cfg, True, rm1, rv1, False)
    a_s2d = Fn.NormActFn.apply(zd2, None, gamma.to(DEV), beta.to(DEV), None, cfg, True, rm2, rv2, True)
    assert a_s2d.shape == (n, 3, 5, 5, 8 * c)
    ref = to_s2d(from_act(a_plain, c), dtype, c)
    assert torch.equal(a_s2d.float().cpu(), ref.float().cpu())
    ga = q(torch.rand(n, c, *sp, generator=g) - 0.5, dtype)
    a_plain.backward(to_act(ga, dtype))
    a_s2d.backward(to_s2d(ga, dtype, c))
    assert torch.equal(zd1.grad.float().cpu(), zd2.grad.float().cpu())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_repack_matches_single_pack(hip, dtype):
    """After an optimiser step the packed weights are rebuilt by mi355_weight_pack_multi (dense 3x3x3 packings
    through the LDS-transposing kernel, the rest through the gather kernel): must equal the single-descriptor
    pack bit for bit, for forward and data-gradient layouts, padded channel counts included."""
    from unet_bssfp_amd import functional as Fn, nn as N
    ops = _ops()
    torch.manual_seed(0)
    net = torch.nn.ModuleList([N.Conv3d(24, 32, 3, 1, 1), N.Conv3d(96, 32, 3, 1, 1), N.Conv3d(32, 6, 3, 1, 1),
                               N.Conv3d(64, 128, 3, 1, 1), N.Conv3d(24, 24, 1, 1, 0), N.ConvTranspose3d(64, 32),
                               N.Conv3d(30, 32, 4, 2, 1)]).to(DEV)
    packed = []
    for m in net:
        s, w = m.spec, m.weight
        cinp = ops.round_up(s.cin, 16)
        if s.kind == "conv" and s.stride == 1:
            packed += [s.w_fwd(w, dtype, cinp)[0], s.w_dgrad_s1(w, dtype, ops.round_up(s.cout, 16), cinp)[0]]
        elif s.kind == "conv":
            packed += [s.w_fwd_s2d(w, dtype, 32)[0], s.w_dgrad_s2d(w, dtype, 32, 32)[0]]
        else:
            packed += [s.w_deconv_fwd_all(w, dtype, cinp)[0], s.w_deconv_dgrad(w, dtype, ops.round_up(s.cout, 16))[0]]
    with torch.no_grad():
        for m in net:
            m.weight.mul_(1.5).add_(0.25)                     # "optimiser step": versions bump, caches go stale
    Fn.repack_weights(net)                                    # batched path, in place
    batched = [p.clone() for p in packed]
    for m in net:
        m.spec.cache.clear()                                  # rebuild through the single-descriptor path
    k = 0
    for m in net:
        s, w = m.spec, m.weight
        cinp = ops.round_up(s.cin, 16)
        if s.kind == "conv" and s.stride == 1:
            fresh = [s.w_fwd(w, dtype, cinp)[0], s.w_dgrad_s1(w, dtype, ops.round_up(s.cout, 16), cinp)[0]]
        elif s.kind == "conv":
            fresh = [s.w_fwd_s2d(w, dtype, 32)[0], s.w_dgrad_s2d(w, dtype, 32, 32)[0]]
        else:
            fresh = [s.w_deconv_fwd_all(w, dtype, cinp)[0], s.w_deconv_dgrad(w, dtype, ops.round_up(s.cout, 16))[0]]
        for f in fresh:
            assert torch.equal(f, batched[k]), (type(m).__name__, k)
            k += 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_skip_pool_sums_both_gradients_in_the_pool_backward(hip, dtype):
    """SkipPoolFn: one node for the two uses of an encoder level (skip connection + MaxPool3d(2)); its backward adds
    the skip gradient -- here a channel slice of a wider buffer, as the concat conv's data gradient is -- inside the
    max-pool backward kernel.  Reference: plain autograd on the CPU."""
    from unet_bssfp_amd import functional as Fn
    g = torch.Generator().manual_seed(9)
    x = q(torch.rand(2, 32, 8, 12, 16, generator=g), dtype)
    gs = q(torch.rand(2, 32, 8, 12, 16, generator=g) - 0.5, dtype)
    gp = q(torch.rand(2, 32, 4, 6, 8, generator=g) - 0.5, dtype)
    xr = x.clone().requires_grad_(True)
    (xr * gs).sum().backward(retain_graph=True)
    (F.max_pool3d(xr, 2) * gp).sum().backward()
    a = to_act(x, dtype).requires_grad_(True)
    skip, pooled = Fn.SkipPoolFn.apply(a)
    assert torch.equal(skip, a) and torch.equal(from_act(pooled, 32), F.max_pool3d(x, 2))
    wide = torch.zeros(2, 8, 12, 16, 48, dtype=dtype, device=DEV)
    wide[..., :32] = to_act(gs, dtype)
    torch.autograd.backward([skip, pooled], [wide[..., :32], to_act(gp, dtype)])
    close(from_act(a.grad, 32), xr.grad, dtype, "skip + pool")
    # only one of the two uses reaches the loss
    a2 = to_act(x, dtype).requires_grad_(True)
    s2, p2 = Fn.SkipPoolFn.apply(a2)
    p2.backward(to_act(gp, dtype))
    x2 = x.clone().requires_grad_(True)
    (F.max_pool3d(x2, 2) * gp).sum().backward()
    close(from_act(a2.grad, 32), x2.grad, dtype, "pool only")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_source_pack_equals_two_single_packs(hip, dtype):
    """mi355_pack2_ncdhw(_s2d): torch.cat([x, y], 1) in one pass == the two single-source packs, plain and S() layout"""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x, y = torch.rand(2, 24, 4, 6, 8, generator=g).to(DEV), torch.rand(2, 6, 4, 6, 8, generator=g).to(DEV)
    a = ops.new_act(2, 4, 6, 8, 32, dtype, DEV); a.fill_(7)
    b = ops.new_act(2, 4, 6, 8, 32, dtype, DEV); b.fill_(7)
    ops.pack_ncdhw(x, a, 0, 24); ops.pack_ncdhw(y, a, 24, 32)
    ops.pack2(x, y, b, 0, 32)
    assert torch.equal(a, b)
    sa = torch.zeros(ops.s2d_shape(2, 4, 6, 8, 32), dtype=dtype, device=DEV)
    sb = torch.zeros_like(sa)
    ops.pack_ncdhw_s2d(x, sa, 32, 0, 24); ops.pack_ncdhw_s2d(y, sa, 32, 24, 32)
    ops.pack2(x, y, sb, 0, 32, s2d_cblk=32)
    assert torch.equal(sa, sb)


# ------------------------------------------------------------------ UpCat's up-branch as one transposed convolution (csrc/upcat.hip)
def _compose_ref(wd, wc_up):
    """k4[ci][co][4][4][4] = full correlation of the transposed convolution's 2x2x2 kernel with the 3x3x3 kernel of the
    concatenated convolution's up-branch, summed over the intermediate channels: conv_transpose3d of wd (as a batch of
    (cu, 2, 2, 2) images) with the spatially FLIPPED wc (t = a - k + 2  <=>  t = a + (2 - k))."""
    return F.conv_transpose3d(wd, wc_up.flip(2, 3, 4).permute(1, 0, 2, 3, 4).contiguous())       # weight: (cu, co, 3, 3, 3)


def test_upcat_compose_chain_and_border_sums_match_torch(hip):
    """The small kernels of the fused up-branch against plain torch on the CPU: k4 and its depth-to-space packing, the bias
    vector and the 27 border-class corrections, the border-region sums of a gradient, and the chain rule from dk4 back to
    dW_d / dW_c[:, ce:] / db_d (autograd of the torch formula)."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    cl, cu, ce, co = 64, 48, 32, 32
    wd = (torch.rand(cl, cu, 2, 2, 2, generator=g) - 0.5).requires_grad_(True)
    wc = (torch.rand(co, ce + cu, 3, 3, 3, generator=g) - 0.5).requires_grad_(True)
    bd = (torch.rand(cu, generator=g) - 0.5).requires_grad_(True)
    bc = torch.rand(co, generator=g) - 0.5
    k4_ref = _compose_ref(wd, wc[:, ce:])
    k4, wp, biasp, delta = ops.upcat_compose(wd.detach().to(DEV), wc.detach().to(DEV), bd.detach().to(DEV), bc.to(DEV), ce)
    torch.testing.assert_close(k4.cpu(), k4_ref.detach(), rtol=1e-5, atol=1e-5)
    # packing: wp[ci >> 4][e][blk * co + o][ci & 15] = k4[ci][o][3 - b - 2 e] per axis
    wpf = wp.float().cpu()
    for blk in range(8):
        for e in range(8):
            t = [3 - ((blk >> s) & 1) - 2 * ((e >> s) & 1) for s in (2, 1, 0)]
            ref = k4.cpu()[:, :, t[0], t[1], t[2]]                                          # [ci][o] (the kernel's own f32 values)
            got = wpf[:, e, blk * co:(blk + 1) * co, :].permute(0, 2, 1).reshape(cl, co)    # [chunk][o][16] -> [ci][o]
            torch.testing.assert_close(got, ref.to(torch.bfloat16).float(), rtol=0, atol=0)
    # bias tables: z_up of a ZERO input is the bias term alone
    d, h, w = 4, 6, 8
    up0 = bd.detach().view(1, cu, 1, 1, 1).expand(1, cu, d, h, w)
    zb = F.conv3d(up0, wc.detach()[:, ce:], bc, 1, 1)[0]                                    # [co][d][h][w]
    cls = lambda i, n_: 0 if i == 0 else (2 if i == n_ - 1 else 1)
    got = torch.empty_like(zb)
    for i in range(d):
        for j in range(h):
            for k in range(w):
                got[:, i, j, k] = biasp.cpu() + delta.cpu()[cls(i, d) * 9 + cls(j, h) * 3 + cls(k, w)]
    torch.testing.assert_close(got, zb, rtol=1e-5, atol=1e-5)
    # border sums of a gradient tensor
    for dtype in (torch.float32, torch.bfloat16):
        gz = q(torch.rand(2, co, 6, 10, 12, generator=g) - 0.5, dtype)
        e = ops.border_sums(to_act(gz, dtype), co).cpu()
        sel = lambda s_, n_: slice(None) if s_ == 0 else (slice(0, 1) if s_ == 1 else slice(n_ - 1, n_))
        for sd in range(3):
            for sh in range(3):
                for sw in range(3):
                    ref = gz[:, :, sel(sd, 6), sel(sh, 10), sel(sw, 12)].sum((0, 2, 3, 4)) if (sd or sh or sw) else torch.zeros(co)
                    torch.testing.assert_close(e[sd * 9 + sh * 3 + sw], ref, rtol=1e-4, atol=1e-4)
    # chain rule: loss = <dk4, k4> + the bias path <dz, bias term> with dz summed over border regions
    dk4 = torch.rand(cl, co, 4, 4, 4, generator=g) - 0.5
    gz = torch.rand(1, co, 6, 10, 12, generator=g) - 0.5
    gz = gz - gz.mean((2, 3, 4), keepdim=True)                  # zero sum per channel, like a normalisation's input gradient
    up0 = bd.view(1, cu, 1, 1, 1).expand(1, cu, 6, 10, 12)
    loss = (dk4 * k4_ref).sum() + (F.conv3d(up0, wc[:, ce:], None, 1, 1) * gz).sum()
    loss.backward()
    esum = ops.border_sums(to_act(gz, torch.float32), co)
    dwd = torch.zeros(cl, cu, 2, 2, 2, device=DEV)
    dwc = torch.full((co, ce + cu, 3, 3, 3), 7.0, device=DEV)
    dbd = torch.zeros(cu, device=DEV)
    ops.upcat_chain(dk4.to(DEV), wd.detach().to(DEV), wc.detach().to(DEV), bd.detach().to(DEV), esum, ce, dwd, dwc, dbd, False)
    torch.testing.assert_close(dwd.cpu(), wd.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dwc.cpu()[:, ce:], wc.grad[:, ce:], rtol=1e-4, atol=1e-4)
    assert float((dwc.cpu()[:, :ce] - 7.0).abs().max()) == 0.0            # the skip part belongs to another launch
    torch.testing.assert_close(dbd.cpu(), bd.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,ce,cl,cu,co,low", [(1, 32, 64, 64, 32, (24, 32, 64)), (2, 32, 64, 64, 32, (10, 24, 40)),
                                                (1, 64, 128, 64, 64, (16, 16, 32))])
def test_upcat_fused_up_branch_matches_torch(hip, n, ce, cl, cu, co, low):
    """Fn.UpCatConvFn (bf16) -- ConvTranspose3d(k2, s2) + concatenation + Conv3d(k3, p1) of MONAI's UpCat (BasicUNet at
    src/model.py:22-28) without the up-sampled tensor -- against the three torch ops on the CPU: z, the fused statistics,
    every input and parameter gradient.  Shapes: upcat_1 (64 -> 64 up, 96 -> 32), two samples with ragged tiles, upcat_2."""
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd.nn import Conv3d, ConvTranspose3d
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(37)
    torch.manual_seed(11)
    deconv, conv = ConvTranspose3d(cl, cu), Conv3d(ce + cu, co, 3, 1, 1)
    with torch.no_grad():
        deconv.weight.copy_(q(deconv.weight, dtype)); conv.weight.copy_(q(conv.weight, dtype))
        deconv.bias.mul_(4.0)                                        # a deconv bias that matters at the border
    skip = tuple(2 * e for e in low)
    x_e = q(torch.rand(n, ce, *skip, generator=g) - 0.3, dtype).requires_grad_(True)
    x_l = q(torch.rand(n, cl, *low, generator=g) - 0.3, dtype).requires_grad_(True)
    wd, bd = deconv.weight.detach().clone().requires_grad_(True), deconv.bias.detach().clone().requires_grad_(True)
    wc, bc = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    z_ref = F.conv3d(torch.cat([x_e, F.conv_transpose3d(x_l, wd, bd, 2)], 1), wc, bc, 1, 1)
    gz = torch.rand(z_ref.shape, generator=g) - 0.5
    gz = q(gz - gz.mean((2, 3, 4), keepdim=True), dtype)              # (zero channel sums up to bf16 rounding: a norm's input gradient)
    z_ref.backward(gz)
    deconv, conv = deconv.to(DEV), conv.to(DEV)
    a_e, a_l = to_act(x_e.detach(), dtype).requires_grad_(True), to_act(x_l.detach(), dtype).requires_grad_(True)
    tables = Fn.UpCatTables()
    plans = []
    _ops().CONV_PROBE = lambda pid, d, real: plans.append((pid, d.d2s, d.y_f32))
    try:
        z, part = Fn.UpCatConvFn.apply(a_e, a_l, deconv.weight, deconv.bias, conv.weight, conv.bias, conv.spec, tables, True)
        # (a) the kernels: against the same arithmetic on the operands the launch reads -- the composite kernel k4 ROUNDED to
        #     bf16 (a derived operand, like the unfused path's bf16 `up`) and the skip part rounded to bf16 between the two launches
        k4q = tables.bufs[0].cpu().to(dtype).float()
        xl2 = x_l.detach().clone().requires_grad_(True)
        bias_term = F.conv3d(bd.detach().view(1, cu, 1, 1, 1).expand(1, cu, *skip), wc.detach()[:, ce:], bc.detach(), 1, 1)
        p_skip = F.conv3d(x_e.detach(), wc.detach()[:, :ce], None, 1, 1)
        z_q = q(p_skip, dtype) + F.conv_transpose3d(xl2, k4q, None, 2, 1) + bias_term
        # the skip part was rounded to bf16 by ITS launch, whose f32 sums differ from the CPU's in the last bits: where they
        # straddle a rounding boundary the two roundings are one bf16 ulp of the skip part apart (2^-8 ... 2^-7 of its magnitude)
        err = (from_act(z, co) - z_q.detach()).abs()
        bound = 1e-2 * z_q.detach().abs() + 2e-3 * float(z_q.detach().std()) + 2.0 ** -7 * p_skip.abs()
        assert bool((err <= bound).all()), float((err - bound).max())
        # (b) the function: against the three torch ops in f32 -- the operand rounding of k4 (2^-9 relative per weight) is what
        #     separates the two, as the rounding of `up` does in the unfused path
        got = from_act(z, co)
        assert float((got - z_ref.detach()).norm() / z_ref.detach().norm()) <= 3e-3
        shift = tables.bufs[2].cpu()
        zc = z_q.detach() - shift.view(1, -1, 1, 1, 1)
        st = part.sum(0).cpu()
        e0 = (st[0, :co] - zc.sum((0, 2, 3, 4))).abs()
        assert bool((e0 <= 2e-3 * ((zc.numel() / co) * (zc * zc).sum((0, 2, 3, 4))).sqrt() + 1e-6).all()), e0.max()
        close_f32_sum(st[1, :co], (zc * zc).sum((0, 2, 3, 4)), "sum (z - shift)^2")
        z.backward(to_act(gz, dtype))
    finally:
        _ops().CONV_PROBE = None
    assert sum(1 for p in plans if p[1]) == 1 and all((p[0] % 10000) // 100 == 24 for p in plans if p[1]), plans
    close(from_act(a_e.grad, ce), x_e.grad, dtype, "dx_e")
    z_q.backward(gz)
    close(from_act(a_l.grad, cl), xl2.grad, dtype, "dx_low (bf16 k4)")
    assert float((from_act(a_l.grad, cl) - x_l.grad).norm() / x_l.grad.norm()) <= 3e-3
    close_f32_sum(conv.weight.grad.cpu(), wc.grad, "dW_c")
    close_f32_sum(deconv.weight.grad.cpu(), wd.grad, "dW_d")
    # db_d lives on the border shell only (the interior cancels exactly behind a normalisation): compare on the scale of the
    # shell's contribution, with the volume sum the reference also carries (bf16 rounding noise of gz) taken out
    gsum = gz.sum((0, 2, 3, 4))
    bd_ref = bd.grad - torch.einsum("ock,o->c", wc.detach()[:, ce:].reshape(co, cu, 27), gsum)
    close_f32_sum(deconv.bias.grad.cpu(), bd_ref, "db_d")
    assert float(conv.bias.grad.abs().max()) == 0.0


@pytest.mark.parametrize("which", ["both", "loss_only", "discr_only"])
def test_generator_output_seam_equals_unpack_add_pack(hip, which):
    """Fn.GenOutFn: the generator's NDHWC output -> (NCDHW f32, S(output)); its backward joins the loss head's NCDHW gradient and
    the PatchGAN's space-to-depth gradient in one pass (mi355_seam_grad).  Against the separate kernels it replaces: unpack,
    pack_ncdhw_s2d forward; unpack_ncdhw_s2d + add + pack_ncdhw backward -- bit for bit (f32 add, one rounding to bf16)."""
    from unet_bssfp_amd import functional as Fn
    ops = _ops()
    dtype, sp, n = torch.bfloat16, (8, 12, 16), 2
    g = torch.Generator().manual_seed(41)
    z = to_act(q(torch.rand(n, 6, *sp, generator=g) - 0.5, dtype), dtype).requires_grad_(True)       # (n, d, h, w, 16)
    y, s = Fn.GenOutFn.apply(z, 6, 8)
    assert torch.equal(y, ops.unpack_ncdhw(z.detach(), 6, 0))
    ref_s = torch.zeros(ops.s2d_shape(n, *sp, 8), dtype=dtype, device=DEV)
    ops.pack_ncdhw_s2d(y.detach(), ref_s, 8, 0, 8)
    assert torch.equal(s, ref_s)
    gy = (torch.rand(n, 6, *sp, generator=g) - 0.5).to(DEV)
    gs = to_s2d(q(torch.rand(n, 6, *sp, generator=g) - 0.5, dtype), dtype, 8)
    outs, grads = [], []
    if which != "discr_only":
        outs.append(y); grads.append(gy)
    if which != "loss_only":
        outs.append(s); grads.append(gs)
    torch.autograd.backward(outs, grads)
    total = torch.zeros(n, 6, *sp, device=DEV)
    if which != "discr_only":
        total = total + gy
    if which != "loss_only":
        total = total + ops.unpack_ncdhw_s2d(gs, 6, sp, 8, 0)
    ref = ops.new_act(n, *sp, 16, dtype, DEV)
    ops.pack_ncdhw(total.contiguous(), ref, 0, 16)
    assert torch.equal(z.grad, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop_p", [0.0, 0.05])
def test_norm_backward_forms_the_maxpool_gradient_itself(hip, dtype, drop_p):
    """Fn.LazyPool: MaxPool3d(2)'s backward (+ the skip connection's gradient) formed per row inside the producing norm + act
    node's backward kernels from the window positions the forward max-pool recorded (mi355_maxpool2_fwd_idx,
    mi355_normact_desc::pool_idx) -- bit-identical to the max-pool backward launch followed by the plain kernels: ops level
    (with and without a skip gradient, ties and a skip gradient that is a channel slice of a wider buffer) and through autograd
    (NormActFn -> SkipPoolFn.apply_to) against the eager graph; the recorded positions are torch's max_pool3d indices."""
    from unet_bssfp_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(29)
    n, c, sp = 2, 32, (8, 12, 32)
    z = to_act(q(torch.randn(n, c, *sp, generator=g) * 1.5 + 0.3, dtype), dtype)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.rand(c, generator=g) - 0.5).to(DEV)
    rows = sp[0] * sp[1] * sp[2]
    part, ppg = ops.channel_stats(z, n)
    mean, rstd = ops.norm_finalize(part, ppg, n, c, rows, None, 1e-5, None, None, 0.1, n_real=0)
    seed_t = Fn.DropoutState.base(DEV) if drop_p > 0 else None
    a = ops.normact_fwd(z, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, seed_t=seed_t)
    # the pool in the launch that writes a (ops.normact_fwd(pool=True), Fn.PoolSide): the same three tensors, bit for bit
    a_f, y_f, idx_f = ops.normact_fwd(z, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, seed_t=seed_t, pool=True)
    y_s, idx_s = ops.maxpool2_fwd(a, want_idx=True)
    assert torch.equal(a_f, a) and torch.equal(y_f, y_s) and torch.equal(idx_f, idx_s)
    a[:, :2, :2, :2, :] = 0.0                              # ties: the first maximum in scan order takes the gradient
    y, idx = ops.maxpool2_fwd(a, want_idx=True)
    assert torch.equal(y, ops.maxpool2_fwd(a))
    # the recorded window position against torch's flat argmax index
    a_nc = a.float().permute(0, 4, 1, 2, 3).cpu()
    _, flat = F.max_pool3d(a_nc, 2, return_indices=True)
    fd, fh, fw = flat // (sp[1] * sp[2]), (flat // sp[2]) % sp[1], flat % sp[2]
    ref_idx = ((fd & 1) * 4 + (fh & 1) * 2 + (fw & 1)).permute(0, 2, 3, 4, 1).to(torch.uint8)
    assert torch.equal(idx.cpu(), ref_idx)
    dy = to_act(q(torch.randn(n, c, sp[0] // 2, sp[1] // 2, sp[2] // 2, generator=g), dtype), dtype)
    wide = to_act(q(torch.randn(n, 2 * c, *sp, generator=g), dtype), dtype)
    for add in (None, wide[..., c:]):
        da = ops.maxpool2_bwd(a, y, dy, add)
        e = ops.normact_bwd(z, da, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, True, True, seed_t=seed_t)
        i = ops.normact_bwd(z, add, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, True, True, seed_t=seed_t, pool=(idx, dy))
        torch.cuda.synchronize()
        for t_e, t_i in zip(e, i):
            assert torch.equal(t_e, t_i)
    # through autograd
    cfg = Fn.NormCfg("instance", c, slope=0.1, p=drop_p)
    outs = []
    try:
        for lazy in (False, True):
            Fn.LazyPool.enabled = Fn.PoolSide.enabled = lazy
            Fn.DropoutState._salt = 0
            zz = z.clone().requires_grad_(True)
            gg, bb = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
            act = Fn.NormActFn.apply(zz, None, gg, bb, None, cfg, True, None, None, False, None, False, 1, None, None, None, True)
            assert bool(Fn.PoolSide._by_ptr) == lazy
            skip, pooled = Fn.SkipPoolFn.apply_to(act)
            torch.autograd.backward([skip, pooled], [wide[..., c:], dy])
            torch.cuda.synchronize()
            outs.append((zz.grad, gg.grad, bb.grad, pooled.detach().clone(), skip.detach().clone()))
        assert not Fn.LazyPool._by_ptr and not Fn.PoolSide._by_ptr
    finally:
        Fn.LazyPool.enabled = Fn.PoolSide.enabled = True
    for t_e, t_i in zip(*outs):
        assert torch.equal(t_e, t_i)


@pytest.mark.parametrize("drop_p", [0.0, 0.05])
def test_norm_backward_forms_the_final_convs_data_gradient_itself(hip, drop_p):
    """mi355_normact_bwd_reduce / _apply with gz / gw (Fn.LazyDx): the data gradient of the 1x1x1 convolution that consumed a
    (BasicUNet.final_conv, src/model.py:22-28) is formed per row inside the norm backward kernels instead of being written by one
    launch and read by two.  Against the explicit path on da = bf16(gz @ W) (torch): dz within one bf16 ulp of the values (the
    two matmuls sum in different orders: a rounding of da can flip), affine gradients at the f32-sum bound; and through
    autograd (ConvFn(lazy_dx) -> NormActFn) against the unfused graph."""
    from unet_bssfp_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(23)
    n, c, sp, k = 2, 32, (8, 12, 32), 6
    dt = torch.bfloat16
    z = to_act(q(torch.randn(n, c, *sp, generator=g) * 1.5 + 0.3, dt), dt)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), (torch.rand(c, generator=g) - 0.5).to(DEV)
    w = (torch.randn(k, c, 1, 1, 1, generator=g) * 0.2).to(DEV)
    gz = to_act(q(torch.randn(n, 16, *sp, generator=g) * 0.1, dt), dt)
    gz[..., k:] = 0
    rows = sp[0] * sp[1] * sp[2]
    part, ppg = ops.channel_stats(z, n)
    mean, rstd = ops.norm_finalize(part, ppg, n, c, rows, None, 1e-5, None, None, 0.1, n_real=0)
    seed_t = Fn.DropoutState.base(DEV) if drop_p > 0 else None
    da = (gz[..., :k].float() @ w.reshape(k, c).to(dt).float()).to(dt).contiguous()     # (the kernels round W to bf16, like a packed weight)
    dz_e, dg_e, db_e = ops.normact_bwd(z, da, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, True, True, seed_t=seed_t)
    dz_i, dg_i, db_i = ops.normact_bwd(z, None, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, True, True, seed_t=seed_t,
                                       implicit=(gz, w))
    torch.cuda.synchronize()
    e, i = dz_e.float(), dz_i.float()
    # a flipped rounding of one da element moves dz by rstd * gamma * ulp(da): bounded by 2^-7 of the largest |da| * rstd * gamma
    bound = 2.0 ** -7 * float(da.float().abs().max()) * float((rstd.max() * gamma.max()))
    assert float((e - i).abs().max()) <= bound, (float((e - i).abs().max()), bound)
    assert float((e - i).abs().mean()) <= 1e-2 * float(e.abs().mean())
    close_f32_sum(dg_i.cpu(), dg_e.cpu(), "dgamma")
    close_f32_sum(db_i.cpu(), db_e.cpu(), "dbeta")

    # forward: the same convolution evaluated by the launch that writes a (ops.normact_fwd, final=), with and without storing a
    fb = (torch.randn(k, generator=g) * 0.3).to(DEV)
    a_ref = ops.normact_fwd(z, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, seed_t=seed_t)
    y_ref = (a_ref.float() @ w.reshape(k, c).to(dt).float().t() + fb).to(dt)
    for skip in (False, True):
        y = torch.full((n, *sp, 16), float("nan"), dtype=dt, device=DEV)
        a2 = ops.normact_fwd(z, n, mean, rstd, gamma, beta, 0.1, drop_p, 5, seed_t=seed_t, final=(w, fb, y), skip_a=skip)
        torch.cuda.synchronize()
        if not skip:
            assert torch.equal(a2, a_ref)
        assert float(y[..., k:].float().abs().max()) == 0.0
        d = (y[..., :k].float() - y_ref.float()).abs()
        assert bool((d <= 2.0 ** -7 * y_ref.float().abs() + 1e-6).all()), float(d.max())
        assert float((d > 0).float().mean()) < 0.05          # (a different summation order: rare one-ulp flips)

    # through autograd: final conv with lazy_dx on the output of a norm + act node
    cfg = Fn.NormCfg("instance", c, slope=0.1, p=drop_p)
    spec = Fn.ConvSpec("conv", c, k, 1, 1, 0)
    bias = torch.zeros(k, device=DEV)
    outs = []
    for lazy in (False, True):
        Fn.DropoutState._salt = 0
        zz = z.clone().requires_grad_(True)
        gg, bb, ww = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), w.clone().requires_grad_(True)
        fin = (ww, bias, False) if lazy else None            # lazy: forward fused as well (Fn.FusedFinal)
        a = Fn.NormActFn.apply(zz, None, gg, bb, None, cfg, True, None, None, False, None, False, 1, None, None, fin)
        y, _ = Fn.ConvFn.apply(a, None, ww, bias, spec, False, False, 0, False, lazy)
        y.backward(gz)
        torch.cuda.synchronize()
        outs.append((zz.grad.float(), gg.grad, bb.grad, ww.grad, y.detach().float()))
    assert not Fn.LazyDx._by_ptr and not Fn.FusedFinal._by_ptr
    dy = (outs[0][4] - outs[1][4]).abs()
    assert bool((dy <= 2.0 ** -7 * outs[0][4].abs() + 1e-6).all()), float(dy.max())
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2 * bound
    for j in (1, 2, 3):
        close_f32_sum(outs[1][j].cpu(), outs[0][j].cpu(), f"grad {j}")
