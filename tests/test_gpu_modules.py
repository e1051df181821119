"""GPU parity of the drop-in modules and of the full GAN training step.

Checked against (i) the committed golden vectors produced by the reference's own classes
(tests/golden/*.npz, see oracle/gen_golden.py) and (ii) the CPU oracle on the same seeded inputs.
f32 path tolerance: per-voxel L1 <= 1e-4 (the north-star figure) and rtol 1e-3 on gradient digests.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _digest(p):
    g = p.grad.double()
    return np.array([g.sum().item(), g.abs().sum().item()])


def noisy_bias(n):
    """Generator conv biases directly in front of a normalisation (heads, TwoConv): their gradient
    is analytically zero, both sides hold rounding noise only (AdamW turns it into +-lr steps)."""
    return n.endswith("conv.bias") and "deconv" not in n and "final" not in n


def _check_grad_digests(gold, prefix, module, rtol=2e-3, atol=2e-5, skip=(), skip_fn=None):
    for n, p in module.named_parameters():
        ref = gold[f"{prefix}/{n}"]
        if any(s in n for s in skip) or (skip_fn is not None and skip_fn(n)):
            continue
        if p.grad is None:
            assert np.isnan(ref).all(), n
            continue
        d = _digest(p)
        # the signed sum cancels: compare it on the scale of the abs-sum
        assert abs(d[1] - ref[1]) <= rtol * abs(ref[1]) + atol, (n, d, ref)
        assert abs(d[0] - ref[0]) <= rtol * abs(ref[1]) + atol, (n, d, ref)


DS_CFGS = {
    "k1_bn_act": dict(in_channels=24, out_channels=24, kernel=1, strides=1, padding=0),
    "k4s2_bn_act": dict(in_channels=32, out_channels=64),
    "k4s2_nobn": dict(in_channels=30, out_channels=32, batchnorm=False),
    "k4s2_noact": dict(in_channels=16, out_channels=32, activation=False),
}


@pytest.mark.parametrize("idx,name", list(enumerate(DS_CFGS)))
def test_downsample_conv_golden(hip, golden_dir, idx, name):
    from unet_bssfp_amd import DownSampleConv
    gold = _gold(golden_dir, "downsample_conv.npz")
    kw = DS_CFGS[name]
    torch.manual_seed(100 + idx)
    m = DownSampleConv(**kw).to(DEV).train()
    g = torch.Generator().manual_seed(200 + idx)
    x = torch.rand(2, kw["in_channels"], 16, 16, 16, generator=g)
    xd = x.to(DEV).requires_grad_(True)
    y = m(xd)
    w = torch.rand(y.shape, generator=g)
    (y * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), gold[f"{name}/y"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), gold[f"{name}/dx"], rtol=1e-3, atol=2e-5)
    _check_grad_digests(gold, f"{name}/grad", m, skip=("conv.bias",) if kw.get("batchnorm", True) else ())
    if kw.get("batchnorm", True):
        np.testing.assert_allclose(m.bn.running_mean.cpu().numpy(), gold[f"{name}/running_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(m.bn.running_var.cpu().numpy(), gold[f"{name}/running_var"], rtol=1e-4, atol=1e-6)
        assert int(m.bn.num_batches_tracked) == 1
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x.to(DEV)).cpu().numpy(), gold[f"{name}/y_eval"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag,modality,n,s,cin", [("bssfp_n1_s64", "bssfp", 1, 64, 24), ("t1w_n2_s32", "t1w", 2, 32, 6)])
def test_discriminator_golden(hip, golden_dir, tag, modality, n, s, cin):
    from unet_bssfp_amd import Discriminator
    gold = _gold(golden_dir, "discriminator.npz")
    torch.manual_seed(7)
    d = Discriminator(modality)
    assert sorted(d.state_dict().keys()) == list(gold[f"{tag}/keys"])
    assert sum(p.numel() for p in d.parameters()) == int(gold[f"{tag}/nparams"])
    d = d.to(DEV).train()
    x, y = R.synthetic_batch(n, s, seed=1234, cin=cin)
    xd, yd = x.to(DEV), y.to(DEV).requires_grad_(True)
    from unet_bssfp_amd import functional as Fn
    Fn.S2D_POISON = True                  # space-to-depth tensors start as NaN: every slot must be written by the kernels
    try:
        logits = d(xd, yd)
    finally:
        Fn.S2D_POISON = False
    assert bool(torch.isfinite(logits).all())
    loss = F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), gold[f"{tag}/logits"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(loss.item(), gold[f"{tag}/loss"], rtol=1e-4)
    ref = gold[f"{tag}/dy_sample"]
    got = yd.grad[:, :, ::7, ::5, ::3].cpu().numpy()
    # LeakyReLU'(z) jumps at z = 0: a voxel whose d1 pre-activation is within rounding of zero takes the other
    # slope on one side, so a handful of isolated elements may differ; everything else must agree tightly
    bad = np.abs(got - ref) > 5e-3 * np.abs(ref) + 2e-3 * np.abs(ref).max()
    assert bad.mean() <= 1e-3, bad.mean()
    assert np.abs(got - ref).max() <= 2e-2 * np.abs(ref).max()
    dig = gold[f"{tag}/dy_digest"]
    got = yd.grad.double()
    assert abs(got.abs().sum().item() - dig[1]) <= 2e-3 * dig[1]
    _check_grad_digests(gold, f"{tag}/grad", d, skip=("d2.conv.bias", "d3.conv.bias", "d4.conv.bias", "d5.conv.bias"))


@pytest.mark.parametrize("tag,modality,cin", [("bssfp", "bssfp", 24), ("dwi", "dwi-tensor", 6)])
def test_generator_golden(hip, golden_dir, tag, modality, cin):
    from unet_bssfp_amd import Generator
    gold = _gold(golden_dir, "generator.npz")
    torch.manual_seed(11)
    g = Generator(modality, dropout=0.0)
    assert sorted(g.state_dict().keys()) == list(gold[f"{tag}/keys"])
    g = g.to(DEV).train()
    x, y = R.synthetic_batch(1, 32, seed=4321, cin=cin)
    xd = x.to(DEV).requires_grad_(True)
    y_hat = g(xd)
    from unet_bssfp_amd import l1_loss
    loss = l1_loss(y_hat, y.to(DEV))
    loss.backward()
    ref = gold[f"{tag}/y_hat"]
    got = y_hat.detach().cpu().numpy()
    assert np.abs(got - ref).mean() <= 1e-4, np.abs(got - ref).mean()          # per-voxel L1 (north star)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(loss.item(), gold[f"{tag}/loss"], rtol=1e-4)
    dref = gold[f"{tag}/dx_sample"]
    # d|y_hat - y| = sign(.) is discontinuous: a few sign flips between CPU and GPU perturb dx by O(1/N)
    np.testing.assert_allclose(xd.grad[:, :, ::5, ::3, ::2].cpu().numpy(), dref, rtol=2e-2, atol=3e-2 * np.abs(dref).max())
    # conv biases in front of a normalisation have analytically zero gradient (rounding noise on both sides)
    _check_grad_digests(gold, f"{tag}/grad", g, rtol=5e-3, atol=1e-6, skip_fn=noisy_bias)
    g.eval()
    with torch.no_grad():
        got = g(x.to(DEV)).cpu().numpy()
    assert np.abs(got - gold[f"{tag}/y_hat_eval"]).mean() <= 1e-4


def test_generator_matches_oracle_with_other_seed_and_batch(hip):
    from unet_bssfp_amd import Generator
    torch.manual_seed(5)
    g = Generator("pc-bssfp", dropout=0.0)
    ref = R.RefGenerator("pc-bssfp", dropout=0.0).train()
    ref.load_state_dict(g.state_dict())
    x, _ = R.synthetic_batch(2, (32, 48, 64), seed=99)
    with torch.no_grad():
        y_ref = ref(x)
        y = g.to(DEV).train()(x.to(DEV)).cpu()
    assert (y - y_ref).abs().mean().item() <= 1e-4
    torch.testing.assert_close(y, y_ref, rtol=2e-3, atol=2e-4)


def test_generator_odd_skip_extents_replicate_pad(hip):
    """MONAI UpCat's is_pad branch: extents that are not divisible by 16 leave odd levels (40 -> 20 -> 10 -> 5 -> 2; the
    up-sampled 4 meets a skip of 5).  Forward AND parameter gradients against the oracle (oracle/unet_ref.py:236-243), f32."""
    from unet_bssfp_amd import Generator
    torch.manual_seed(7)
    g = Generator("bssfp", dropout=0.0)
    ref = R.RefGenerator("bssfp", dropout=0.0).train()
    ref.load_state_dict(g.state_dict())
    x, y = R.synthetic_batch(1, (40, 24, 56), seed=12)
    y_ref = ref(x)
    (y_ref - y).abs().mean().backward()
    g = g.to(DEV).train()
    y_hip = g(x.to(DEV))
    (y_hip - y.to(DEV)).abs().mean().backward()
    assert (y_hip.detach().cpu() - y_ref.detach()).abs().mean().item() <= 1e-4
    torch.testing.assert_close(y_hip.detach().cpu(), y_ref.detach(), rtol=2e-3, atol=2e-4)
    refp = dict(ref.named_parameters())
    for name, p in g.named_parameters():
        if refp[name].grad is None:                                   # the other modalities' input heads
            assert p.grad is None, name
            continue
        gr, gh = refp[name].grad, p.grad.cpu()
        if name.endswith(".conv.bias"):
            # a bias in front of a normalisation has a mathematically zero gradient: the oracle holds 1e-8 .. 1e-6 of
            # round-off, this path (which never adds the bias before the statistics) exactly 0
            assert float(gh.abs().max()) <= 1e-5, name
            continue
        err = (gh - gr).norm()                                        # 5e-3: the bound of the golden-vector test above
        assert err <= 5e-3 * gr.norm() + 1e-7, (name, float(err), float(gr.norm()))


def test_basic_unet_spatial_dims_2_matches_oracle(hip):
    """BASELINE.json configs[0] (2-D U-Net 1 -> 6 channels on 64 x 64 slices): MONAI's 2-D parameter names / shapes and
    default initialisation (same RNG consumption as the torch 2-D modules), forward and parameter gradients vs the oracle's
    2-D network (oracle/unet_ref.py: RefBasicUNet(spatial_dims=2)), f32."""
    from unet_bssfp_amd import BasicUNet
    torch.manual_seed(31)
    net = BasicUNet(spatial_dims=2, in_channels=1, out_channels=6, dropout=0.0)
    torch.manual_seed(31)
    ref = R.RefBasicUNet(spatial_dims=2, in_channels=1, out_channels=6, dropout=0.0).train()
    sd, sr = net.state_dict(), ref.state_dict()
    assert list(sd.keys()) == list(sr.keys())
    for k in sd:
        assert sd[k].shape == sr[k].shape, k
        assert torch.equal(sd[k], sr[k]), k                           # identical default initialisation
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 64, 64, generator=g)
    y = torch.rand(2, 6, 64, 64, generator=g)
    y_ref = ref(x)
    (y_ref - y).abs().mean().backward()
    net = net.to(DEV).train()
    y_hip = net(x.to(DEV))
    assert y_hip.shape == (2, 6, 64, 64)
    (y_hip - y.to(DEV)).abs().mean().backward()
    assert (y_hip.detach().cpu() - y_ref.detach()).abs().mean().item() <= 1e-4
    torch.testing.assert_close(y_hip.detach().cpu(), y_ref.detach(), rtol=2e-3, atol=2e-4)
    refp = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        gr, gh = refp[name].grad, p.grad.cpu()
        assert gh.shape == gr.shape, name
        if name.endswith(".conv.bias"):                               # zero by construction in front of a normalisation
            assert float(gh.abs().max()) <= 1e-5, name
            continue
        err = (gh - gr).norm()
        assert err <= 5e-3 * gr.norm() + 1e-7, (name, float(err), float(gr.norm()))


def test_reference_construction_site_drop_in(hip):
    """A reference-style Generator.forward (src/model.py:36-39) chaining the PUBLIC forwards of our
    DownSampleConv and BasicUNet gives the same result as our fused Generator (zero-copy hand-off)."""
    from unet_bssfp_amd import BasicUNet, DownSampleConv, Generator
    torch.manual_seed(21)
    ours = Generator("bssfp", dropout=0.0).to(DEV).train()
    head = DownSampleConv(24, 24, kernel=1, strides=1, padding=0).to(DEV).train()
    unet = BasicUNet(spatial_dims=3, in_channels=24, out_channels=6, features=(32, 64, 128, 256, 512, 32), dropout=0.0).to(DEV).train()
    head.load_state_dict(ours.blocks["bssfp"].state_dict())
    unet.load_state_dict(ours.blocks["unet"].state_dict())
    x, _ = R.synthetic_batch(1, 32, seed=3)
    xd = x.to(DEV)
    h = head(xd)
    assert h.shape == (1, 24, 32, 32, 32) and getattr(h, "_mi355_act", None) is not None
    y_chain = unet(h)
    y_fused = ours(xd)
    assert torch.equal(y_chain, y_fused)
    # and the public tensors behave like ordinary NCDHW tensors
    ref_head = R.RefDownSampleConv(24, 24, kernel=1, strides=1, padding=0).train()
    ref_head.load_state_dict(head.state_dict())
    torch.testing.assert_close(h.detach().cpu().contiguous(), ref_head(x).detach(), rtol=2e-4, atol=2e-5)


def test_state_dict_roundtrip_with_oracle(hip):
    from unet_bssfp_amd import Discriminator, Generator
    torch.manual_seed(1)
    rg, rd = R.RefGenerator("bssfp"), R.RefDiscriminator("bssfp")
    g, d = Generator("bssfp"), Discriminator("bssfp")
    g.load_state_dict(rg.state_dict())
    d.load_state_dict(rd.state_dict())
    for k, v in rg.state_dict().items():
        assert torch.equal(g.state_dict()[k], v), k
    # shared heads stay shared after loading (src/model.py:29-33, :74)
    assert g.blocks["bssfp"] is g.blocks["pc-bssfp"] and d.d1["t1w"] is d.blocks["dwi-tensor"]


def test_gan_training_step_golden(hip, golden_dir):
    """Two full training steps at 64^3 against the goldens produced by the reference's own classes, with float64 as the
    yard-stick (gan_step_f64.npz: the same two steps in f64) and an ENSEMBLE of CPU f32 runs as the measure of what f32
    rounding alone does (gan_step_f32_spread.npz: the reference classes with 1 / 2 / 3 / 4 / 8 threads and oneDNN off, i.e.
    different f32 summation orders; per quantity the largest deviation from f64).

    Step 0 is compared directly and tightly with the f32 golden.  From the first AdamW update on (lr * g / |g|: rounding
    noise decides the sign of small gradient elements -- 0.15 % of them on BOTH the CPU and the HIP side, measured against
    f64 by tests/diag/diag_grad_noise.py) f32 runs drift apart chaotically: the step-1 adversarial loss of the CPU ensemble
    deviates from f64 by 4e-5 ... 3.2e-3 depending on the thread count alone.  So instead of a chosen tolerance:

        |hip_f32 - f64|  <=  3 * max(ensemble deviation of the quantity, class floor)

    class floor = RMS ensemble deviation of the network's parameter digests: one flipped AdamW sign in a 256-element bias
    moves its digest by up to 1e-3 whether or not any of the 6 ensemble draws happened to flip one there (observed: HIP
    6e-5 on upcat_4's deconv bias at step 0, ensemble 1e-5).  All quantities go to gpurun_out/f64_triangulation.log."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    gold, g64, spread = (_gold(golden_dir, "gan_step.npz"), _gold(golden_dir, "gan_step_f64.npz"),
                         _gold(golden_dir, "gan_step_f32_spread.npz"))
    torch.manual_seed(0)
    gen = M.Generator("bssfp", dropout=0.0)
    discr = M.Discriminator("bssfp")
    model = bSSFPToDWITensorModel("bssfp", gen=gen.to(DEV), discr=discr.to(DEV)).train()
    batch = synthetic_batch(1, 64, seed=1234, device=DEV)
    names = {"train_gen_loss_adversarial": "gen_loss_adversarial", "train_gen_loss_recon_L1": "gen_loss_recon_L1",
             "train_gen_loss_recon": "gen_loss_recon", "train_gen_loss": "gen_loss", "train_discr_loss": "discr_loss"}
    def noisy(tag, n):            # zero-gradient biases in front of a norm: AdamW turns rounding noise (also f64's) into +-lr steps
        if tag == "discr":
            return n in ("d2.conv.bias", "d3.conv.bias", "d4.conv.bias", "d5.conv.bias")
        return noisy_bias(n)
    report, bad = [], []
    for step in range(2):
        model.training_step(batch, step)
        for k, gk in names.items():
            key = f"step{step}/{gk}"
            got = float(model.last_logs[k])
            if step == 0:
                np.testing.assert_allclose(got, gold[key], rtol=1e-3, err_msg=k)
            dev = abs(got - float(g64[key])) / abs(float(g64[key]))
            yard = float(spread[key])
            report.append(f"{key}: hip {dev:.2e} cpu-ensemble {yard:.2e}")
            if dev > 3 * yard + 1e-7:
                bad.append((key, dev, yard))
        for net, tag in ((model.gen, "gen"), (model.discr, "discr")):
            items = [(n, p) for n, p in net.named_parameters() if not noisy(tag, n)]
            floor = float(np.sqrt(np.mean([float(spread[f"step{step}/{tag}/{n}"]) ** 2 for n, _ in items])))
            worst = 0.0
            for n, p in items:
                key = f"step{step}/{tag}/{n}"
                if abs(g64[key][1]) < 1e-12:
                    continue
                dev = abs(p.detach().double().abs().sum().item() - g64[key][1]) / abs(g64[key][1])
                yard = max(float(spread[key]), floor)
                worst = max(worst, dev / yard)
                report.append(f"{key}: hip {dev:.2e} cpu-ensemble {float(spread[key]):.2e}")
                if dev > 3 * yard + 1e-7:
                    bad.append((key, dev, float(spread[key]), floor))
            report.append(f"step{step} {tag}: worst hip / yard-stick ratio {worst:.2f} (class floor {floor:.2e})")
    assert all(p.requires_grad for p in model.parameters())
    model.gen.eval()
    with torch.no_grad():
        y = model.gen(batch["bssfp"]["data"])[:, :, ::4, ::4, ::4].cpu().numpy()
    hip_dev = np.abs(y - g64["final/y_hat_eval_sample"]).mean()
    yard = float(spread["final/y_hat_eval_sample"])
    report.append(f"final eval output (mean abs deviation from f64): hip {hip_dev:.2e} cpu-ensemble {yard:.2e}")
    try:
        with open(os.path.join(os.path.dirname(golden_dir), "..", "gpurun_out", "f64_triangulation.log"), "w") as fh:
            fh.write("\n".join(report) + "\n")
    except OSError:
        pass
    assert not bad, bad
    assert hip_dev <= 3 * yard, (hip_dev, yard)


def test_gan_step_matches_oracle_n2_s64_with_torch_adamw(hip):
    """Same step driven with torch.optim.AdamW on both sides isolates the kernels from the optimiser.  Step 0 tightly;
    step 1 (after the sign-like first AdamW update) triangulated against an f64 run of the oracle made here:
    |hip - f64| <= 3 * max(largest |cpu_f32 - f64| among the losses, the committed ensemble spread of that loss) -- one
    CPU run is a single draw of the chaotic drift (see test_gan_training_step_golden)."""
    import copy
    import unet_bssfp_amd as M
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    torch.manual_seed(3)
    gen, discr = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
    rgen, rdiscr = R.RefGenerator("bssfp", dropout=0.0).train(), R.RefDiscriminator("bssfp").train()
    rgen.load_state_dict(gen.state_dict())
    rdiscr.load_state_dict(discr.state_dict())
    dgen, ddiscr = copy.deepcopy(rgen).double(), copy.deepcopy(rdiscr).double()
    model = bSSFPToDWITensorModel("bssfp", gen=gen.to(DEV), discr=discr.to(DEV), optimizer_class=torch.optim.AdamW).train()
    # 64^3 so that the last PatchGAN BatchNorm sees 16 values per channel (2 would be chaotic)
    batch = synthetic_batch(2, 64, seed=77, device=DEV)
    x, y = R.synthetic_batch(2, 64, seed=77)
    g_opt, d_opt = R.make_optimizers(rgen, rdiscr)
    g_opt64, d_opt64 = R.make_optimizers(dgen, ddiscr)
    spread = _gold(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"), "gan_step_f32_spread.npz")
    keys = ("gen_loss_adversarial", "gen_loss_recon_L1", "gen_loss_recon", "gen_loss", "discr_loss")
    for step in range(2):
        model.training_step(batch, step)
        ref = R.gan_training_step(rgen, rdiscr, g_opt, d_opt, x, y)
        r64 = R.gan_training_step(dgen, ddiscr, g_opt64, d_opt64, x.double(), y.double())
        cpu_dev = max(abs(float(ref[k]) - float(r64[k])) / abs(float(r64[k])) for k in keys)
        for k in keys:
            got = float(model.last_logs["train_" + k])
            if step == 0:
                np.testing.assert_allclose(got, float(ref[k]), rtol=1e-3, err_msg=f"{step}/{k}")
            dev = abs(got - float(r64[k])) / abs(float(r64[k]))
            assert dev <= 3 * max(cpu_dev, float(spread[f"step{step}/{k}"])) + 1e-7, (step, k, dev, cpu_dev)
    # BatchNorm buffers advanced identically: head BN twice per step, PatchGAN BN three times per step
    assert int(model.gen.blocks["bssfp"].bn.num_batches_tracked) == int(rgen.blocks["bssfp"].bn.num_batches_tracked) == 4
    assert int(model.discr.d2.bn.num_batches_tracked) == int(rdiscr.d2.bn.num_batches_tracked) == 6
    torch.testing.assert_close(model.discr.d3.bn.running_var.cpu(), rdiscr.d3.bn.running_var, rtol=3e-2, atol=1e-5)


def test_fused_adamw_matches_torch(hip):
    from unet_bssfp_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(32, 24, 3, 3, 3), (32,), (7,), (512, 256, 2, 2, 2)]
    ps = [torch.randn(s) for s in shapes]
    a = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    b = [p.clone().requires_grad_(True) for p in ps]
    oa, ob = FusedAdamW(a, lr=1e-3), torch.optim.AdamW(b, lr=1e-3)
    for step in range(3):
        for pa, pb in zip(a, b):
            g = torch.randn(pb.shape)
            pa.grad, pb.grad = g.to(DEV), g.clone()
        if step == 1:
            a[2].grad = None
            b[2].grad = None                              # a parameter without gradient is skipped
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa.detach().cpu(), pb.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_size_128_generator_properties(hip, dtype):
    """BASELINE.json full size (1x24x128^3): determinism, finite output, and the InstanceNorm
    invariance G(head(x)) is unchanged when the U-Net input is scaled... replaced by: bias-shift
    invariance -- adding a constant to a conv bias in front of an InstanceNorm cannot change y."""
    import unet_bssfp_amd as M
    torch.manual_seed(2)
    g = M.set_compute_dtype(M.Generator("bssfp", dropout=0.0).to(DEV).train(), dtype)
    x, _ = R.synthetic_batch(1, 128, seed=1234)
    xd = x.to(DEV)
    with torch.no_grad():
        y1 = g(xd)
        y2 = g(xd)
        assert torch.equal(y1, y2)                        # deterministic reductions
        assert torch.isfinite(y1).all() and y1.shape == (1, 6, 128, 128, 128)
        g.blocks["unet"].conv_0.conv_1.conv.bias.add_(0.5)
        y3 = g(xd)
    tol = 1e-4 if dtype == torch.float32 else 5e-2
    assert (y3 - y1).abs().mean().item() <= tol


def test_full_size_128_forward_parity_with_oracle(hip):
    """The north-star check at BASELINE.json's size: per-voxel L1 <= 1e-4 vs the CPU f32 reference."""
    import unet_bssfp_amd as M
    torch.manual_seed(0)
    g = M.Generator("bssfp", dropout=0.0)
    ref = R.RefGenerator("bssfp", dropout=0.0).train()
    ref.load_state_dict(g.state_dict())
    x, _ = R.synthetic_batch(1, 128, seed=1234)
    torch.set_num_threads(min(16, os.cpu_count() or 1))       # the GPU box grants a 16-core share
    with torch.no_grad():
        y_ref = ref(x)
        y = g.to(DEV).train()(x.to(DEV)).cpu()
    err = (y - y_ref).abs().mean().item()
    assert err <= 1e-4, err


def test_thesis_volume_96x128x128_forward_parity(hip):
    """The one I/O shape the reference documents for the whole generator (ONNX export rendered in
    doc/thesis/img/model.onnx.png: 1x24x96x128x128 -> 1x6x96x128x128, the CropOrPad size of
    src/data_module.py:124-127): non-cubic, every level ragged differently.  f32 parity with the oracle, bf16 runs."""
    import unet_bssfp_amd as M
    torch.manual_seed(0)
    g = M.Generator("bssfp", dropout=0.0)
    ref = R.RefGenerator("bssfp", dropout=0.0).train()
    ref.load_state_dict(g.state_dict())
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(1, 24, 96, 128, 128, generator=gen)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        y_ref = ref(x)
        g = g.to(DEV).train()
        y = g(x.to(DEV)).cpu()
        assert y.shape == (1, 6, 96, 128, 128)
        assert (y - y_ref).abs().mean().item() <= 1e-4
        yb = M.set_compute_dtype(g, torch.bfloat16)(x.to(DEV)).cpu()
    assert torch.isfinite(yb).all() and (yb - y_ref).abs().mean().item() <= 0.05 * max(y_ref.abs().mean().item(), 0.1)


def test_config5_size_160_forward_parity_and_step(hip):
    """BASELINE.json configs[4] size (1x24x160^3: bottleneck 10^3, ragged 40/20/10-wide levels): f32 forward
    parity with the CPU oracle (per-voxel L1 <= 1e-4), then one full bf16 GAN step (finite losses, deterministic
    forward).  The e4m3 arithmetic that configuration names is exercised at this size by
    tests/test_gpu_fp8.py::test_fp8_gan_step_at_config5_size_160."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd import gan
    torch.manual_seed(0)
    g = M.Generator("bssfp", dropout=0.0)
    ref = R.RefGenerator("bssfp", dropout=0.0).train()
    ref.load_state_dict(g.state_dict())
    x, y = R.synthetic_batch(1, 160, seed=77)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        y_ref = ref(x)
        out = g.to(DEV).train()(x.to(DEV)).cpu()
    assert out.shape == (1, 6, 160, 160, 160)
    assert (out - y_ref).abs().mean().item() <= 1e-4
    del g, ref, y_ref, out
    torch.manual_seed(1)
    model = gan.bSSFPToDWITensorModel("bssfp", batch_size=1).to(DEV)
    M.set_compute_dtype(model.gen, torch.bfloat16), M.set_compute_dtype(model.discr, torch.bfloat16)
    batch = {"bssfp": {"data": x.to(DEV)}, "dwi-tensor_orig": {"data": y.to(DEV)}}
    model.training_step(batch)
    logs = {k: float(v) for k, v in model.last_logs.items()}
    assert all(np.isfinite(v) for v in logs.values()), logs
    assert 0.2 < logs["train_gen_loss_recon_L1"] < 1.0 and 0.3 < logs["train_discr_loss"] < 2.0, logs


def test_hipgraph_replayed_step_equals_eager_step(hip):
    """GraphedTrainingStep (whole step as one hipGraph) must produce bit-identical parameters to the
    eager training_step (same kernels, same order), and keep AdamW's bias correction and the dropout
    counter advancing on the device."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch

    def build():
        torch.manual_seed(4)
        DropoutState.reset()
        gen, discr = M.Generator("bssfp", dropout=0.05), M.Discriminator("bssfp")
        return bSSFPToDWITensorModel("bssfp", gen=gen.to(DEV), discr=discr.to(DEV)).train()

    batch = synthetic_batch(2, 32, seed=9, device=DEV)
    eager = build()
    for i in range(4):
        eager.training_step(batch, i)
    graphed = build()
    gs = GraphedTrainingStep(graphed, batch, warmup=2)         # 2 eager warm-up steps, then capture (records, does not run)
    gs()                                                       # replay = step 3
    gs()                                                       # replay = step 4
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert torch.equal(p, q), n
    for k in ("train_gen_loss", "train_discr_loss"):
        assert float(eager.last_logs[k]) == float(graphed.last_logs[k])
    # the multi-rank structure (five segments, two-stage backward passes cut at the marked activations) on one rank
    seg = build()
    gs2 = GraphedTrainingStep(seg, batch, warmup=2, force_segments=True)
    assert len(gs2.graphs) == 5
    gs2()
    gs2()
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(eager.named_parameters(), seg.named_parameters()):
        assert torch.equal(p, q), n
    g_opt, _ = graphed.optimizers()
    g_opt.sync_step_counts()
    assert g_opt.state[graphed.gen.blocks["unet"].final_conv.weight]["step"] == 4


def test_hipgraph_two_input_sets_equal_eager_steps_on_alternating_batches(hip):
    """GraphedTrainingStep.add_instance: the step captured over a second static input set (the feed of
    src/data_module.py:185-188 -- a new batch every step -- alternates between the two sets instead of copying into one):
    replays 0, 1, 0, 1 over batches A, B must equal eager training steps on A, B, A, B bit for bit."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch

    def build():
        torch.manual_seed(4)
        DropoutState.reset()
        gen, discr = M.Generator("bssfp", dropout=0.05), M.Discriminator("bssfp")
        return bSSFPToDWITensorModel("bssfp", gen=gen.to(DEV), discr=discr.to(DEV)).train()

    a, b = synthetic_batch(2, 32, seed=9, device=DEV), synthetic_batch(2, 32, seed=10, device=DEV)
    c = synthetic_batch(2, 32, seed=11, device=DEV)
    # the eager model runs first and is snapshotted: build() resets the process-wide dropout counter, which a model that is
    # still stepping must not lose
    eager = build()
    snaps, elogs = [], []
    for i, batch in enumerate((a, a, a, b, a, b, c)):         # two warm-up steps on A (as the graphed model), then A B A B, then C
        eager.training_step(batch, i)
        if i in (5, 6):
            torch.cuda.synchronize()
            snaps.append([p.detach().clone() for p in eager.parameters()])
            elogs.append({k: float(v) for k, v in eager.last_logs.items()})
    graphed = build()
    gs = GraphedTrainingStep(graphed, a, warmup=2)
    assert gs.add_instance(b) == 1
    for k in (0, 1, 0, 1):
        gs(k)
    torch.cuda.synchronize()
    for (n, q), p in zip(graphed.named_parameters(), snaps[0]):
        assert torch.equal(p, q), n
    # every instance logs into its own tensors (the second capture used to overwrite the first one's dict entries):
    # after a replay of instance 1 last_logs holds the losses of THAT step (batch B), after instance 0 those of its step
    assert gs.instances[0][2] is not gs.instances[1][2] and graphed.last_logs is gs.instances[1][2]
    assert {k: float(v) for k, v in graphed.last_logs.items()} == elogs[0]
    # new data written into an input set between replays is what the next replay of that set trains on
    for key in ("bssfp", "dwi-tensor_orig"):
        a[key]["data"].copy_(c[key]["data"])
    gs(0)
    torch.cuda.synchronize()
    for (n, q), p in zip(graphed.named_parameters(), snaps[1]):
        assert torch.equal(p, q), n
    assert graphed.last_logs is gs.instances[0][2]
    assert {k: float(v) for k, v in graphed.last_logs.items()} == elogs[1]


def test_discriminator_first_block_split_equals_the_unsplit_network(hip):
    """bf16 mode: the PatchGAN with its first block split into x-part (once per step) + y-part (Fn.SplitS2dConvFn; the default
    where the marching k2 kernel applies) against the same network with the block unsplit: the generator-phase use (D frozen,
    gradient w.r.t. y) and the discriminator-phase use (stacked pair, parameter gradients).  The first block's own outputs
    differ by f32 summation order only; downstream the BatchNorms over 8 - 16 values per channel amplify every rounding
    difference (bounds of test_discriminator_forward_pair_equals_two_calls)."""
    import copy
    import unet_bssfp_amd as M
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd import ops
    torch.manual_seed(8)
    state = copy.deepcopy(M.Discriminator("bssfp").state_dict())
    g = torch.Generator().manual_seed(9)
    x = torch.rand(1, 24, 64, 64, 64, generator=g).to(DEV)
    ya = torch.rand(1, 6, 64, 64, 64, generator=g).to(DEV)
    yb = torch.rand(1, 6, 64, 64, 64, generator=g).to(DEV)
    wts = torch.rand(2, 1, 2, 2, 2, generator=g).to(DEV) - 0.5
    res = {}
    for split in (True, False):
        d = M.Discriminator("bssfp")
        d.load_state_dict(state)
        d = M.set_compute_dtype(d.to(DEV).train(), torch.bfloat16)
        d.split_first_block = split
        Fn.DropoutState.advance(torch.device(DEV))                  # a new step: the per-step memos are empty
        launches = []
        ops.CONV_PROBE = lambda pid, dd, real: launches.append((pid, bool(dd.addend)))
        try:
            for p in d.parameters():
                p.requires_grad_(False)
            yg = ya.clone().requires_grad_(True)
            la = d(x, yg)
            (la * wts[:1]).sum().backward()
            for p in d.parameters():
                p.requires_grad_(True)
            both = d.forward_pair(x, ya, yb, stacked=True)
            (both * wts).sum().backward()
        finally:
            ops.CONV_PROBE = None
        torch.cuda.synchronize()
        n_add = sum(1 for _, a in launches if a)
        assert n_add == (2 if split else 0), launches              # one y-part launch per use; the x-part ran once (memo)
        if split:
            assert sum(1 for pid, a in launches if (pid % 10000) // 100 == 24 and not a) >= 1
        res[split] = (la.detach().float().cpu(), yg.grad.cpu(), both.detach().float().cpu(),
                      {n: p.grad.detach().cpu() for n, p in d.named_parameters() if p.grad is not None})
    a, b = res[True], res[False]
    torch.testing.assert_close(a[0], b[0], rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(a[2], b[2], rtol=2e-2, atol=2e-3)
    rel = float((a[1] - b[1]).norm() / b[1].norm())
    assert rel <= 0.05, rel
    assert set(a[3]) == set(b[3])
    for n, ga in a[3].items():
        gb = b[3][n]
        if float(gb.abs().max()) == 0.0:
            assert float(ga.abs().max()) == 0.0, n
            continue
        rel = float((ga - gb).norm() / gb.norm().clamp_min(1e-30))
        cos = float(torch.nn.functional.cosine_similarity(ga.flatten().double(), gb.flatten().double(), dim=0))
        assert rel <= 0.3 and cos >= 0.95, (n, rel, cos)


def test_discriminator_forward_pair_in_eval_mode_equals_two_calls_and_the_oracle(hip):
    """on a model in .eval() BatchNorm normalises with its running statistics -- ONE (1, c) mean / rstd row whatever
    the stacked pass's bn_groups says (the kernel read mean[g * c + ch] for g = 1: out of bounds).  forward_pair in eval mode
    == two calls (same kernels, same statistics) == the oracle's Discriminator in eval mode within the f32 bound;
    the running statistics are loaded with non-trivial values so that a wrong row cannot hide behind mean 0 / var 1."""
    import unet_bssfp_amd as M
    from oracle import unet_ref as R
    torch.manual_seed(4)
    d = M.Discriminator("bssfp")
    g = torch.Generator().manual_seed(6)
    sd = d.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
    d.load_state_dict(sd)
    ref = R.RefDiscriminator("bssfp")
    ref.load_state_dict(sd)
    ref.eval()
    x = torch.rand(2, 24, 64, 64, 64, generator=g)
    ya = torch.rand(2, 6, 64, 64, 64, generator=g)
    yb = torch.rand(2, 6, 64, 64, 64, generator=g)
    d = d.to(DEV).eval()
    with torch.no_grad():
        la, lb = d.forward_pair(x.to(DEV), ya.to(DEV), yb.to(DEV))
        ta, tb = d(x.to(DEV), ya.to(DEV)), d(x.to(DEV), yb.to(DEV))
        ra, rb = ref(x, ya), ref(x, yb)
    torch.cuda.synchronize()
    # (same kernels per sample; the split-K plan of the low levels may differ between 4 and 2 stacked samples: summation order)
    torch.testing.assert_close(la, ta, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(lb, tb, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(la.cpu(), ra, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(lb.cpu(), rb, rtol=1e-3, atol=1e-4)
    for k, v in d.state_dict().items():                       # eval mode: no buffer moves
        assert torch.equal(v.cpu(), sd[k]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_discriminator_forward_pair_equals_two_calls(hip, dtype):
    """Discriminator.forward_pair(x, y_a, y_b) -- the discriminator phase's two calls (src/model.py:184-186) as one pass over
    the stacked inputs -- against the two separate calls: logits, BatchNorm running statistics and counters after the pair
    (each half normalised with its own batch statistics, running statistics updated in call order), and the parameter
    gradients of (BCE(fake, 0) + BCE(real, 1)) / 2.  Same kernels per sample; only summation orders differ."""
    import copy
    import unet_bssfp_amd as M
    torch.manual_seed(3)
    d0 = M.Discriminator("bssfp")
    state = copy.deepcopy(d0.state_dict())
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 24, 64, 64, 64, generator=g).to(DEV)
    ya = torch.rand(2, 6, 64, 64, 64, generator=g).to(DEV)
    yb = torch.rand(2, 6, 64, 64, 64, generator=g).to(DEV)
    res = {}
    for mode in ("pair", "two"):
        d = M.Discriminator("bssfp")
        d.load_state_dict(state)
        d = M.set_compute_dtype(d.to(DEV).train(), dtype)
        if mode == "pair":
            la, lb = d.forward_pair(x, ya, yb)
        else:
            la, lb = d(x, ya), d(x, yb)
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        loss = (bce(la, torch.zeros_like(la)) + bce(lb, torch.ones_like(lb))) / 2
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (la.detach().float().cpu(), lb.detach().float().cpu(),
                     {n: p.grad.detach().cpu() for n, p in d.named_parameters() if p.grad is not None},
                     {n: b.detach().clone().cpu() for n, b in d.named_buffers()})
    # bf16: the stacked pass and the two calls take different plans from d2 on (segment lengths, split-K factors: other f32
    # summation orders, hence bf16 ulp flips in the activations), and the last BatchNorms normalise 8 - 16 values per channel:
    # a flipped ulp (2^-8 relative) moves a logit -- a sum over 512 such channels -- by a few 1e-3
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=6e-3)
    torch.testing.assert_close(res["pair"][0], res["two"][0], **tol)
    torch.testing.assert_close(res["pair"][1], res["two"][1], **tol)
    assert set(res["pair"][2]) == set(res["two"][2])
    # gradients: the PatchGAN's BatchNorms see 8 - 16 values per channel at this size, which amplifies every rounding
    # difference (two f32 evaluations in different summation orders differ by ~1e-3 of the largest element; bf16 storage
    # moves these gradients by 20 - 45 % rel-L2 against f32): direction and size must agree
    for n, gp in res["pair"][2].items():
        gt = res["two"][2][n]
        if float(gt.abs().max()) == 0.0:                      # conv bias in front of a BatchNorm: exact zeros on both sides
            assert float(gp.abs().max()) == 0.0, n
            continue
        rel = float((gp - gt).norm() / gt.norm().clamp_min(1e-30))
        cos = float(torch.nn.functional.cosine_similarity(gp.flatten().double(), gt.flatten().double(), dim=0))
        if dtype == torch.float32:
            assert rel <= 5e-3 and cos >= 0.9999, (n, rel, cos)
        else:
            assert rel <= 0.3 and cos >= 0.95, (n, rel, cos)
    for n, bp in res["pair"][3].items():
        bt = res["two"][3][n]
        if bp.dtype == torch.long:
            assert int(bp) == int(bt) == 2, n                  # two forward calls were counted
        else:
            torch.testing.assert_close(bp, bt, **(dict(rtol=1e-3, atol=5e-6) if dtype == torch.float32 else dict(rtol=5e-2, atol=1e-3)))


def test_side_stream_weight_gradients_equal_main_stream(hip):
    """functional.SideStream (weight gradients of the small layers on a second stream; off by default: measured slower) must
    not change a single bit, eagerly and under hipGraph capture (fork / join edges inside the captured step)."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch
    batch = synthetic_batch(2, 32, seed=9, device=DEV)
    out = {}
    try:
        for mode in ("off", "on_eager", "on_graph"):
            Fn.SideStream.allowed = mode != "off"
            torch.manual_seed(4)
            DropoutState.reset()
            model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.05).to(DEV), discr=M.Discriminator("bssfp").to(DEV)).train()
            if mode == "on_graph":
                gs = GraphedTrainingStep(model, batch, warmup=2)
                gs(); gs()
            else:
                for i in range(4):
                    model.training_step(batch, i)
            torch.cuda.synchronize()
            out[mode] = [p.detach().clone() for p in model.parameters()]
    finally:
        Fn.SideStream.allowed = False
    for mode in ("on_eager", "on_graph"):
        assert all(torch.equal(a, b) for a, b in zip(out["off"], out[mode])), mode


@pytest.mark.parametrize("pair", [True, False])
def test_deferred_weight_gradient_reduction_is_bit_identical(hip, pair):
    """functional.DeferredReduce (mi355_conv_wgrad_partial + ONE mi355_wgrad_reduce_multi launch per 16 layers at the end of a
    backward pass) against every weight-gradient launch followed by its own reduction: the same kernel bodies and summation
    order per layer, so every parameter after four steps is bit-identical, eagerly and under hipGraph replay.  pair = False: the
    discriminator phase is two calls (as for extents that are not multiples of 32), every discriminator weight receives two
    contributions and the second one (accumulating) flushes the pending reductions first."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch
    batch = synthetic_batch(2, 32, seed=9, device=DEV)
    out = {}
    try:
        for mode in ("immediate", "deferred_eager", "deferred_graph"):
            Fn.DeferredReduce.allowed = mode != "immediate"
            n0 = Fn.DeferredReduce.launches
            torch.manual_seed(4)
            DropoutState.reset()
            model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.05).to(DEV), discr=M.Discriminator("bssfp").to(DEV)).train()
            model.pair_discriminator_calls = pair
            if mode == "deferred_graph":
                gs = GraphedTrainingStep(model, batch, warmup=2)
                gs(); gs()
            else:
                for i in range(4):
                    model.training_step(batch, i)
            torch.cuda.synchronize()
            assert (Fn.DeferredReduce.launches > n0) == (mode != "immediate")
            assert not Fn.DeferredReduce._jobs and not Fn.DeferredReduce._after and not Fn.DeferredReduce.enabled
            out[mode] = [p.detach().clone() for p in model.parameters()]
    finally:
        Fn.DeferredReduce.allowed = True
    for mode in ("deferred_eager", "deferred_graph"):
        assert all(torch.equal(a, b) for a, b in zip(out["immediate"], out[mode])), mode


def test_wgrad_reduce_multi_matches_single_launches(hip):
    """mi355_wgrad_reduce_multi over a mixed list of jobs (marching 3x3x3, tile-form 3x3x3 with the dense reduce, 1x1x1 streaming,
    transposed-conv classes, k2 on a space-to-depth operand; more than 16 jobs: two launches) == mi355_conv_wgrad per layer,
    bit for bit; accumulate = 1 jobs add to what dw held."""
    from unet_bssfp_amd import ops
    torch.manual_seed(0)
    bf = torch.bfloat16

    def act(n, d, c):
        return ops.as_act(torch.randn(n, d, d, d, c, device=DEV).to(bf))
    cases = []
    for (d, ci, co) in ((32, 32, 32), (32, 64, 32), (16, 128, 64), (8, 256, 256), (16, 64, 128)):      # 3x3x3 pad 1
        cases.append(dict(x=act(1, d, ci), g=act(1, d, co), grid=(d, d, d), ks=3, pad=(1, 1, 1), cout=co, cin=ci,
                          geo=(ci * 27, 27, (9, 3, 1), (0, 0, 0), (1, 1, 1)), shape=(co, ci, 3, 3, 3), kw={}))
    for (d, ci, co) in ((64, 32, 16), (16, 64, 32)):                                                  # 1x1x1
        cases.append(dict(x=act(1, d, ci), g=act(1, d, co), grid=(d, d, d), ks=1, pad=(0, 0, 0), cout=co, cin=ci,
                          geo=(ci, 1, (1, 1, 1), (0, 0, 0), (1, 1, 1)), shape=(co, ci, 1, 1, 1), kw={}))
    for (d, ci, co) in ((16, 64, 64), (8, 128, 128)):                                                 # ConvTranspose3d(k2, s2) classes
        cases.append(dict(x=act(1, d, ci), g=act(1, 2 * d, co), grid=(d, d, d), ks=1, pad=(0, 0, 0), cout=co, cin=ci,
                          geo=(8, co * 8, (4, 2, 1), (0, 0, 0), (0, 0, 0)), shape=(ci, co, 2, 2, 2), kw=dict(g_cls_cout=co)))
    cases = cases * 2                                                                                  # 18 jobs
    outs = []
    for mode in ("single", "multi"):
        dws, jobs = [], []
        for i, c in enumerate(cases):
            acc = i >= len(cases) // 2
            dw = torch.full(c["shape"], 0.5 if acc else float("nan"), dtype=torch.float32, device=DEV)
            ops.conv_wgrad(c["x"], None, c["g"], c["grid"], 1, (0, 0, 0), c["ks"], 1, c["pad"], dw, c["cout"], c["cin"], *c["geo"],
                           accumulate=acc, defer=jobs if mode == "multi" else None, **c["kw"])
            dws.append(dw)
        if mode == "multi":
            assert len(jobs) == len(cases)
            ops.wgrad_reduce_multi(jobs)
        torch.cuda.synchronize()
        outs.append(dws)
    for a, b in zip(*outs):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)


def test_gradient_sinks_equal_autograd_accumulation(hip):
    """gradsink.GradBuckets (gradient kernels write parameter gradients in place, second uses accumulate in the kernel) must
    give the same gradients and the same parameters after two steps as plain autograd accumulation (.grad tensors created by
    AccumulateGrad, duplicate uses summed by the engine): same kernels, so bit-identical except where the discriminator's
    two contributions are added in a different order (in-kernel accumulate vs engine add: commutative, still identical)."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    batch = synthetic_batch(2, 32, seed=3, device=DEV)
    out = []
    for sinks in (True, False):
        torch.manual_seed(6)
        DropoutState.reset()
        model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.05).to(DEV), discr=M.Discriminator("bssfp").to(DEV)).train()
        model.use_grad_sinks = sinks
        for i in range(2):
            model.training_step(batch, i)
        assert (model.sinks_gen is not None) == sinks
        out.append(([p.detach().clone() for p in model.parameters()], {k: float(v) for k, v in model.last_logs.items()}))
        if sinks:
            assert model.sinks_gen.complete() and model.sinks_discr.complete()
            # every used parameter's .grad is a view of its bucket
            flat = model.sinks_gen.flat[0]
            p0 = model.sinks_gen.params[0][0]
            assert flat.data_ptr() <= p0.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a, b)
    assert out[0][1] == out[1][1]


def test_gan_loss_heads_one_launch_each_way_match_torch(hip):
    """functional.GanGenLossFn / GanDiscrLossFn (the loss heads of src/model.py:126-137 and :183-193 as single launches)
    against the composed torch form on the same device tensors: values to 1e-6 relative, gradients to 1e-6 of their scale;
    the stacked form of the discriminator loss (one logits tensor from Discriminator.forward_pair) equals the two-tensor form
    bit for bit."""
    import torch.nn.functional as F
    from unet_bssfp_amd.functional import GanDiscrLossFn, GanGenLossFn
    g = torch.Generator().manual_seed(31)
    logits = (torch.randn(2, 1, 4, 4, 4, generator=g) * 3).to(DEV).requires_grad_(True)
    y_hat = torch.rand(2, 6, 16, 24, 32, generator=g).to(DEV).requires_grad_(True)
    y = torch.rand(2, 6, 16, 24, 32, generator=g).to(DEV)
    total, parts = GanGenLossFn.apply(logits, y_hat, y, 1.0, 100.0)
    (total * 0.7).backward()
    l2, yh2 = logits.detach().clone().requires_grad_(True), y_hat.detach().clone().requires_grad_(True)
    adv = F.binary_cross_entropy_with_logits(l2, torch.ones_like(l2))
    l1 = F.l1_loss(yh2, y)
    recon = l1 / 1 * 100.0
    ((adv + recon) * 0.7).backward()
    for got, ref in zip(parts.tolist(), (float(l1), float(recon), float(adv), float(adv + recon))):
        assert abs(got - ref) <= 1e-6 * abs(ref) + 1e-7, (got, ref)
    assert float(total) == float(parts[3])
    assert (logits.grad - l2.grad).abs().max() <= 1e-6 * l2.grad.abs().max()
    assert (y_hat.grad - yh2.grad).abs().max() <= 1e-6 * yh2.grad.abs().max()
    fake = (torch.randn(2, 1, 4, 4, 4, generator=g) * 2).to(DEV).requires_grad_(True)
    real = (torch.randn(2, 1, 4, 4, 4, generator=g) * 2).to(DEV).requires_grad_(True)
    loss = GanDiscrLossFn.apply(fake, real)
    loss.backward()
    f2, r2 = fake.detach().clone().requires_grad_(True), real.detach().clone().requires_grad_(True)
    ref = (F.binary_cross_entropy_with_logits(r2, torch.ones_like(r2)) + F.binary_cross_entropy_with_logits(f2, torch.zeros_like(f2))) / 2
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    assert (fake.grad - f2.grad).abs().max() <= 1e-6 * f2.grad.abs().max()
    assert (real.grad - r2.grad).abs().max() <= 1e-6 * r2.grad.abs().max()
    both = torch.cat([fake.detach(), real.detach()]).requires_grad_(True)
    loss_s = GanDiscrLossFn.apply(both, None)
    loss_s.backward()
    assert float(loss_s) == float(loss)
    assert torch.equal(both.grad, torch.cat([fake.grad, real.grad]))


def test_training_step_with_fused_loss_heads_tracks_the_composed_form(hip):
    """One GAN training step with the loss heads as single launches (default on the HIP path) against the same step with
    the composed torch losses (``fused_loss_heads = False``): the logged losses agree to 1e-6 relative and the parameters
    after the step to the f32 parity tolerance of the step (identical kernels everywhere else; the loss gradients differ in
    the last bit)."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    batch = synthetic_batch(2, 32, seed=3, device=DEV)
    res = {}
    for fused in (True, False):
        torch.manual_seed(12)
        DropoutState.reset()
        model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp", dropout=0.0).to(DEV), discr=M.Discriminator("bssfp").to(DEV)).train()
        model.fused_loss_heads = fused
        model.training_step(batch, 0)
        torch.cuda.synchronize()
        res[fused] = ({k: float(v) for k, v in model.last_logs.items()}, [p.detach().clone() for p in model.parameters()])
    assert list(res[True][0]) == list(res[False][0])                      # same keys, same order
    for k, v in res[False][0].items():
        assert abs(res[True][0][k] - v) <= 1e-6 * abs(v) + 1e-7, (k, res[True][0][k], v)
    # AdamW's first step moves every element by ~lr whatever the gradient's size: compare where the gradient sign is robust
    moved = 0
    for p, q in zip(res[True][1], res[False][1]):
        close = (p - q).abs() <= 1e-4 * q.abs() + 1e-6
        moved += int(close.sum())
        assert close.float().mean() > 0.98, float(close.float().mean())
    assert moved > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transposed_conv_bias_gradient_from_the_producing_launch(hip, dtype):
    """The bias gradient of the transposed convolution under a skip concatenation (MONAI UpCat, src/model.py:22-28) is the
    per-channel sum of the data gradient that the following convolution's backward writes for its second source; that
    launch emits the sums through its fused-statistics epilogue (functional.ColSumSide) instead of a separate pass over
    the gradient.  Against the separate pass: same tensor summed in another order (f32 partial sums, f64 combination)."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd.nn import UpCat
    torch.manual_seed(5)
    blk = M.set_compute_dtype(UpCat(64, 32, 32, 0.0).to(DEV).train(), dtype)
    g = torch.Generator().manual_seed(6)
    from tests.test_gpu_ops import to_act
    x = to_act(torch.randn(1, 64, 8, 16, 32, generator=g), dtype)
    xe = to_act(torch.randn(1, 32, 16, 32, 64, generator=g), dtype)
    gy = to_act(torch.randn(1, 32, 16, 32, 64, generator=g), dtype)
    grads = {}
    for carried in (True, False):
        Fn.ColSumSide.enabled = carried
        try:
            blk.zero_grad()
            xin = x.clone().requires_grad_(True)
            y = blk.forward_act(xin, xe)
            y.backward(gy)
            grads[carried] = {n: p.grad.detach().clone() for n, p in blk.named_parameters()}
        finally:
            Fn.ColSumSide.enabled = True
    a, b = grads[True]["upsample.deconv.bias"], grads[False]["upsample.deconv.bias"]
    assert not torch.equal(a, torch.zeros_like(a))
    # bf16: the separate pass sums the bf16-ROUNDED gradient, the carried sums come from the f32 accumulators before rounding:
    # they differ by the accumulated rounding noise, ~ sqrt(32768 voxels) * 2^-9 * rms (0.45 of a maximum of 61 measured)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (a - b).abs().max() <= tol * b.abs().max() + 1e-6, ((a - b).abs().max(), b.abs().max())
    for n in grads[True]:
        if n != "upsample.deconv.bias":
            assert torch.equal(grads[True][n], grads[False][n]), n


def test_gan_step_at_the_reference_batch_8x64_matches_oracle(hip):
    """The reference's REAL training shape (src/data_module.py:12, 18, 185-188): 8 patches of 64^3 per step -- the same voxel
    count as one 128^3 volume, but every BatchNorm (generator head, PatchGAN) sees N = 8 and the discriminator phase stacks
    16 samples through ``forward_pair``.  f32 parity mode against the CPU oracle: generator output per-voxel L1 <= 1e-4
    (north_star's bound) and the step-0 losses within rtol 1e-3 (the discriminator loss, which already sees the generator's
    first AdamW update: 2e-3, the bound of smoke())."""
    import unet_bssfp_amd as M
    from oracle import unet_ref as R
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    torch.manual_seed(21)
    gen, discr = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
    rgen, rdiscr = R.RefGenerator("bssfp", dropout=0.0).train(), R.RefDiscriminator("bssfp").train()
    rgen.load_state_dict(gen.state_dict())
    rdiscr.load_state_dict(discr.state_dict())
    x, y = R.synthetic_batch(8, 64, seed=4321)
    batch = synthetic_batch(8, 64, seed=4321, device=DEV)
    assert torch.equal(batch["bssfp"]["data"].cpu(), x)
    gen, discr = gen.to(DEV).train(), discr.to(DEV).train()
    with torch.no_grad():                      # (train mode on both sides: the head's BatchNorm uses the statistics of the 8 patches)
        y_hip = gen(batch["bssfp"]["data"]).cpu()
        y_ref = rgen(x)
    err = float((y_hip - y_ref).abs().mean())
    assert err <= 1e-4, err
    model = bSSFPToDWITensorModel("bssfp", gen=gen, discr=discr).train()
    assert model._discr_uses(batch["bssfp"]["data"], batch["dwi-tensor_orig"]["data"]) == 1      # the stacked pass is what runs
    model.training_step(batch, 0)
    torch.cuda.synchronize()
    g_opt, d_opt = R.make_optimizers(rgen, rdiscr)
    ref = R.gan_training_step(rgen, rdiscr, g_opt, d_opt, x, y)
    for k in ("gen_loss_adversarial", "gen_loss_recon_L1", "gen_loss_recon", "gen_loss", "discr_loss"):
        a, b = float(model.last_logs["train_" + k]), float(ref[k])
        tol = 2e-3 if k == "discr_loss" else 1e-3
        assert abs(a - b) <= tol * max(1.0, abs(b)), (k, a, b)
    # BatchNorm running statistics after the step: generator head (2 forward passes of 8 patches), PatchGAN (3 passes)
    for (n, b_hip), (_, b_ref) in zip(sorted(dict(gen.named_buffers()).items()), sorted(dict(rgen.named_buffers()).items())):
        if b_hip.dtype.is_floating_point and "bssfp" in n and "pc-" not in n:
            torch.testing.assert_close(b_hip.cpu(), b_ref, rtol=2e-3, atol=2e-5, msg=n)
