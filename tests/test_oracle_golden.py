"""CPU: the oracle restatement reproduces the vectors produced by the reference's own
classes (tests/golden/*.npz, written by oracle/gen_golden.py from src/model.py:15-92)."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _digest(module):
    out = {}
    for n, p in module.named_parameters():
        if p.grad is None:
            out[n] = None
        else:
            g = p.grad.double()
            out[n] = np.array([g.sum().item(), g.abs().sum().item()])
    return out


def _check_digest(gold, prefix, module, rtol=1e-4, atol=1e-6):
    for n, v in _digest(module).items():
        ref = gold[f"{prefix}/{n}"]
        if v is None:
            assert np.isnan(ref).all(), n
        else:
            np.testing.assert_allclose(v, ref, rtol=rtol, atol=atol, err_msg=n)


DS_CFGS = {
    "k1_bn_act": dict(in_channels=24, out_channels=24, kernel=1, strides=1, padding=0),
    "k4s2_bn_act": dict(in_channels=32, out_channels=64),
    "k4s2_nobn": dict(in_channels=30, out_channels=32, batchnorm=False),
    "k4s2_noact": dict(in_channels=16, out_channels=32, activation=False),
}


@pytest.mark.parametrize("idx,name", list(enumerate(DS_CFGS)))
def test_downsample_conv_matches_reference(golden_dir, idx, name):
    gold = _load(golden_dir, "downsample_conv.npz")
    kw = DS_CFGS[name]
    torch.manual_seed(100 + idx)
    m = R.RefDownSampleConv(**kw).train()
    g = torch.Generator().manual_seed(200 + idx)
    x = torch.rand(2, kw["in_channels"], 16, 16, 16, generator=g, requires_grad=True)
    y = m(x)
    w = torch.rand(y.shape, generator=g)
    (y * w).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), gold[f"{name}/y"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), gold[f"{name}/dx"], rtol=1e-4, atol=1e-6)
    _check_digest(gold, f"{name}/grad", m)
    if kw.get("batchnorm", True):
        np.testing.assert_allclose(m.bn.running_mean.numpy(), gold[f"{name}/running_mean"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m.bn.running_var.numpy(), gold[f"{name}/running_var"], rtol=1e-5, atol=1e-7)
        assert int(m.bn.num_batches_tracked) == 1
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x.detach()).numpy(), gold[f"{name}/y_eval"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,modality,n,s,cin", [("bssfp_n1_s64", "bssfp", 1, 64, 24),
                                                   ("t1w_n2_s32", "t1w", 2, 32, 6)])
def test_discriminator_matches_reference(golden_dir, tag, modality, n, s, cin):
    gold = _load(golden_dir, "discriminator.npz")
    torch.manual_seed(7)
    d = R.RefDiscriminator(modality).train()
    assert sorted(d.state_dict().keys()) == list(gold[f"{tag}/keys"])
    assert sum(p.numel() for p in d.parameters()) == int(gold[f"{tag}/nparams"])
    x, y = R.synthetic_batch(n, s, seed=1234, cin=cin)
    y.requires_grad_(True)
    logits = d(x, y)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    loss.backward()
    np.testing.assert_allclose(logits.detach().numpy(), gold[f"{tag}/logits"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y.grad[:, :, ::7, ::5, ::3].numpy(), gold[f"{tag}/dy_sample"], rtol=1e-4, atol=1e-9)
    _check_digest(gold, f"{tag}/grad", d)


def test_discriminator_param_count_bssfp():
    # 11 230 593 unique parameters
    assert sum(p.numel() for p in R.RefDiscriminator("bssfp").parameters()) == 11_230_593


@pytest.mark.parametrize("tag,modality,cin", [("bssfp", "bssfp", 24), ("dwi", "dwi-tensor", 6)])
def test_generator_wiring_matches_reference(golden_dir, tag, modality, cin):
    gold = _load(golden_dir, "generator.npz")
    torch.manual_seed(11)
    g = R.RefGenerator(modality, dropout=0.0).train()
    assert sorted(g.state_dict().keys()) == list(gold[f"{tag}/keys"])
    assert sum(p.numel() for p in g.parameters()) == int(gold[f"{tag}/nparams"]) == 22_646_182
    x, y = R.synthetic_batch(1, 32, seed=4321, cin=cin)
    x.requires_grad_(True)
    y_hat = g(x)
    loss = torch.nn.functional.l1_loss(y_hat, y)
    loss.backward()
    np.testing.assert_allclose(y_hat.detach().numpy(), gold[f"{tag}/y_hat"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x.grad[:, :, ::5, ::3, ::2].numpy(), gold[f"{tag}/dx_sample"], rtol=1e-3, atol=1e-8)
    _check_digest(gold, f"{tag}/grad", g, rtol=1e-3, atol=1e-6)
    g.eval()
    with torch.no_grad():
        np.testing.assert_allclose(g(x.detach()).numpy(), gold[f"{tag}/y_hat_eval"], rtol=1e-4, atol=1e-5)


def test_basic_unet_shapes_and_keys():
    u = R.RefBasicUNet(3, 24, 6, R.UNET_FEATURES, dropout=0.05)
    assert sum(p.numel() for p in u.parameters()) == 22_645_318
    keys = set(u.state_dict().keys())
    for k in ("conv_0.conv_0.conv.weight", "conv_0.conv_1.adn.N.bias", "down_4.convs.conv_1.conv.bias",
              "upcat_4.upsample.deconv.weight", "upcat_1.convs.conv_0.adn.N.weight", "final_conv.bias"):
        assert k in keys, k
    assert u.upcat_1.convs.conv_0.conv.weight.shape == (32, 96, 3, 3, 3)
    assert u.upcat_4.upsample.deconv.weight.shape == (512, 256, 2, 2, 2)
    with torch.no_grad():
        assert u.eval()(torch.rand(1, 24, 32, 32, 32)).shape == (1, 6, 32, 32, 32)


def test_basic_unet_2d_plumbing_config():
    # BASELINE.json configs[0]: 2D U-Net 1->6ch on 64x64 slices, CPU plumbing
    u = R.RefBasicUNet(2, 1, 6, R.UNET_FEATURES, dropout=0.0).train()
    x = torch.rand(2, 1, 64, 64, requires_grad=True)
    y = u(x)
    assert y.shape == (2, 6, 64, 64)
    y.abs().mean().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_gan_step_matches_reference(golden_dir):
    gold = _load(golden_dir, "gan_step.npz")
    torch.manual_seed(0)
    gen = R.RefGenerator("bssfp", dropout=0.0).train()
    discr = R.RefDiscriminator("bssfp").train()
    g_opt, d_opt = R.make_optimizers(gen, discr)
    x, y = R.synthetic_batch(1, 64, seed=1234)
    for step in range(2):
        logs = R.gan_training_step(gen, discr, g_opt, d_opt, x, y)
        for k, v in logs.items():
            np.testing.assert_allclose(v.item(), gold[f"step{step}/{k}"], rtol=2e-4, err_msg=f"{step}/{k}")
        for n, p in gen.named_parameters():
            d = np.array([p.detach().double().sum().item(), p.detach().double().abs().sum().item()])
            np.testing.assert_allclose(d, gold[f"step{step}/gen/{n}"], rtol=1e-3, atol=1e-4, err_msg=n)
        for n, p in discr.named_parameters():
            d = np.array([p.detach().double().sum().item(), p.detach().double().abs().sum().item()])
            np.testing.assert_allclose(d, gold[f"step{step}/discr/{n}"], rtol=1e-3, atol=1e-4, err_msg=n)
    # all parameters require grad again after the step (toggle/untoggle restored)
    assert all(p.requires_grad for p in gen.parameters())
    assert all(p.requires_grad for p in discr.parameters())


def test_storage_emulation_rounds_where_the_bf16_mode_stores():
    """oracle.storage_emulation (test infrastructure for the bf16 GPU tests): off = bit-identical to the plain oracle;
    on = outputs and data gradients are bf16-representable, weight gradients stay full-precision f32 sums and reach the
    f32 master weights (straight-through)."""
    import torch
    from oracle import unet_ref as R
    torch.manual_seed(0)
    m = R.RefDownSampleConv(16, 32).train()
    x = torch.rand(2, 16, 8, 8, 8)
    y0 = m(x)
    with R.storage_emulation(None):
        assert torch.equal(m(x), y0)
    xr = x.clone().requires_grad_(True)
    with R.storage_emulation(torch.bfloat16):
        y1 = m(xr)
        (y1 * torch.rand_like(y1)).sum().backward()
    assert torch.equal(y1, y1.to(torch.bfloat16).float()) and not torch.equal(y1, y0)
    assert (y1 - y0).abs().max() <= 0.05 * y0.abs().max()
    w = m.conv.weight
    assert w.grad is not None and not torch.equal(w.grad, w.grad.to(torch.bfloat16).float())
    assert xr.grad is not None          # (x itself is rounded by the caller: Generator / Discriminator entry)
