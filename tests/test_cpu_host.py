"""CPU-only tests (no GPU, no compute calls): the C-ABI library loads and exports every symbol the
header declares, host-side module logic (constructor signatures, state_dict keys, init parity with
torch), error behaviour without a GPU, and the GAN step harness driven with the oracle modules."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import unet_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi355_unet.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from unet_bssfp_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mi355_unet.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_library_version_and_error_string_without_gpu():
    from unet_bssfp_amd import _lib
    lib = _lib.load()
    assert lib.mi355_version() >= 100
    # argument validation happens on the host before any launch: a null descriptor is an error, not a crash
    rc = lib.mi355_conv_fwd(None, None)
    assert rc < 0
    assert b"null" in lib.mi355_last_error()
    assert lib.mi355_l1_blocks(4096 * 3 + 1) == 4
    assert lib.mi355_channel_stats_blocks(2 ** 21) == 1024
    assert lib.mi355_channel_stats_blocks(32 ** 3) == 256 and lib.mi355_channel_stats_blocks(100) == 6
    assert lib.mi355_channel_stats_blocks(16 ** 3) == 256 and lib.mi355_channel_stats_blocks(8) == 1     # >= min(256, rows / 16) workgroups


def test_missing_library_fails_loudly(monkeypatch):
    from unet_bssfp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmi355_unet.so")
    with pytest.raises(_lib.Mi355Error):
        _lib.load()


def test_modules_match_reference_signatures_keys_and_init():
    import unet_bssfp_amd as M
    for modality in ("bssfp", "pc-bssfp", "dwi-tensor", "t1w"):
        torch.manual_seed(42)
        g, d = M.Generator(modality), M.Discriminator(modality)
        torch.manual_seed(42)
        rg, rd = R.RefGenerator(modality), R.RefDiscriminator(modality)
        for ours, ref in ((g, rg), (d, rd)):
            sd, rsd = ours.state_dict(), ref.state_dict()
            assert list(sd.keys()) == list(rsd.keys())
            for k in rsd:
                assert sd[k].shape == rsd[k].shape and torch.equal(sd[k], rsd[k]), k
    assert sum(p.numel() for p in M.Generator("bssfp").parameters()) == 22_646_182
    assert sum(p.numel() for p in M.Discriminator("bssfp").parameters()) == 11_230_593


def test_constructor_signatures_of_the_construction_sites():
    import inspect
    import unet_bssfp_amd as M
    assert list(inspect.signature(M.DownSampleConv.__init__).parameters)[1:] == [
        "in_channels", "out_channels", "kernel", "strides", "padding", "activation", "batchnorm"]   # src/model.py:43-46
    p = inspect.signature(M.BasicUNet.__init__).parameters
    assert list(p)[1:] == ["spatial_dims", "in_channels", "out_channels", "features", "act", "norm", "bias",
                           "dropout", "upsample"]                                               # MONAI 1.3.0
    u = M.BasicUNet(spatial_dims=3, in_channels=24, out_channels=6, features=(32, 64, 128, 256, 512, 32), dropout=0.05)
    assert u.upcat_1.convs.conv_0.conv.weight.shape == (32, 96, 3, 3, 3)
    u2 = M.BasicUNet(spatial_dims=2, in_channels=1, out_channels=6)             # BASELINE configs[0]: MONAI's 2-D shapes
    assert u2.upcat_1.convs.conv_0.conv.weight.shape == (32, 64, 3, 3) and u2.upcat_1.upsample.deconv.weight.shape == (32, 32, 2, 2)
    with pytest.raises(NotImplementedError):
        M.BasicUNet(spatial_dims=1, in_channels=1, out_channels=6)
    with pytest.raises(NotImplementedError):
        M.BasicUNet(spatial_dims=3, in_channels=1, out_channels=6, upsample="nontrainable")


def test_cpu_input_raises_instead_of_falling_back():
    import unet_bssfp_amd as M
    from unet_bssfp_amd import _lib
    with pytest.raises(_lib.Mi355Error):
        M.Discriminator("bssfp")(torch.rand(1, 24, 64, 64, 64), torch.rand(1, 6, 64, 64, 64))
    with pytest.raises(_lib.Mi355Error):
        M.l1_loss(torch.rand(4), torch.rand(4))


def test_gan_harness_with_oracle_modules_matches_oracle_step():
    """gan.bSSFPToDWITensorModel is module-agnostic: driven with the oracle networks on CPU it must
    reproduce oracle.gan_training_step (the restated src/model.py:259-281) exactly."""
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    torch.manual_seed(0)
    g1, d1 = R.RefGenerator("bssfp", dropout=0.0).train(), R.RefDiscriminator("bssfp").train()
    g2, d2 = R.RefGenerator("bssfp", dropout=0.0).train(), R.RefDiscriminator("bssfp").train()
    g2.load_state_dict(g1.state_dict())
    d2.load_state_dict(d1.state_dict())
    model = bSSFPToDWITensorModel("bssfp", gen=g1, discr=d1, optimizer_class=torch.optim.AdamW)
    batch = synthetic_batch(2, 32, seed=5)
    x, y = R.synthetic_batch(2, 32, seed=5)
    assert torch.equal(batch["bssfp"]["data"], x) and torch.equal(batch["dwi-tensor_orig"]["data"], y)
    g_opt, d_opt = R.make_optimizers(g2, d2)
    for step in range(2):
        model.training_step(batch, step)
        ref = R.gan_training_step(g2, d2, g_opt, d_opt, x, y)
        for k in ("gen_loss_adversarial", "gen_loss_recon_L1", "gen_loss_recon", "gen_loss", "discr_loss"):
            assert float(model.last_logs["train_" + k]) == pytest.approx(float(ref[k]), rel=1e-6), (step, k)   # the reference's log names
    for (n, p), (_, q) in zip(g1.named_parameters(), g2.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n
    assert model.stacked_logs().shape == (5,)
    assert all(p.requires_grad for p in model.parameters())


def test_unpack_batch_layout():
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    m = bSSFPToDWITensorModel("t1w", gen=torch.nn.Identity(), discr=torch.nn.Identity())
    b = synthetic_batch(1, 8, seed=1, modality="t1w")
    x, y = m.unpack_batch(b)
    assert x.shape == (1, 6, 8, 8, 8) and y.shape == (1, 6, 8, 8, 8)
    assert m.unpack_batch(b, test=True)[1] is b["dwi-tensor"]["data"]


def test_weight_pack_index_algebra_matches_conv_definitions():
    """The (s_co, s_ci, s_k, tbase, tstep) tuples handed to mi355_weight_pack express forward,
    flipped-transposed (data gradient) and the stride-2 parity classes.  Check the algebra in numpy
    against torch convolutions (host logic only, no kernel)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(0)

    def gather(w, cout, cin, ks, s_co, s_ci, s_k, tb, ts):
        flat = w.reshape(-1)
        out = np.zeros((ks, ks, ks, cout, cin), dtype=w.dtype)
        for td in range(ks):
            for th in range(ks):
                for tw in range(ks):
                    off = (tb[0] + ts[0] * td) * s_k[0] + (tb[1] + ts[1] * th) * s_k[1] + (tb[2] + ts[2] * tw) * s_k[2]
                    idx = off + np.arange(cout)[:, None] * s_co + np.arange(cin)[None, :] * s_ci
                    out[td, th, tw] = flat[idx]
        return out

    def conv_taps(x, wt, stride, pad_lo, out_sz):
        # direct evaluation of z[p, co] = sum_tap,ci x[p*stride + tap - pad, ci] * wt[tap][co][ci]
        ks = wt.shape[0]
        cin = x.shape[0]
        z = np.zeros((wt.shape[3],) + tuple(out_sz))
        xp = np.pad(x, ((0, 0),) + ((4, 4),) * 3)
        for td in range(ks):
            for th in range(ks):
                for tw in range(ks):
                    sl = xp[:, 4 + td - pad_lo[0]: 4 + td - pad_lo[0] + stride * out_sz[0]: stride,
                            4 + th - pad_lo[1]: 4 + th - pad_lo[1] + stride * out_sz[1]: stride,
                            4 + tw - pad_lo[2]: 4 + tw - pad_lo[2] + stride * out_sz[2]: stride]
                    z += np.einsum("oc,cdhw->odhw", wt[td, th, tw][:, :cin], sl)
        return z

    # data gradient of k4 s2 p1 through the 8 parity classes (functional.ConvSpec.w_dgrad_s2)
    cin, cout, k = 3, 5, 4
    w = rng.standard_normal((cout, cin, k, k, k))
    g = rng.standard_normal((cout, 3, 3, 3))                       # grad wrt conv output (6^3 input)
    ref = F.conv_transpose3d(torch.from_numpy(g)[None], torch.from_numpy(w), stride=2, padding=1)[0].numpy()
    dx = np.zeros((cin, 6, 6, 6))
    for cls in [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]:
        tb = tuple(3 if p == 0 else 2 for p in cls)
        wt = gather(w, cin, cout, 2, k ** 3, cin * k ** 3, (k * k, k, 1), tb, (-2, -2, -2))
        z = conv_taps(g, wt, 1, tuple(1 if p == 0 else 0 for p in cls), (3, 3, 3))
        dx[:, cls[0]::2, cls[1]::2, cls[2]::2] = z
    np.testing.assert_allclose(dx, ref, rtol=1e-10, atol=1e-10)

    # data gradient of k3 s1 p1 = correlation with the flipped, transposed kernel, pad k-1-p
    w3 = rng.standard_normal((cout, cin, 3, 3, 3))
    g3 = rng.standard_normal((cout, 4, 4, 4))
    ref3 = F.conv_transpose3d(torch.from_numpy(g3)[None], torch.from_numpy(w3), stride=1, padding=1)[0].numpy()
    wt = gather(w3, cin, cout, 3, 27, cin * 27, (9, 3, 1), (2, 2, 2), (-1, -1, -1))
    np.testing.assert_allclose(conv_taps(g3, wt, 1, (1, 1, 1), (4, 4, 4)), ref3, rtol=1e-10, atol=1e-10)

    # forward of ConvTranspose3d(k2, s2) as 8 one-tap classes (functional.ConvSpec.w_deconv_fwd)
    wd = rng.standard_normal((cin, cout, 2, 2, 2))
    xd = rng.standard_normal((cin, 3, 3, 3))
    refd = F.conv_transpose3d(torch.from_numpy(xd)[None], torch.from_numpy(wd), stride=2)[0].numpy()
    out = np.zeros((cout, 6, 6, 6))
    for cls in [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]:
        wt = gather(wd, cout, cin, 1, 8, cout * 8, (4, 2, 1), cls, (0, 0, 0))
        out[:, cls[0]::2, cls[1]::2, cls[2]::2] = conv_taps(xd, wt, 1, (0, 0, 0), (3, 3, 3))
    np.testing.assert_allclose(out, refd, rtol=1e-10, atol=1e-10)


def test_lds_dma_kernels_contain_no_compiler_generated_m0_use():
    """common.h: dma_lds_b128 overwrites m0 inside an asm statement and cannot declare it (hipcc rejects the clobber).  That is safe
    only while the compiler itself neither sets nor reads m0 in those kernels: no movrel / s_set_gpr_idx (dynamically indexed
    register arrays), no LDS-direct / addtid / GWS / s_sendmsg use, no m0 operand other than the statement's own `s_mov_b32 m0`.
    The build emits the device ISA of the two sources that hold such kernels (csrc/Makefile: *.gfx950.s); this scans it."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "unet_bssfp_amd", "csrc", "*.gfx950.s")))
    if not files:
        pytest.skip("device ISA listings not built (make -C unet_bssfp_amd/csrc)")
    forbidden = re.compile(r"\b(v_movrel\w*|s_set_gpr_idx\w*|ds_\w*addtid\w*|ds_gws\w*|s_sendmsg\w*|lds_direct\w*|v_interp\w*|s_movrel\w*)\b")
    dma = 0
    for f in files:
        for ln, line in enumerate(open(f), 1):
            code = line.split(";")[0]
            if not code.strip() or code.lstrip().startswith((".", "//")):
                continue
            assert not forbidden.search(code), f"{os.path.basename(f)}:{ln}: {code.strip()}"
            if re.search(r"\bm0\b", code):
                assert re.match(r"\s*s_mov_b32\s+m0\s*,", code), f"{os.path.basename(f)}:{ln}: m0 used by the compiler: {code.strip()}"
            dma += bool(re.search(r"buffer_load_dwordx4\b.*\blds\b", code))
    assert dma > 100, dma            # the listings really are those of the LDS-DMA kernels
