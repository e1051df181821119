"""Validation metrics: oracle known answers on the CPU, HIP vs oracle on the GPU."""
import math

import pytest
import torch

from oracle import metrics_ref as MR


def test_oracle_known_answers():
    torch.manual_seed(0)
    y = torch.rand(2, 3, 16, 16, 16)
    assert torch.allclose(MR.ssim3d(y, y), torch.ones(2, 1), atol=1e-6)
    assert torch.equal(MR.mae(y, y), torch.zeros(2, 1))
    p = y + 0.1
    assert torch.allclose(MR.mae(p, y), torch.full((2, 1), 0.1), atol=1e-6)
    assert torch.allclose(MR.psnr(p, y), torch.full((2, 1), 20.0), atol=1e-4)       # -10 log10(0.01)
    # constant images: sigma terms vanish, SSIM = (2ab + c1) / (a^2 + b^2 + c1)
    a, b = torch.full((1, 1, 12, 12, 12), 0.3), torch.full((1, 1, 12, 12, 12), 0.5)
    want = (2 * 0.3 * 0.5 + 1e-4) / (0.09 + 0.25 + 1e-4)
    assert abs(MR.ssim3d(a.double(), b.double()).item() - want) < 1e-5
    assert abs(MR.ssim3d(a, b).item() - want) < 2e-3     # f32: the variance terms are rounding noise against c2 = 9e-4
    g = MR.gaussian_1d()
    assert abs(g.sum().item() - 1) < 1e-6 and torch.allclose(g, g.flip(0)) and g.argmax().item() == 5


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 6, 32, 32, 32), (1, 6, 48, 40, 33), (3, 1, 11, 11, 11), (1, 2, 64, 64, 64)])
def test_gpu_metrics_match_oracle(shape):
    from unet_bssfp_amd import metrics as M
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.rand(shape, generator=g)
    p = (y + 0.2 * torch.randn(shape, generator=g)).clamp(0, 1)
    yd, pd = y.cuda(), p.cuda()
    for name, fn, ref, tol in (("mae", M.MAEMetric(), MR.mae, 1e-6), ("psnr", M.PSNRMetric(1), MR.psnr, 1e-4),
                               ("ssim", M.SSIMMetric(3, data_range=1), MR.ssim3d, 2e-5)):
        got, want = fn(pd, yd).cpu(), ref(p, y)
        assert got.shape == want.shape == (shape[0], 1), name
        assert torch.allclose(got, want, atol=tol, rtol=1e-5), (name, got, want)
    assert torch.equal(M.SSIMMetric(3)(pd, yd), M.SSIMMetric(3)(pd, yd))           # deterministic
    if min(shape[2:]) >= 16:                                                       # a non-default window: runtime-length loops
        got, want = M.SSIMMetric(3, win_size=7, kernel_sigma=1.0)(pd, yd).cpu(), MR.ssim3d(p, y, win_size=7, kernel_sigma=1.0)
        assert torch.allclose(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.gpu
def test_gpu_metrics_known_answers_and_errors():
    from unet_bssfp_amd import metrics as M, _lib
    y = torch.rand(2, 6, 24, 24, 24, device="cuda")
    assert torch.allclose(M.SSIMMetric(3)(y, y), torch.ones(2, 1, device="cuda"), atol=1e-6)
    assert torch.equal(M.MAEMetric()(y, y), torch.zeros(2, 1, device="cuda"))
    assert torch.isinf(M.PSNRMetric(1)(y, y)).all()                                # log10(0) like MONAI
    assert torch.allclose(M.PSNRMetric(1)(y + 0.1, y), torch.full((2, 1), 20.0, device="cuda"), atol=1e-3)
    names = [n for _, n in M.reference_metric_fns()]
    assert names == ["PSNR", "SSIM", "L1"]
    with pytest.raises(ValueError):
        M.MAEMetric()(y, y[:, :3])
    with pytest.raises(ValueError):
        M.SSIMMetric(3)(y[..., :8], y[..., :8])                                    # smaller than the window
    with pytest.raises(_lib.Mi355Error):
        M.MAEMetric()(y.cpu(), y.cpu())
    with pytest.raises(NotImplementedError):
        M.SSIMMetric(2)


@pytest.mark.gpu
def test_gpu_validation_step_logs_reference_keys():
    from oracle import unet_ref as R
    from unet_bssfp_amd import gan
    torch.manual_seed(0)
    model = gan.bSSFPToDWITensorModel("bssfp", batch_size=1).cuda().eval()
    x, y = R.synthetic_batch(1, 64, seed=2)
    loss = model.validation_step({"bssfp": {"data": x.cuda()}, "dwi-tensor_orig": {"data": y.cuda()}})
    logs = {k: float(v) for k, v in model.last_logs.items()}
    assert {"val_loss", "val_gen_loss_recon", "val_gen_loss_recon_L1", "val_gen_loss_adversarial", "val_metric_PSNR", "val_metric_SSIM",
            "val_metric_L1"} <= set(logs)
    assert abs(logs["val_metric_L1"] - logs["val_gen_loss_recon_L1"]) < 1e-5          # MAE metric == L1 loss term
    assert abs(logs["val_metric_PSNR"] + 10 * math.log10(max(logs["val_metric_L1"] ** 2, 1e-12))) < 6   # same order as -10 log10(mse)
    assert -1.0 <= logs["val_metric_SSIM"] <= 1.0 and float(loss) == pytest.approx(logs["val_loss"])
