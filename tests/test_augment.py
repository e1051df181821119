"""GPU augmentations (augmentation part): oracle known answers on the CPU, kernels vs oracle."""
import numpy as np
import pytest
import torch

from oracle import augment_ref as AR


def test_oracle_bias_field_known_answers():
    assert AR.n_coefficients(3) == 20 and AR.n_coefficients(0) == 1
    f = AR.bias_field((4, 6, 8), np.zeros(20))
    np.testing.assert_array_equal(f, np.ones((4, 6, 8), np.float32))
    c = np.zeros(20); c[0] = 0.3                                   # constant term only
    np.testing.assert_allclose(AR.bias_field((4, 4, 4), c), np.exp(0.3), rtol=1e-6)
    c = np.zeros(20); c[1] = 1.0                                   # first non-constant monomial = z (innermost loop)
    f = AR.bias_field((3, 5, 8), c)
    np.testing.assert_allclose(f[0, 0, :], np.exp((np.arange(-4, 4) + 0.5) / 3.5), rtol=1e-6)
    assert np.allclose(f[0, 0], f[2, 4])
    x = np.array([-0.5, 0.0, 0.25, 2.0])
    np.testing.assert_allclose(AR.apply_gamma(x, 2.0), [-0.25, 0.0, 0.0625, 4.0])


def test_crop_or_pad_cpu():
    from unet_bssfp_amd import augment as A
    x = torch.arange(2 * 5 * 6 * 7, dtype=torch.float32).reshape(2, 5, 6, 7)
    y = A.crop_or_pad(x, (3, 8, 7))
    assert y.shape == (2, 3, 8, 7)
    assert torch.equal(y[:, :, 1:7, :], x[:, 1:4])
    assert float(y[:, :, 0].abs().max()) == 0 and float(y[:, :, 7].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,order", [((3, 16, 24, 32), 3), ((1, 7, 9, 11), 3), ((2, 8, 8, 8), 1), ((6, 32, 32, 32), 4)])
def test_gpu_bias_field_and_gamma_match_oracle(shape, order):
    from unet_bssfp_amd import augment as A
    rng = np.random.default_rng(sum(shape))
    x = (rng.random(shape) - 0.2).astype(np.float32)
    coef = (rng.random(AR.n_coefficients(order)) - 0.5).astype(np.float32)
    got = A.RandomBiasField(order=order).apply(torch.from_numpy(x).cuda(), coef).cpu().numpy()
    np.testing.assert_allclose(got, AR.apply_bias_field(x, coef, order), rtol=2e-5, atol=1e-6)
    g = 1.27
    got = A.RandomGamma().apply(torch.from_numpy(x).cuda(), g).cpu().numpy()
    np.testing.assert_allclose(got, AR.apply_gamma(x.astype(np.float64), g), rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_noise_statistics_and_subject_interface():
    from unet_bssfp_amd import augment as A, _lib
    x = torch.zeros(4, 64, 64, 64, device="cuda")
    n = A.RandomNoise()
    a = n.apply(x, (0.25, 0.1, 1234))
    assert abs(a.mean().item() - 0.25) < 2e-4 and abs(a.std().item() - 0.1) < 2e-4
    flat = (a.flatten()[:-1] - 0.25) * (a.flatten()[1:] - 0.25)
    assert abs(flat.mean().item()) < 1e-4                              # neighbouring voxels uncorrelated
    k = (((a - 0.25) / 0.1) ** 4).mean().item()
    assert abs(k - 3.0) < 0.05                                         # Gaussian kurtosis
    assert torch.equal(a, n.apply(x, (0.25, 0.1, 1234))) and not torch.equal(a, n.apply(x, (0.25, 0.1, 1235)))
    torch.manual_seed(0)
    subj = {"bssfp": {"data": torch.rand(24, 16, 16, 16, device="cuda")}, "dwi-tensor": {"data": torch.rand(6, 16, 16, 16, device="cuda")}}
    out = A.RandomGamma(p=1.0)(subj)
    gamma = torch.log(out["bssfp"]["data"][0, 0, 0, 0]) / torch.log(subj["bssfp"]["data"][0, 0, 0, 0])
    got = out["dwi-tensor"]["data"][3, 5, 5, 5]
    assert torch.allclose(got, subj["dwi-tensor"]["data"][3, 5, 5, 5] ** gamma, rtol=1e-4)    # one gamma for the whole subject
    assert A.RandomGamma(p=0.0)(subj) is subj
    assert [type(t).__name__ for t in A.reference_augmentation()] == ["RandomBiasField", "RandomNoise", "RandomGamma"]
    with pytest.raises(_lib.Mi355Error):
        A.RandomGamma().apply(torch.zeros(1, 2, 2, 2), 1.0)
