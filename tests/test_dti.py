"""DTI scalar maps: oracle vs the golden vectors produced by the reference's own
voxel loop (CPU), HIP kernel vs oracle / golden (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import dti_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dti_scalar_maps.npz")
SIGN_FREE = ("fa", "md", "ad", "rd", "rgb")


def test_oracle_matches_reference_loop():
    g = np.load(GOLD)
    o = dti_ref.scalar_maps(g["data"])
    for k in SIGN_FREE + ("azimuth", "inclination"):
        np.testing.assert_array_equal(o[k], g[k], err_msg=k)          # same LAPACK call: bit-exact


def test_oracle_edge_cases():
    o = dti_ref.scalar_maps(np.array([[1e-3, 0, 0, 1e-3, 0, 1e-3], [0, 0, 0, 0, 0, 0.0]]))
    assert o["fa"][0] == 0 and np.isnan(o["fa"][1])
    assert o["md"][0] == pytest.approx(1e-3)
    d = dti_ref.invert_dwi_tensor_norm(np.array([0.0, 0.5, 1.0]), -2.0, 6.0)
    np.testing.assert_allclose(d, [-2.0, 2.0, 6.0])


def _families(n, rng):
    """tensors with prescribed spectra, incl. (nearly) repeated eigenvalues at every gap from 1e-16 to 1e-1"""
    def fromeig(w):
        q = np.linalg.qr(rng.standard_normal((len(w), 3, 3)))[0]
        m = q @ (w[:, :, None] * np.swapaxes(q, -1, -2))
        return np.stack([m[:, 0, 0], m[:, 0, 1], m[:, 0, 2], m[:, 1, 1], m[:, 1, 2], m[:, 2, 2]], -1)
    eps, one = 10.0 ** rng.uniform(-16, -1, n), np.ones(n)
    return np.concatenate([fromeig(np.stack([one, 3 * one, 3 + eps], -1)), fromeig(np.stack([one, 1 + eps, 3 * one], -1)),
                           fromeig(np.stack([one, 1 + eps, 1 + 2 * eps], -1)), fromeig(np.stack([-one, eps, one], -1)) * 1e-3,
                           rng.random((n, 6)), dti_ref.synthetic_tensor_field((n,), seed=2)])


def test_kernel_arithmetic_on_host_matches_lapack(tmp_path):
    """csrc/dti_core.h is plain C++: the SAME per-voxel code the HIP kernel runs, compiled with g++ and
    compared with the oracle (LAPACK) -- eigenvalues to a few ulp of |A| for every eigenvalue gap, the
    principal direction wherever it is defined (gap > 1e-9), and A e = w e everywhere."""
    import subprocess
    exe = tmp_path / "hdc"
    subprocess.check_call(["g++", "-O2", "-o", str(exe), os.path.join(os.path.dirname(__file__), "host_dti_check.cpp"), "-lm"])
    d = _families(4000, np.random.default_rng(0))
    d.tofile(tmp_path / "in.bin")
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(len(d)), "0"])
    o = np.fromfile(tmp_path / "out.bin").reshape(-1, 9)
    ref = dti_ref.scalar_maps(d)
    scale = np.sqrt((d ** 2).sum(-1))
    for i, k in enumerate(("fa", "md", "ad", "rd")):
        err = np.abs(o[:, i] - ref[k]) / (1.0 if k == "fa" else scale)
        assert err.max() < 1e-14, k
    m = _full(d)
    w = np.linalg.eigvalsh(m)
    az, inc = np.deg2rad(o[:, 4]), np.deg2rad(o[:, 5])
    e = np.stack([np.sin(inc) * np.cos(az), np.sin(inc) * np.sin(az), np.cos(inc)], -1)
    assert (np.linalg.norm(np.einsum("nij,nj->ni", m, e) - o[:, 2:3] * e, axis=-1) / scale).max() < 1e-12
    defined = (w[:, 2] - w[:, 1]) > 1e-9 * np.abs(w).max(-1)
    assert dti_ref.antipodal_close(o[:, 4], o[:, 5], ref["azimuth"], ref["inclination"], 1e-4)[defined].all()
    np.testing.assert_allclose(o[:, 6:][defined], ref["rgb"][defined], atol=1e-6)


def _check(out, ref, data, rtol, atol_deg):
    """eigenvalue maps by relative tolerance (scaled by the tensor norm), directions up to the sign,
    and only where the principal eigenvalue is simple (otherwise the direction is not defined)."""
    scale = np.sqrt((data.reshape(-1, 6) ** 2).sum(-1)).reshape(ref["md"].shape) + 1e-300
    for k in ("md", "ad", "rd"):
        assert np.nanmax(np.abs(out[k] - ref[k]) / scale) <= rtol, k
    nan_ref = np.isnan(ref["fa"])
    assert np.array_equal(np.isnan(out["fa"]), nan_ref)
    assert np.nanmax(np.abs(out["fa"] - ref["fa"])) <= 10 * rtol
    w = np.linalg.eigvalsh(_full(data))
    simple = (w[..., 2] - w[..., 1]) > 1e-3 * np.abs(w).max(-1)
    ok = dti_ref.antipodal_close(out["azimuth"], out["inclination"], ref["azimuth"], ref["inclination"], atol_deg)
    assert ok[simple].all(), f"{(~ok[simple]).sum()} directions differ"
    assert np.nanmax(np.abs(out["rgb"] - ref["rgb"])[simple]) <= 1e3 * rtol
    assert (out["inclination"][~nan_ref] <= 90 + 1e-9).all()          # the documented sign convention


def _full(d):
    m = np.empty(d.shape[:-1] + (3, 3))
    m[..., 0, 0], m[..., 0, 1], m[..., 0, 2], m[..., 1, 1], m[..., 1, 2], m[..., 2, 2] = np.moveaxis(d, -1, 0)
    m[..., 1, 0], m[..., 2, 0], m[..., 2, 1] = m[..., 0, 1], m[..., 0, 2], m[..., 1, 2]
    return m


@pytest.mark.gpu
def test_gpu_matches_golden_f64():
    from unet_bssfp_amd import eval as E
    g = np.load(GOLD)
    out = E.calc_scalar_maps(torch.from_numpy(g["data"]).cuda())
    out = {k: v.cpu().numpy() for k, v in out.items()}
    _check(out, {k: g[k] for k in E.MAP_NAMES}, g["data"], rtol=1e-13, atol_deg=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 5, 257), (32, 32, 32)])
def test_gpu_matches_oracle_f32_both_layouts(shape):
    from unet_bssfp_amd import eval as E
    data = dti_ref.synthetic_tensor_field(shape, seed=11).astype(np.float32)
    ref = dti_ref.scalar_maps(data)                                     # f64 arithmetic on the f32 inputs
    a = E.calc_scalar_maps(torch.from_numpy(data).cuda())
    b = E.calc_scalar_maps(torch.from_numpy(np.ascontiguousarray(np.moveaxis(data, -1, 0))).cuda(), channels_first=True)
    for k in E.MAP_NAMES:
        assert torch.equal(a[k].isnan(), b[k].isnan()) and torch.equal(a[k].nan_to_num(), b[k].nan_to_num()), k
    out = {k: v.double().cpu().numpy() for k, v in a.items()}
    _check(out, ref, data.astype(np.float64), rtol=2e-7, atol_deg=1e-3)  # f32 output rounding


@pytest.mark.gpu
def test_gpu_fused_denorm_and_unit_interval_inputs():
    from unet_bssfp_amd import eval as E
    rng = np.random.default_rng(5)
    data = rng.random((16, 16, 16, 6))                                   # U[0,1) like a generator output
    mn, mx = -3.1e-3, 4.2e-3
    ref = dti_ref.scalar_maps(dti_ref.invert_dwi_tensor_norm(data, mn, mx))
    out = E.calc_scalar_maps(torch.from_numpy(data).cuda(), mn, mx)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    _check(out, ref, dti_ref.invert_dwi_tensor_norm(data, mn, mx), rtol=1e-13, atol_deg=1e-4)


@pytest.mark.gpu
def test_gpu_degenerate_voxels_and_errors():
    from unet_bssfp_amd import eval as E, _lib
    d = np.array([[1e-3, 0, 0, 1e-3, 0, 1e-3],        # isotropic
                  [3e-3, 0, 0, 2e-3, 0, 1e-3],        # diagonal
                  [0, 0, 0, 0, 0, 0.0],               # zero tensor -> FA NaN
                  [2.0, 1.0, 0.0, 2.0, 0.0, 3.0],     # double LARGEST eigenvalue (3, 3, 1)
                  [1.0, 1.0, 1.0, 1.0, 1.0, 1.0]])    # rank one (3, 0, 0)
    out = {k: v.cpu().numpy() for k, v in E.calc_scalar_maps(torch.from_numpy(d).cuda()).items()}
    ref = dti_ref.scalar_maps(d)
    for k in ("md", "ad", "rd"):
        np.testing.assert_allclose(out[k], ref[k], rtol=0, atol=2e-14, err_msg=k)
    np.testing.assert_allclose(out["fa"][[0, 1, 3, 4]], ref["fa"][[0, 1, 3, 4]], atol=1e-12)
    assert np.isnan(out["fa"][2])
    # principal direction of the double eigenvalue must lie in the eigen-plane: A e = 3 e
    az, inc = np.deg2rad(out["azimuth"][3]), np.deg2rad(out["inclination"][3])
    e = np.array([np.sin(inc) * np.cos(az), np.sin(inc) * np.sin(az), np.cos(inc)])
    np.testing.assert_allclose(_full(d[3]) @ e, 3 * e, atol=1e-9)
    assert dti_ref.antipodal_close(out["azimuth"][4], out["inclination"][4], ref["azimuth"][4], ref["inclination"][4], 1e-4)
    # every eigenvalue gap, on the device
    fam = _families(2000, np.random.default_rng(1))
    o = {k: v.cpu().numpy() for k, v in E.calc_scalar_maps(torch.from_numpy(fam).cuda()).items()}
    r = dti_ref.scalar_maps(fam)
    sc = np.sqrt((fam ** 2).sum(-1))
    for k in ("md", "ad", "rd"):
        assert (np.abs(o[k] - r[k]) / sc).max() < 1e-14, k
    assert np.abs(o["fa"] - r["fa"]).max() < 1e-14
    assert E.calc_scalar_maps(torch.zeros(0, 6, device="cuda"))["fa"].numel() == 0
    with pytest.raises(ValueError):
        E.calc_scalar_maps(torch.zeros(4, 5, device="cuda"))
    with pytest.raises(_lib.Mi355Error):
        E.calc_scalar_maps(torch.zeros(4, 6))
