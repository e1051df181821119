"""Sliding-window inference.  TorchIO is absent: the oracle restates its published
algorithm (parity unpinned), so besides oracle-vs-HIP equality the tests pin properties: exact round trip,
full coverage, overwrite order."""
import numpy as np
import pytest
import torch

from oracle import grid_ref
from unet_bssfp_amd import inference as I


def test_locations_reference_volume():
    # the reference's test volume: CropOrPad((96, 128, 128)) (src/data_module.py:124-127), patch_sz=64 (:18)
    locs = I.grid_locations((96, 128, 128), 64)
    assert locs.shape == (8, 6)
    assert sorted(set(locs[:, 0])) == [0, 32] and sorted(set(locs[:, 1])) == [0, 64] and sorted(set(locs[:, 2])) == [0, 64]
    assert (locs[:, 3:] - locs[:, :3] == 64).all()
    assert [tuple(l[:3]) for l in locs] == sorted(tuple(l[:3]) for l in locs)


@pytest.mark.parametrize("shape,patch,overlap", [((96, 128, 128), 64, 0), ((40, 33, 50), (16, 16, 32), (4, 0, 8)),
                                                 ((16, 16, 16), 16, 0), ((20, 20, 20), 8, 2), ((17, 9, 30), (5, 9, 7), (2, 0, 4))])
def test_locations_match_oracle_and_cover(shape, patch, overlap):
    locs = I.grid_locations(shape, patch, overlap)
    np.testing.assert_array_equal(locs, grid_ref.grid_locations(shape, patch, overlap))
    cover = np.zeros(shape, dtype=int)
    for l in locs:
        assert (l[:3] >= 0).all() and (l[3:] <= np.array(shape)).all()
        ki, kf = grid_ref.kept_region(l, shape, overlap)
        cover[ki[0]:kf[0], ki[1]:kf[1], ki[2]:kf[2]] += 1
    assert (cover >= 1).all()                                   # crop mode leaves no hole
    rng = np.random.default_rng(0)
    vol = rng.random((3,) + tuple(shape)).astype(np.float32)
    for mode in ("crop", "average"):                            # aggregating the input patches returns the input
        out = grid_ref.aggregate(grid_ref.extract(vol, locs), locs, shape, overlap, mode)
        if mode == "crop":
            np.testing.assert_array_equal(out, vol)
        else:
            np.testing.assert_allclose(out, vol, rtol=1e-6)


def test_location_errors():
    with pytest.raises(ValueError):
        I.grid_locations((32, 32, 32), 64)
    with pytest.raises(ValueError):
        I.grid_locations((32, 32, 32), 16, 3)
    with pytest.raises(ValueError):
        I.grid_locations((32, 32, 32), 16, 16)
    with pytest.raises(Exception):
        I.GridSampler({"x": {"data": torch.zeros(1, 8, 8, 8)}}, 4)        # CPU tensor: no fallback


@pytest.mark.gpu
@pytest.mark.parametrize("shape,patch,overlap", [((24, 32, 40), (8, 16, 8), 0), ((40, 33, 50), (16, 16, 32), (4, 0, 8)),
                                                 ((20, 20, 20), 8, 2), ((17, 9, 30), (5, 9, 7), (2, 0, 4))])
def test_gpu_gather_and_aggregate_bit_exact(shape, patch, overlap):
    rng = np.random.default_rng(1)
    vol = rng.standard_normal((5,) + shape).astype(np.float32)
    sampler = I.GridSampler({"a": {"data": torch.from_numpy(vol).cuda()}}, patch, overlap)
    locs = grid_ref.grid_locations(shape, patch, overlap)
    np.testing.assert_array_equal(sampler.locations, locs)
    want = grid_ref.extract(vol, locs)
    got = torch.cat([b["a"]["data"] for b in sampler.batches(7)])
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(sampler[len(sampler) - 1]["a"]["data"].cpu().numpy(), want[-1])
    np.testing.assert_array_equal(sampler[-1]["location"].numpy(), locs[-1])
    # aggregate DIFFERENT values per patch so that the overwrite order / averaging is visible
    pred = rng.standard_normal(want.shape).astype(np.float32)
    for mode in ("crop", "average"):
        for bs in (len(locs), 3, 100):                          # one launch, several launches
            agg = I.GridAggregator(sampler, mode)
            for b in range(0, len(locs), bs):
                agg.add_batch(torch.from_numpy(pred[b:b + bs]).cuda(), torch.from_numpy(locs[b:b + bs]))
            out = agg.get_output_tensor().cpu().numpy()
            np.testing.assert_array_equal(out, grid_ref.aggregate(pred, locs, shape, overlap, mode), err_msg=f"{mode} bs={bs}")


@pytest.mark.gpu
def test_gpu_many_patches_chunking_and_errors():
    from unet_bssfp_amd import _lib
    shape, patch = (12, 12, 12), 4                               # 27 patches ... with overlap 2 -> 125 > MI355_MAX_PATCHES
    vol = torch.arange(2 * 12 ** 3, dtype=torch.float32, device="cuda").reshape(2, *shape)
    sampler = I.GridSampler({"v": {"data": vol}}, patch, 2)
    assert len(sampler) == 125
    agg = I.GridAggregator(sampler)
    for b in sampler.batches(125):
        agg.add_batch(b["v"]["data"], b["location"])
    assert torch.equal(agg.get_output_tensor(), vol)
    with pytest.raises(ValueError):
        agg.add_batch(torch.zeros(1, 2, 4, 4, 4, device="cuda"), np.array([[0, 0, 0, 5, 4, 4]]))
    with pytest.raises(_lib.Mi355Error):
        agg.add_batch(torch.zeros(1, 2, 4, 4, 4, device="cuda"), np.array([[10, 0, 0, 14, 4, 4]]))
    with pytest.raises(NotImplementedError):
        I.GridAggregator(sampler, "hann")
    with pytest.raises(ValueError):
        I.GridAggregator(sampler, "max")


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,mode", [(0, "crop"), (8, "crop"), (8, "average")])
def test_gpu_predict_volume_matches_oracle(overlap, mode):
    """eval-mode generator over the grid of a (24, 48, 64, 64) volume, 32^3 patches: per-voxel L1 <= 1e-4"""
    from oracle import unet_ref as R
    from unet_bssfp_amd import nn as N
    torch.manual_seed(3)
    ref = R.RefGenerator("bssfp")
    gen = N.Generator("bssfp")
    gen.load_state_dict(ref.state_dict())
    gen.cuda()
    vol = torch.rand(24, 48, 64, 64, generator=torch.Generator().manual_seed(5))
    want = grid_ref.predict_volume(ref, vol.numpy(), 32, overlap, batch_size=4, overlap_mode=mode)
    got = I.predict_volume(gen, vol.cuda(), 32, overlap, batch_size=5, overlap_mode=mode).cpu().numpy()
    assert got.shape == (6, 48, 64, 64)
    assert np.abs(got - want).mean() <= 1e-4 and np.abs(got - want).max() <= 5e-3
    assert gen.training                                          # mode restored


@pytest.mark.gpu
def test_gpu_predict_step_mirrors_reference_loop():
    from unet_bssfp_amd import gan, nn as N
    torch.manual_seed(4)
    model = gan.bSSFPToDWITensorModel("bssfp", batch_size=3, gen=N.Generator("bssfp").cuda(), discr=N.Discriminator("bssfp").cuda())
    model.eval()
    g = torch.Generator().manual_seed(6)
    subject = {"bssfp": {"data": torch.rand(24, 32, 48, 32, generator=g).cuda()},
               "dwi-tensor": {"data": torch.rand(6, 32, 48, 32, generator=g).cuda()}}
    sampler = I.GridSampler(subject, 32)
    out = model.predict_step((sampler, I.GridAggregator(sampler), I.GridAggregator(sampler), I.GridAggregator(sampler)))
    assert torch.equal(out.x, subject["bssfp"]["data"])          # aggregated input == input (what the reference returns)
    assert torch.equal(out.y, subject["dwi-tensor"]["data"])
    assert torch.equal(out.y_hat, I.predict_volume(model.gen, subject["bssfp"]["data"], 32, batch_size=3))


@pytest.mark.gpu
def test_gpu_test_step_logs_and_sums_patch_losses():
    from unet_bssfp_amd import gan, nn as N
    torch.manual_seed(4)
    model = gan.bSSFPToDWITensorModel("bssfp", batch_size=2, gen=N.Generator("bssfp").cuda(), discr=N.Discriminator("bssfp").cuda())
    model.eval()
    g = torch.Generator().manual_seed(6)
    subject = {"bssfp": {"data": torch.rand(24, 64, 64, 96, generator=g).cuda()},
               "dwi-tensor": {"data": torch.rand(6, 64, 64, 96, generator=g).cuda()}}
    sampler = I.GridSampler(subject, 64)                          # 2 patches -> one batch of 2
    tot = model.test_step((sampler, I.GridAggregator(sampler), I.GridAggregator(sampler), I.GridAggregator(sampler)))
    logs = {k: float(v) for k, v in model.last_logs.items()}
    assert {"test_gen_loss_subject", "test_gen_loss_recon_L1", "test_gen_loss_adversarial", "test_metric_PSNR", "test_metric_SSIM",
            "test_metric_L1"} <= set(logs)
    assert float(tot) == pytest.approx(logs["test_gen_loss_subject"]) and np.isfinite(float(tot))
    assert abs(logs["test_metric_L1"] - logs["test_gen_loss_recon_L1"]) < 5e-3      # overlapping patches: last one wins in the volume
