"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Sliding-window (grid) inference, SURVEY.md 8(f) rank 1.

The reference's ``predict_step`` / ``test_step`` (/root/reference/src/model.py:291-333) walk a
``tio.inference.GridSampler(subject, patch_sz)`` (src/data_module.py:171-176, patch_overlap defaults to 0)
and feed three ``tio.inference.GridAggregator(sampler)`` (:177-183, overlap_mode defaults to 'crop').
TorchIO (requirements.txt: torchio==0.19.5) is NOT installed and not vendored, so what follows restates
its published algorithm -- PARITY UNPINNED for the TorchIO part; the tests anchor on properties instead
(exact round trip of the input volume, coverage, order of overwrites):

* patch locations per axis: range(0, size - patch + 1, patch - overlap) plus a last patch flush with the
  end when the stride does not land there; all axes combined, lexicographically sorted;
* 'crop' aggregation: each patch loses overlap // 2 voxels on every side that is not a volume border and is
  ASSIGNED into the output, in sampler order (later patches overwrite earlier ones);
* 'average' aggregation: uncropped patches are summed, a counter per voxel divides at the end.
"""
import numpy as np


def grid_locations(spatial_shape, patch_size, patch_overlap=0):
    spatial_shape = tuple(int(s) for s in spatial_shape)
    patch_size = _triple(patch_size)
    patch_overlap = _triple(patch_overlap)
    per_axis = []
    for size, p, o in zip(spatial_shape, patch_size, patch_overlap):
        if p > size:
            raise ValueError(f"patch size {p} larger than image size {size}")
        if o >= p or o % 2:
            raise ValueError(f"patch overlap {o} must be even and smaller than the patch size {p}")
        idx = list(range(0, size + 1 - p, p - o))
        if idx[-1] != size - p:
            idx.append(size - p)
        per_axis.append(idx)
    ini = np.array([(i, j, k) for i in per_axis[0] for j in per_axis[1] for k in per_axis[2]], dtype=np.int64)
    ini = np.unique(ini, axis=0)                       # sorted lexicographically
    return np.hstack([ini, ini + np.array(patch_size)])


def _triple(v):
    return (int(v),) * 3 if np.isscalar(v) else tuple(int(a) for a in v)


def kept_region(location, spatial_shape, patch_overlap):
    """volume-coordinate (ini, fin) of what 'crop' mode keeps of a patch"""
    half = np.array(_triple(patch_overlap)) // 2
    ini, fin = np.array(location[:3]), np.array(location[3:])
    return ini + np.where(ini == 0, 0, half), fin - np.where(fin == np.array(spatial_shape), 0, half)


def extract(volume, locations):
    """volume (C, D, H, W) -> patches (P, C, pd, ph, pw)"""
    return np.stack([volume[:, l[0]:l[3], l[1]:l[4], l[2]:l[5]] for l in locations])


def aggregate(patches, locations, spatial_shape, patch_overlap=0, overlap_mode="crop"):
    c = patches.shape[1]
    out = np.zeros((c,) + tuple(spatial_shape), dtype=patches.dtype)
    if overlap_mode == "crop":
        for p, l in zip(patches, locations):
            ki, kf = kept_region(l, spatial_shape, patch_overlap)
            oi = ki - l[:3]
            of = oi + (kf - ki)
            out[:, ki[0]:kf[0], ki[1]:kf[1], ki[2]:kf[2]] = p[:, oi[0]:of[0], oi[1]:of[1], oi[2]:of[2]]
        return out
    if overlap_mode == "average":
        cnt = np.zeros(tuple(spatial_shape), dtype=patches.dtype)
        for p, l in zip(patches, locations):
            out[:, l[0]:l[3], l[1]:l[4], l[2]:l[5]] += p
            cnt[l[0]:l[3], l[1]:l[4], l[2]:l[5]] += 1
        with np.errstate(invalid="ignore", divide="ignore"):
            return out / cnt
    raise ValueError(overlap_mode)


def predict_volume(gen, volume, patch_size, patch_overlap=0, batch_size=8, overlap_mode="crop"):
    """patch-wise eval-mode generator over the grid (the loop of src/model.py:314-322) with a torch CPU module"""
    import torch
    locs = grid_locations(volume.shape[1:], patch_size, patch_overlap)
    outs = []
    gen.eval()
    with torch.no_grad():
        for b in range(0, len(locs), batch_size):
            x = torch.from_numpy(extract(volume, locs[b:b + batch_size]))
            outs.append(gen(x).numpy())
    return aggregate(np.concatenate(outs), locs, volume.shape[1:], patch_overlap, overlap_mode)
