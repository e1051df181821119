"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Per-voxel DTI scalar maps.

Vectorised numpy restatement of the per-voxel arithmetic of ``do_calc_scalar_maps``
(/root/reference/src/eval.py:73-135): symmetric 3x3 diffusion tensor from its upper triangle
(Dxx, Dxy, Dxz, Dyy, Dyz, Dzz) (:85-94) -> ``np.linalg.eigh(., 'U')`` (:95, ascending eigenvalues) ->
AD (:97), RD (:98), MD (:99), FA (:101-103), azimuth / inclination of the principal eigenvector
(:105-112) and the FA-weighted |eigenvector| colour map (:114-116).  The file I/O (nibabel, :74-75,
:118-135) is not part of the arithmetic and is not restated.  float64 like the reference.

The SIGN of an eigenvector is an arbitrary LAPACK choice: azimuth/inclination are defined up to the
antipodal map (az, inc) ~ (az +- 180, 180 - inc); everything else is sign-free.  ``antipodal_close``
is the comparison the tests use for the two angles.

Pinned by tests/golden/dti_scalar_maps.npz, produced by executing the reference's own voxel loop
(AST-extracted from eval.py, see oracle/gen_golden.py) on a small synthetic tensor field.
"""
import numpy as np


def scalar_maps(data: np.ndarray) -> dict:
    """data: (..., 6) -> dict of fa, md, ad, rd, azimuth, inclination (...,) and rgb (..., 3), float64."""
    data = np.asarray(data, dtype=np.float64)
    dxx, dxy, dxz, dyy, dyz, dzz = (data[..., i] for i in range(6))
    m = np.stack([np.stack([dxx, dxy, dxz], -1), np.stack([dxy, dyy, dyz], -1), np.stack([dxz, dyz, dzz], -1)], -2)
    w, v = np.linalg.eigh(m, "U")                                   # ascending; columns are eigenvectors
    ad = w[..., 2]
    rd = (w[..., 0] + w[..., 1]) / 2
    md = w.mean(-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        var = np.sqrt(((w - md[..., None]) ** 2).sum(-1))
        norm = np.sqrt((w ** 2).sum(-1))
        fa = np.sqrt(1.5) * var / norm
        e = v[..., :, 2]                                            # principal eigenvector
        az = 180 / np.pi * np.arctan2(e[..., 1], e[..., 0])
        az = np.where(az > 180, az - 360, az)
        r = np.sqrt((e ** 2).sum(-1))
        inc = 180 / np.pi * np.arccos(e[..., 2] / r)
        rgb = fa[..., None] * np.abs(e)
    return dict(fa=fa, md=md, ad=ad, rd=rd, azimuth=az, inclination=inc, rgb=rgb)


def antipodal_close(az_a, inc_a, az_b, inc_b, atol_deg=1e-3):
    """True where the two directions agree up to the eigenvector sign."""
    def unit(az, inc):
        az, inc = np.deg2rad(az), np.deg2rad(inc)
        return np.stack([np.sin(inc) * np.cos(az), np.sin(inc) * np.sin(az), np.cos(inc)], -1)
    dot = np.abs((unit(az_a, inc_a) * unit(az_b, inc_b)).sum(-1))
    return np.rad2deg(np.arccos(np.clip(dot, -1, 1))) <= atol_deg


def synthetic_tensor_field(shape, seed=0):
    """Diffusion-tensor-like symmetric fields: SPD part + small indefinite perturbation, O(1e-3) scale,
    with a few exactly isotropic / diagonal / zero voxels (the degenerate cases)."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(shape + (3, 3))
    spd = a @ np.swapaxes(a, -1, -2) * 3e-4 + np.eye(3) * 2e-4
    spd += rng.standard_normal(shape + (1, 1)) * 1e-5 * np.eye(3)
    d = np.stack([spd[..., 0, 0], spd[..., 0, 1], spd[..., 0, 2], spd[..., 1, 1], spd[..., 1, 2], spd[..., 2, 2]], -1)
    flat = d.reshape(-1, 6)
    if len(flat) < 4:
        return d
    flat[0] = [1e-3, 0, 0, 1e-3, 0, 1e-3]          # isotropic
    flat[1] = [3e-3, 0, 0, 2e-3, 0, 1e-3]          # diagonal, distinct
    flat[2] = [1e-3, 0, 0, 1e-3, 0, 2e-3]          # diagonal, two equal small eigenvalues
    flat[3] = [0.5, 0.25, 0.1, 0.75, 0.3, 0.9]     # a U[0,1)-like generator output
    return flat.reshape(shape + (6,))


def invert_dwi_tensor_norm(data: np.ndarray, min_v: float, max_v: float) -> np.ndarray:
    """/root/reference/src/eval.py:39-47 without the file I/O."""
    return np.asarray(data, dtype=np.float64) * np.abs(max_v - min_v) + min_v
