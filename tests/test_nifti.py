"""NIfTI-1 writer/reader (data-format part): format constants of nifti1.h + round trips."""
import gzip
import struct

import numpy as np
import pytest

from unet_bssfp_amd import nifti


@pytest.mark.parametrize("dtype,code,bitpix", [(np.float32, 16, 32), (np.float64, 64, 64), (np.int16, 4, 16), (np.uint8, 2, 8)])
def test_header_constants_and_roundtrip(tmp_path, dtype, code, bitpix):
    rng = np.random.default_rng(0)
    a = (rng.random((5, 6, 7, 3)) * 100).astype(dtype)
    p = tmp_path / "v.nii.gz"
    nifti.save(a, p, np.eye(4))
    raw = gzip.open(p, "rb").read()
    assert len(raw) == 352 + a.nbytes
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:348] == b"n+1\x00"
    assert struct.unpack_from("<8h", raw, 40) == (4, 5, 6, 7, 3, 1, 1, 1)
    assert struct.unpack_from("<2h", raw, 70) == (code, 0) or struct.unpack_from("<h", raw, 70)[0] == code
    assert struct.unpack_from("<h", raw, 72)[0] == bitpix
    assert struct.unpack_from("<f", raw, 108)[0] == 352.0
    assert struct.unpack_from("<2h", raw, 252) == (0, 2)                           # qform unknown, sform aligned
    assert struct.unpack_from("<4f", raw, 280) == (1.0, 0.0, 0.0, 0.0)
    # voxel (i, j, k, c) sits at Fortran offset: first axis fastest
    first = np.frombuffer(raw, dtype=np.dtype(dtype).newbyteorder("<"), count=5, offset=352)
    np.testing.assert_array_equal(first, a[:, 0, 0, 0])
    b, aff = nifti.load(p)
    assert b.dtype == a.dtype and b.shape == a.shape
    np.testing.assert_array_equal(b, a)
    np.testing.assert_array_equal(aff, np.eye(4))


def test_affine_zooms_uncompressed_and_errors(tmp_path):
    aff = np.array([[2.0, 0, 0, -10], [0, 1.5, 0, 4], [0, 0, 3.0, 7], [0, 0, 0, 1]])
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    nifti.save(a, tmp_path / "v.nii", aff)
    raw = open(tmp_path / "v.nii", "rb").read()
    assert struct.unpack_from("<8f", raw, 76)[1:4] == (2.0, 1.5, 3.0)
    b, aff2 = nifti.load(tmp_path / "v.nii")
    np.testing.assert_array_equal(b, a)
    np.testing.assert_allclose(aff2, aff)
    with pytest.raises(TypeError):
        nifti.save(a.astype(np.complex64), tmp_path / "c.nii")
    (tmp_path / "bad.nii").write_bytes(b"\x00" * 400)
    with pytest.raises(ValueError):
        nifti.load(tmp_path / "bad.nii")


def test_save_predictions_names_and_layout(tmp_path):
    import torch
    x, y, y_hat = torch.rand(1, 24, 8, 8, 8), torch.rand(1, 6, 8, 8, 8), torch.rand(1, 6, 8, 8, 8)
    names = nifti.save_predictions(x, y, y_hat, 3, "bssfp", ("01", "2"), directory=str(tmp_path))
    assert [n.split("/")[-1] for n in names] == ["input-3_mod-bssfp_sub-01_ses-2.nii.gz", "pred-3_mod-bssfp_sub-01_ses-2.nii.gz",
                                                "target-3_mod-bssfp_sub-01_ses-2.nii.gz"]
    arr, _ = nifti.load(names[1])
    assert arr.shape == (8, 8, 8, 6)                                                # channels last, as the reference writes
    np.testing.assert_array_equal(arr, np.moveaxis(y_hat.numpy().squeeze(), 0, -1))
