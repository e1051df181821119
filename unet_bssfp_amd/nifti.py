"""Minimal NIfTI-1 (.nii / .nii.gz) writer and reader -- SURVEY.md 8(f) rank 4 (data-format part).

The reference saves predictions with ``nib.save(nib.Nifti1Image(array, np.eye(4)), name)``
(src/model.py:335-357) and reads volumes through nibabel everywhere else; nibabel is absent from this image.
This module writes the single-file NIfTI-1 layout that call produces (348-byte header, 4 bytes of extension
flags, data at offset 352, Fortran order like nibabel) and reads files of the same family back.  Host-side
I/O only -- no GPU code.  Written against the published NIfTI-1 header definition (nifti1.h); with no nibabel
here to cross-check, parity is pinned by the format constants in the tests (**parity unpinned** beyond them).
"""
from __future__ import annotations

import gzip
import struct
from typing import Tuple

import numpy as np

_DTYPES = {np.dtype("uint8"): (2, 8), np.dtype("int16"): (4, 16), np.dtype("int32"): (8, 32),
           np.dtype("float32"): (16, 32), np.dtype("float64"): (64, 64), np.dtype("int8"): (256, 8),
           np.dtype("uint16"): (512, 16), np.dtype("uint32"): (768, 32), np.dtype("int64"): (1024, 64),
           np.dtype("uint64"): (1280, 64)}
_CODES = {code: dt for dt, (code, _) in _DTYPES.items()}


def _header(shape, dtype, affine) -> bytes:
    if len(shape) > 7:
        raise ValueError("NIfTI-1 stores at most 7 dimensions")
    code, bitpix = _DTYPES[np.dtype(dtype)]
    dim = [len(shape)] + list(shape) + [1] * (7 - len(shape))
    zooms = np.sqrt((np.asarray(affine, dtype=np.float64)[:3, :3] ** 2).sum(0))
    pixdim = [1.0] + [float(z) for z in zooms] + [1.0] * 4
    for i in range(4, len(shape) + 1):
        pixdim[i] = 1.0
    h = bytearray(348)
    struct.pack_into("<i", h, 0, 348)                      # sizeof_hdr
    struct.pack_into("<8h", h, 40, *dim)                   # dim[8]
    struct.pack_into("<h", h, 70, code)                    # datatype
    struct.pack_into("<h", h, 72, bitpix)                  # bitpix
    struct.pack_into("<8f", h, 76, *pixdim)                # pixdim[8] (pixdim[0] = qfac)
    struct.pack_into("<f", h, 108, 352.0)                  # vox_offset
    struct.pack_into("<f", h, 112, 1.0)                    # scl_slope
    struct.pack_into("<f", h, 116, 0.0)                    # scl_inter
    struct.pack_into("<B", h, 123, 10)                     # xyzt_units: mm + seconds, nibabel's default
    struct.pack_into("<h", h, 252, 0)                      # qform_code: unknown
    struct.pack_into("<h", h, 254, 2)                      # sform_code: aligned (an affine was given)
    a = np.asarray(affine, dtype=np.float32)
    struct.pack_into("<4f", h, 280, *a[0]); struct.pack_into("<4f", h, 296, *a[1]); struct.pack_into("<4f", h, 312, *a[2])
    h[344:348] = b"n+1\x00"                                # magic: single file
    return bytes(h)


def save(array: np.ndarray, path: str, affine=None) -> None:
    """``nib.save(nib.Nifti1Image(array, affine), path)``; gzip when the name ends in .gz."""
    array = np.asarray(array)
    if array.dtype not in _DTYPES:
        raise TypeError(f"dtype {array.dtype} has no NIfTI-1 code")
    affine = np.eye(4) if affine is None else np.asarray(affine)
    blob = _header(array.shape, array.dtype, affine) + b"\x00" * 4 + array.astype(array.dtype.newbyteorder("<")).tobytes(order="F")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as fh:
        fh.write(blob)


def load(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (array, affine).  scl_slope / scl_inter are applied like nibabel's ``get_fdata``."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as fh:
        raw = fh.read()
    for end in ("<", ">"):
        if struct.unpack_from(end + "i", raw, 0)[0] == 348:
            break
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file (sizeof_hdr != 348)")
    if raw[344:347] not in (b"n+1", b"ni1"):
        raise ValueError(f"{path}: bad magic {raw[344:348]!r}")
    if raw[344:347] == b"ni1":
        raise NotImplementedError("header/image pairs (.hdr/.img) are not read")
    dim = struct.unpack_from(end + "8h", raw, 40)
    shape = tuple(dim[1:1 + dim[0]])
    code = struct.unpack_from(end + "h", raw, 70)[0]
    if code not in _CODES:
        raise NotImplementedError(f"datatype code {code}")
    off = int(struct.unpack_from(end + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(end + "2f", raw, 112)
    dt = _CODES[code].newbyteorder(end)
    data = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F")
    if slope not in (0.0, 1.0) or inter != 0.0:
        if np.isfinite(slope) and slope != 0.0:
            data = data.astype(np.float64) * slope + inter
    sform_code = struct.unpack_from(end + "h", raw, 254)[0]
    affine = np.eye(4)
    if sform_code > 0:
        affine[0] = struct.unpack_from(end + "4f", raw, 280)
        affine[1] = struct.unpack_from(end + "4f", raw, 296)
        affine[2] = struct.unpack_from(end + "4f", raw, 312)
    else:
        pix = struct.unpack_from(end + "8f", raw, 76)
        affine[:3, :3] = np.diag(pix[1:4])
    return np.ascontiguousarray(data), affine


def save_predictions(x, y, y_hat, batch_idx: int, input_modality: str, sub_ses=("unknown", "unknown"), stamp: str = "",
                     directory: str = "."):
    """The three files of ``save_predicitions`` (src/model.py:335-357): channels moved last, identity affine,
    names ``{input,pred,target}-{idx}_mod-{modality}{stamp}_sub-{sub}_ses-{ses}.nii.gz``.  Tensors may live on the GPU."""
    import os
    out = []
    for tag, t in (("input", x), ("pred", y_hat), ("target", y)):
        arr = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        arr = np.moveaxis(arr.squeeze(), 0, -1)
        name = os.path.join(directory, f"{tag}-{batch_idx}_mod-{input_modality}{stamp}_sub-{sub_ses[0]}_ses-{sub_ses[1]}.nii.gz")
        save(arr, name, np.eye(4))
        out.append(name)
    return out
