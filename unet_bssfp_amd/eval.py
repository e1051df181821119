"""Evaluation-side device ops.

``calc_scalar_maps`` is the GPU counterpart of the voxel loop of ``do_calc_scalar_maps``
(src/eval.py:73-135) with ``do_invert_dwi_tensor_norm`` (src/eval.py:39-47) optionally fused in
front.  The reference works file-to-file through nibabel (absent here); this mirror works on
device tensors -- the NIfTI load/save either side stays with the caller.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib

_DT = {torch.float32: 0, torch.float64: 2}
MAP_NAMES = ("fa", "md", "ad", "rd", "azimuth", "inclination", "rgb")


def calc_scalar_maps(data: torch.Tensor, min_v: Optional[float] = None, max_v: Optional[float] = None,
                     channels_first: bool = False) -> Dict[str, torch.Tensor]:
    """data: (..., 6) like the reference's NIfTI array, or (6, ...) with ``channels_first`` (a generator
    output without its batch axis); f32 or f64 on the GPU.  Returns the seven maps of the reference
    (``rgb`` is (..., 3)), in the dtype of ``data``.  ``min_v``/``max_v`` apply the inverse min-max
    normalisation x * |max - min| + min first."""
    if not data.is_cuda:
        raise _lib.Mi355Error("calc_scalar_maps runs on the GPU only (no CPU fallback)")
    if data.dtype not in _DT:
        raise _lib.Mi355Error(f"calc_scalar_maps: unsupported dtype {data.dtype}")
    if (min_v is None) != (max_v is None):
        raise ValueError("give both min_v and max_v or neither")
    data = data.contiguous()
    if channels_first:
        if data.shape[0] != 6:
            raise ValueError(f"expected 6 tensor components first, got shape {tuple(data.shape)}")
        spatial = tuple(data.shape[1:])
        nvox = data[0].numel()
        cs, vs = nvox, 1
    else:
        if data.shape[-1] != 6:
            raise ValueError(f"expected 6 tensor components last, got shape {tuple(data.shape)}")
        spatial = tuple(data.shape[:-1])
        nvox = data.numel() // 6
        cs, vs = 1, 6
    scale, offset = (1.0, 0.0) if min_v is None else (abs(float(max_v) - float(min_v)), float(min_v))
    out = {k: torch.empty(spatial, dtype=data.dtype, device=data.device) for k in MAP_NAMES[:-1]}
    out["rgb"] = torch.empty(spatial + (3,), dtype=data.dtype, device=data.device)
    _lib.check(_lib.load().mi355_dti_scalar_maps(
        data.data_ptr(), _DT[data.dtype], nvox, cs, vs, scale, offset,
        *[out[k].data_ptr() for k in MAP_NAMES], torch.cuda.current_stream().cuda_stream), "dti_scalar_maps")
    return out
