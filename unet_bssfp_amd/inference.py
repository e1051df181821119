"""Sliding-window (grid) inference on the GPU -- SURVEY.md 8(f) rank 1.

Mirrors what the reference's ``predict_step`` / ``test_step`` (src/model.py:291-333) use from TorchIO:
``GridSampler(subject, patch_size, patch_overlap=0)`` and ``GridAggregator(sampler, overlap_mode='crop')``
with ``add_batch(batch_tensor, locations)`` / ``get_output_tensor()`` (src/data_module.py:168-183), same
names and argument meaning.  The volume, the patch batches and the aggregated outputs never leave HBM:
patch extraction and aggregation are HIP kernels (``mi355_patch_gather`` / ``mi355_patch_aggregate``).
A *subject* here is ``{name: {'data': Tensor[C, D, H, W]}}`` (what ``unpack_batch`` indexes).

TorchIO is absent from this image, so its behaviour is restated from the published algorithm (see
oracle/grid_ref.py: parity unpinned for that part; 'hann' weighting is not implemented).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, NamedTuple, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib

DATA = "data"          # tio.DATA
LOCATION = "location"  # tio.LOCATION
_MODES = {"crop": 0, "average": 1}
Triple = Union[int, Sequence[int]]


def _triple(v: Triple) -> Tuple[int, int, int]:
    if isinstance(v, (int, np.integer)):
        return (int(v),) * 3
    v = tuple(int(a) for a in v)
    if len(v) != 3:
        raise ValueError(f"expected 3 values, got {v}")
    return v


def grid_locations(spatial_shape: Sequence[int], patch_size: Triple, patch_overlap: Triple = 0) -> np.ndarray:
    """(P, 6) int64 array of (ini, fin) corners, lexicographically sorted: per axis
    range(0, size - patch + 1, patch - overlap) plus a last patch flush with the end."""
    spatial_shape = tuple(int(s) for s in spatial_shape)
    patch_size, patch_overlap = _triple(patch_size), _triple(patch_overlap)
    axes = []
    for size, p, o in zip(spatial_shape, patch_size, patch_overlap):
        if p > size:
            raise ValueError(f"Patch size {patch_size} cannot be larger than image size {spatial_shape}")
        if o >= p:
            raise ValueError(f"Patch overlap {patch_overlap} must be smaller than patch size {patch_size}")
        if o % 2:
            raise ValueError(f"Patch overlap must be a tuple of even integers, not {patch_overlap}")
        idx = list(range(0, size + 1 - p, p - o))
        if idx[-1] != size - p:
            idx.append(size - p)
        axes.append(idx)
    ini = np.array(np.meshgrid(*axes, indexing="ij")).reshape(3, -1).T.astype(np.int64)
    ini = ini[np.lexsort((ini[:, 2], ini[:, 1], ini[:, 0]))]
    return np.hstack([ini, ini + np.array(patch_size, dtype=np.int64)])


def _check_volume(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.Mi355Error(f"{what} must live on the GPU (no CPU fallback)")
    if t.dim() != 4:
        raise ValueError(f"{what} must be (C, D, H, W), got {tuple(t.shape)}")


def extract_patches(volume: torch.Tensor, locations: np.ndarray, patch_size: Tuple[int, int, int]) -> torch.Tensor:
    """volume (C, D, H, W) f32 on the GPU -> (P, C, pd, ph, pw)"""
    _check_volume(volume, "volume")
    volume = volume.float().contiguous()
    c, d, h, w = volume.shape
    n = len(locations)
    out = torch.empty((n, c) + tuple(patch_size), dtype=torch.float32, device=volume.device)
    origins = np.ascontiguousarray(np.asarray(locations)[:, :3], dtype=np.int32)
    _lib.check(_lib.load().mi355_patch_gather(volume.data_ptr(), c, d, h, w, origins.ctypes.data_as(C.c_void_p), n,
                                              *patch_size, out.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "patch_gather")
    return out


class GridSampler:
    """``tio.inference.GridSampler(subject, patch_size, patch_overlap)`` for device-resident subjects."""

    def __init__(self, subject: Dict[str, Dict[str, torch.Tensor]], patch_size: Triple, patch_overlap: Triple = 0):
        shapes = {tuple(img[DATA].shape[1:]) for img in subject.values()}
        if len(shapes) != 1:
            raise RuntimeError(f"images of the subject have different spatial shapes: {shapes}")
        for name, img in subject.items():
            _check_volume(img[DATA], f"subject['{name}']")
        self.subject = subject
        self.spatial_shape = next(iter(shapes))
        self.patch_size, self.patch_overlap = _triple(patch_size), _triple(patch_overlap)
        self.locations = grid_locations(self.spatial_shape, self.patch_size, self.patch_overlap)

    def __len__(self):
        return len(self.locations)

    def _take(self, locs: np.ndarray, squeeze: bool):
        out = {}
        for name, img in self.subject.items():
            p = extract_patches(img[DATA], locs, self.patch_size)
            out[name] = {DATA: p[0] if squeeze else p}
        loc = torch.from_numpy(locs.copy())
        out[LOCATION] = loc[0] if squeeze else loc
        return out

    def __getitem__(self, index: int):
        if not -len(self) <= index < len(self):
            raise IndexError(index)
        index %= len(self)
        return self._take(self.locations[index:index + 1], squeeze=True)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def batches(self, batch_size: int) -> Iterator[dict]:
        """what ``DataLoader(sampler, batch_size)`` yields: (B, C, pd, ph, pw) data, (B, 6) locations"""
        for b in range(0, len(self), batch_size):
            yield self._take(self.locations[b:b + batch_size], squeeze=False)


class GridAggregator:
    """``tio.inference.GridAggregator(sampler, overlap_mode)``: 'crop' (default) or 'average'."""

    def __init__(self, sampler: GridSampler, overlap_mode: str = "crop"):
        if overlap_mode not in _MODES:
            if overlap_mode == "hann":
                raise NotImplementedError("overlap_mode='hann' is not implemented (the reference uses the default 'crop')")
            raise ValueError(f'Overlap mode must be "crop", "average" or "hann" but "{overlap_mode}" was passed')
        self.spatial_shape = tuple(sampler.spatial_shape)
        self.patch_overlap = tuple(sampler.patch_overlap)
        self.overlap_mode = overlap_mode
        self._output_tensor = None
        self._avgmask_tensor = None

    def _locs9(self, locations: np.ndarray) -> np.ndarray:
        ini, fin = locations[:, :3], locations[:, 3:]
        if self.overlap_mode == "crop":
            half = np.array(self.patch_overlap) // 2
            kini = ini + np.where(ini == 0, 0, half)
            kfin = fin - np.where(fin == np.array(self.spatial_shape), 0, half)
        else:
            kini, kfin = ini, fin
        return np.ascontiguousarray(np.hstack([ini, kini, kfin]), dtype=np.int32)

    def add_batch(self, batch_tensor: torch.Tensor, locations) -> None:
        if not batch_tensor.is_cuda:
            raise _lib.Mi355Error("GridAggregator.add_batch runs on the GPU only (no CPU fallback)")
        if batch_tensor.dim() != 5:
            raise ValueError(f"batch_tensor must be (B, C, pd, ph, pw), got {tuple(batch_tensor.shape)}")
        locations = np.asarray(locations.cpu() if isinstance(locations, torch.Tensor) else locations, dtype=np.int64)
        b, c = batch_tensor.shape[:2]
        ps = tuple(batch_tensor.shape[2:])
        if locations.shape != (b, 6) or not (locations[:, 3:] - locations[:, :3] == np.array(ps)).all():
            raise ValueError("locations must be (B, 6) corners matching the patch size of batch_tensor")
        batch = batch_tensor.float().contiguous()
        if self._output_tensor is None:
            self._output_tensor = torch.zeros((c,) + self.spatial_shape, dtype=torch.float32, device=batch.device)
            if self.overlap_mode == "average":
                self._avgmask_tensor = torch.zeros(self.spatial_shape, dtype=torch.float32, device=batch.device)
        elif self._output_tensor.shape[0] != c:
            raise ValueError("channel count changed between batches")
        locs9 = self._locs9(locations)
        cnt = self._avgmask_tensor.data_ptr() if self._avgmask_tensor is not None else None
        _lib.check(_lib.load().mi355_patch_aggregate(batch.data_ptr(), locs9.ctypes.data_as(C.c_void_p), b, *ps,
                                                     _MODES[self.overlap_mode], self._output_tensor.data_ptr(), cnt, c,
                                                     *self.spatial_shape, torch.cuda.current_stream().cuda_stream),
                   "patch_aggregate")

    def get_output_tensor(self) -> torch.Tensor:
        if self._output_tensor is None:
            raise RuntimeError("no batch has been added")
        if self.overlap_mode != "average":
            return self._output_tensor
        out = self._output_tensor.clone()
        c = out.shape[0]
        _lib.check(_lib.load().mi355_patch_average_finalize(out.data_ptr(), self._avgmask_tensor.data_ptr(), c,
                                                            self._avgmask_tensor.numel(),
                                                            torch.cuda.current_stream().cuda_stream), "patch_average_finalize")
        return out


class GridPrediction(NamedTuple):
    """the three aggregated volumes of ``predict_step`` (src/model.py:323-325), named by CONTENT"""
    y_hat: torch.Tensor   # reference variable `in_tensor`   (i_agg receives y_hat, :319)
    y: torch.Tensor       # reference variable `true_tensor` (t_agg receives y, :320)
    x: torch.Tensor       # reference variable `pred_tensor` (o_agg receives x, :321) -- what the reference returns


@torch.no_grad()
def predict_volume(gen: torch.nn.Module, volume: torch.Tensor, patch_size: Triple = 64, patch_overlap: Triple = 0,
                   batch_size: int = 8, overlap_mode: str = "crop") -> torch.Tensor:
    """Whole-volume prediction: eval-mode ``gen`` over the patch grid, aggregated on the device.
    volume (C, D, H, W) -> (C_out, D, H, W)."""
    sampler = GridSampler({"x": {DATA: volume}}, patch_size, patch_overlap)
    agg = GridAggregator(sampler, overlap_mode)
    was_training = gen.training
    gen.eval()
    try:
        for batch in sampler.batches(batch_size):
            agg.add_batch(gen(batch["x"][DATA]), batch[LOCATION])
    finally:
        gen.train(was_training)
    return agg.get_output_tensor()
