"""Host-side tap tables for mvit_resample2d (index/weight pairs per output coordinate).

They restate the source-index arithmetic of ``F.interpolate`` that the reference relies on:
bilinear x2, align_corners=False (``Fusion_Block.forward``, /root/reference/src/generators/mipheivit.py:89)
and bicubic A=-0.75, align_corners=False, no antialias (``Encoder.forward``, mipheivit.py:147-151,161-162),
plus the transposed (adjoint) tables used by the backward pass.
"""
from __future__ import annotations

import functools

import numpy as np
import torch


def _dense_bilinear(n_in: int, n_out: int) -> np.ndarray:
    R = np.zeros((n_out, n_in), dtype=np.float64)
    scale = n_in / n_out
    for o in range(n_out):
        src = max((o + 0.5) * scale - 0.5, 0.0)
        i0 = min(int(np.floor(src)), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        lam = src - i0
        R[o, i0] += 1.0 - lam
        R[o, i1] += lam
    return R


def _dense_bicubic(n_in: int, n_out: int, A: float = -0.75) -> np.ndarray:
    R = np.zeros((n_out, n_in), dtype=np.float64)
    scale = n_in / n_out
    for o in range(n_out):
        src = (o + 0.5) * scale - 0.5
        i = int(np.floor(src))
        t = src - i
        c = [((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
             ((A + 2) * t - (A + 3)) * t * t + 1,
             ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
             ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A]
        for k in range(4):
            R[o, min(max(i - 1 + k, 0), n_in - 1)] += c[k]
    return R


def _identity(n_in: int, n_out: int) -> np.ndarray:
    assert n_in == n_out
    return np.eye(n_in, dtype=np.float64)


def _dense_nearest(n_in: int, n_out: int) -> np.ndarray:
    """nn.Upsample(scale_factor=n_out/n_in, mode='nearest'): src = min(floor(dst * n_in / n_out), n_in - 1)"""
    R = np.zeros((n_out, n_in), dtype=np.float64)
    scale = np.float32(1.0) / np.float32(n_out / n_in)       # torch keeps the user's scale_factor and inverts it in fp32
    for o in range(n_out):
        R[o, min(int(np.floor(np.float32(o) * scale)), n_in - 1)] = 1.0
    return R


_DENSE = {"bilinear": _dense_bilinear, "bicubic": _dense_bicubic, "identity": _identity, "nearest": _dense_nearest}


@functools.lru_cache(maxsize=None)
def _tables_np(mode: str, n_in: int, n_out: int, adjoint: bool):
    R = _DENSE[mode](n_in, n_out)
    if adjoint:
        R = R.T.copy()  # [n_in, n_out]: gradient rows gather from the forward outputs
    T = int(max(1, (R != 0).sum(1).max()))
    idx = np.zeros((R.shape[0], T), dtype=np.int32)
    wgt = np.zeros((R.shape[0], T), dtype=np.float32)
    for r in range(R.shape[0]):
        nz = np.nonzero(R[r])[0]
        idx[r, :len(nz)] = nz
        wgt[r, :len(nz)] = R[r, nz]
    return idx, wgt


_dev_cache = {}


def taps(mode: str, n_in: int, n_out: int, device, adjoint: bool = False, pad_to: int | None = None):
    """(idx int32 [rows,T], w f32 [rows,T]) on `device`; rows = n_out (forward) or n_in (adjoint)."""
    key = (mode, n_in, n_out, adjoint, pad_to, str(device))
    if key not in _dev_cache:
        idx, wgt = _tables_np(mode, n_in, n_out, adjoint)
        if pad_to is not None and idx.shape[1] < pad_to:
            pad = pad_to - idx.shape[1]
            idx = np.pad(idx, ((0, 0), (0, pad)))
            wgt = np.pad(wgt, ((0, 0), (0, pad)))
        _dev_cache[key] = (torch.from_numpy(idx.copy()).to(device), torch.from_numpy(wgt.copy()).to(device))
    return _dev_cache[key]
