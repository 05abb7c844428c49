"""Absolute position-embedding re-grid, restated (test infrastructure).

The reference calls ``timm.layers.resample_abs_pos_embed(posemb, new_size=grid, num_prefix_tokens=0)`` at load time
(``/root/reference/src/generators/foundation_models.py:198-208``).  timm 1.0.15 (not on disk) implements it as: reshape the
[1, N, D] table to [1, D, h, w] in fp32 and ``F.interpolate(size=new_size, mode='bicubic', antialias=True)`` (align_corners False).
This file restates THAT resampler from its published definition (the separable anti-aliased "PIL-style" filter of
``upsample_bicubic2d_aa``): cubic convolution kernel with a = -0.5, support 2 * max(scale, 1), weights normalised per output sample.
It is an independent numpy implementation (no call into torch's interpolate), used to pin the product's loader value by value.
"""
from __future__ import annotations

import numpy as np


def _cubic(x, a=-0.5):
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                    np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def aa_bicubic_matrix(n_in: int, n_out: int) -> np.ndarray:
    """[n_out, n_in] interpolation matrix of the anti-aliased bicubic resize along one axis."""
    scale = n_in / n_out
    support = 2.0 * max(scale, 1.0)
    invscale = 1.0 / max(scale, 1.0)
    R = np.zeros((n_out, n_in))
    for i in range(n_out):
        center = scale * (i + 0.5)
        xmin = max(0, int(center - support + 0.5))
        xsize = min(int(center + support + 0.5), n_in) - xmin
        w = _cubic((np.arange(xsize) + xmin - center + 0.5) * invscale)
        R[i, xmin:xmin + xsize] = w / w.sum()
    return R


def resample_abs_pos_embed(posemb: np.ndarray, old_grid, new_grid) -> np.ndarray:
    """posemb [1, h*w, D] -> [1, H*W, D]"""
    D = posemb.shape[-1]
    p = posemb.astype(np.float64).reshape(old_grid[0], old_grid[1], D)
    Ry, Rx = aa_bicubic_matrix(old_grid[0], new_grid[0]), aa_bicubic_matrix(old_grid[1], new_grid[1])
    out = np.einsum("yh,hwd->ywd", Ry, p)
    out = np.einsum("xw,ywd->yxd", Rx, out)
    return out.reshape(1, new_grid[0] * new_grid[1], D).astype(np.float32)
