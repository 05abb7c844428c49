"""Per-nucleus mean extractor and CellMetrics.update state, restated (test infrastructure; pinned to tests/golden/comp_cells.npz,
which oracle/make_golden_cells.py captured from the reference's own classes).

Reference: ``MeanCellExtrator.forward / extract_mean`` ``/root/reference/src/utils.py:23-121`` (area down-sampling of the
images, nearest-exact of the label map, per image: ``torch.unique`` of the non-zero labels, ``scatter_add`` sums / counts,
means; images concatenated) and ``CellMetrics.update`` ``/root/reference/src/metrics.py:38-74``.
"""
from __future__ import annotations

import math

import numpy as np


def area_downsample(x: np.ndarray, scale: float) -> np.ndarray:
    """F.interpolate(mode='area') = adaptive average pooling to floor(size * scale): window [floor(i*H/Ho), ceil((i+1)*H/Ho))."""
    H, W = x.shape[-2:]
    Ho, Wo = int(math.floor(H * scale)), int(math.floor(W * scale))
    out = np.zeros(x.shape[:-2] + (Ho, Wo), dtype=np.float64)
    for i in range(Ho):
        y0, y1 = (i * H) // Ho, -((-(i + 1) * H) // Ho)
        for j in range(Wo):
            x0, x1 = (j * W) // Wo, -((-(j + 1) * W) // Wo)
            out[..., i, j] = x[..., y0:y1, x0:x1].mean(axis=(-2, -1))
    return out.astype(np.float32)


def nearest_exact_downsample(lab: np.ndarray, scale: float) -> np.ndarray:
    """F.interpolate(mode='nearest-exact') with a given scale_factor: src = min(floor((i + 0.5) / scale), size - 1)."""
    H, W = lab.shape[-2:]
    Ho, Wo = int(math.floor(H * scale)), int(math.floor(W * scale))
    ys = np.minimum(np.floor((np.arange(Ho) + 0.5) / scale).astype(np.int64), H - 1)
    xs = np.minimum(np.floor((np.arange(Wo) + 0.5) / scale).astype(np.int64), W - 1)
    return lab[..., ys[:, None], xs[None, :]]


def extract_means(pred: np.ndarray, target: np.ndarray | None, nuclei: np.ndarray, scale: float = 1.0, sums: bool = False):
    """-> (pred_means [n, C], target_means [n, C], cell_ids [n], counts [n]); labels ascending per image, images concatenated.
    sums=True returns the per-nucleus sums instead of the means (CellMetrics.update)."""
    if target is None:
        target = np.zeros_like(pred)
    if nuclei.ndim == 4:
        nuclei = nuclei[:, 0]
    if scale < 1.0:
        pred, target = area_downsample(pred, scale), area_downsample(target, scale)
        nuclei = nearest_exact_downsample(nuclei.astype(np.float32), scale).astype(np.int64)   # (through float, as the reference)
    P, T, I, N = [], [], [], []
    C = pred.shape[1]
    for b in range(pred.shape[0]):
        m = nuclei[b] > 0
        if not m.any():
            continue
        u, inv = np.unique(nuclei[b][m], return_inverse=True)
        pf, tf = pred[b][:, m].T.astype(np.float64), target[b][:, m].T.astype(np.float64)
        ps, ts = np.zeros((len(u), C)), np.zeros((len(u), C))
        np.add.at(ps, inv, pf)
        np.add.at(ts, inv, tf)
        cnt = np.bincount(inv, minlength=len(u)).astype(np.float64)
        d = 1.0 if sums else cnt[:, None]
        P.append(ps / d), T.append(ts / d), I.append(u), N.append(cnt)
    if not P:
        z = np.zeros((0, C), np.float32)
        return z, z.copy(), np.zeros(0, np.int64), np.zeros(0)
    return np.concatenate(P).astype(np.float32), np.concatenate(T).astype(np.float32), np.concatenate(I), np.concatenate(N)


def cell_metrics_update(preds: np.ndarray, nuclei: np.ndarray, marker_idxs):
    """per image with nuclei: (ids uint32, sums*255 -> uint32 [n, len(marker_idxs)], areas uint16 [n, 1])"""
    p = (np.clip(preds[:, marker_idxs], -0.9, 0.9).astype(np.float32) + np.float32(0.9)) / np.float32(1.8)
    out = []
    for b in range(p.shape[0]):
        ps, _, ids, cnt = extract_means(p[b:b + 1], None, nuclei[b:b + 1], sums=True)
        out.append(None if len(ids) == 0 else (ids.astype(np.int64), (ps * 255).astype(np.uint32).astype(np.int64), cnt.astype(np.int64)[:, None]))
    return out
