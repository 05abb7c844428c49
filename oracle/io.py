"""TEST INFRASTRUCTURE (CPU oracle): numpy restatement of the input stage either side of the generator.

  normalize_he / normalize_if : ``NormalizationLayer.__call__`` (/root/reference/src/dataset.py:545-575), f32 arithmetic in the
                                reference's operation order (pinned bit for bit to tests/golden/comp_io.npz, which
                                oracle/make_golden_io.py captures from the reference class itself)
  spatial_augment             : the semantics of the reference's spatial albumentations pipeline (dataset.py:458-468:
                                RandomCrop -> HorizontalFlip -> VerticalFlip -> CoarseDropout with one zero-filled hole,
                                applied jointly to image and target) for GIVEN draws.  albumentations is not installed in the
                                build container, so only the operations' definitions are restated (crop = slice, flips =
                                reversed axes, dropout = zero fill of a rectangle in the flipped frame), not its RNG stream:
                                "parity unpinned" for the draw distribution, exact for the pixel operations.
"""
from __future__ import annotations

import numpy as np

HOPTIMUS_MEAN = np.asarray([0.707223, 0.578729, 0.703617]) * 255     # dataset.py:599-601
HOPTIMUS_STD = np.asarray([0.211883, 0.230117, 0.177517]) * 255


def normalize_he(x_u8: np.ndarray, mean=HOPTIMUS_MEAN, std=HOPTIMUS_STD) -> np.ndarray:
    """dataset.py:553-556,570: mean / std reshaped to (1, 1, C) as float32, x_norm = (x - mean) / std."""
    m = np.float32(np.asarray(mean).reshape((1, 1, -1)))
    s = np.float32(np.asarray(std).reshape((1, 1, -1)))
    return (x_u8 - m) / s


def normalize_if(x_u8: np.ndarray) -> np.ndarray:
    """dataset.py:573: np.float32(x) / 255 * 1.8 - 0.9"""
    return np.float32(x_u8) / 255 * 1.8 - 0.9


def spatial_augment(img: np.ndarray, d: dict, crop) -> np.ndarray:
    """img [Hs, Ws, C] uint8; d = draws (oy, ox, hflip, vflip, drop, y1, x1, hh, hw); crop = (H, W)."""
    H, W = crop
    out = img[d["oy"]:d["oy"] + H, d["ox"]:d["ox"] + W].copy()      # A.RandomCrop
    if d["hflip"]:
        out = out[:, ::-1]                                          # A.HorizontalFlip
    if d["vflip"]:
        out = out[::-1]                                             # A.VerticalFlip
    out = np.ascontiguousarray(out)
    if d["drop"]:
        out[d["y1"]:d["y1"] + d["hh"], d["x1"]:d["x1"] + d["hw"]] = 0   # A.CoarseDropout, fill 0 ('image' targets alike)
    return out
