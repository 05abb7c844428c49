"""On-device input / output stage.

The reference normalises on the CPU inside DataLoader workers (``NormalizationLayer``,
``/root/reference/src/dataset.py:545-575``; H-Optimus-0 mean/std at ``:599-601``) and converts predictions to uint8 on
the host side of ``SavePredictionsCallback`` (``/root/reference/src/callbacks.py:345-346``).  At ~1e3 tiles/s per GPU that
becomes the bottleneck, so both ends run as HBM-bound HIP kernels on raw uint8 tiles.
"""
from __future__ import annotations

import torch

from . import ops

HOPTIMUS_MEAN = (0.707223 * 255, 0.578729 * 255, 0.703617 * 255)
HOPTIMUS_STD = (0.211883 * 255, 0.230117 * 255, 0.177517 * 255)


class InputStage:
    """uint8 RGB [B,H,W,3] -> f32 NCHW (x - mean) / std; uint8 mIF [B,H,W,C] -> f32 NCHW x/255*1.8 - 0.9, with the reference's own
    f32 operation order (bit-identical to ``NormalizationLayer``; tests/golden/comp_io.npz)."""

    def __init__(self, device, mean=HOPTIMUS_MEAN, std=HOPTIMUS_STD):
        self.device = device
        self.mean = [float(torch.tensor(v, dtype=torch.float64).float()) for v in mean]
        self.std = [float(torch.tensor(v, dtype=torch.float64).float()) for v in std]

    def _run(self, rgb_u8, mif_u8):
        ref = rgb_u8 if rgb_u8 is not None else mif_u8
        B, H, W, _ = ref.shape
        if W % 4:
            raise ValueError("tile width must be a multiple of 4")
        img = torch.empty(B, 3, H, W, device=self.device, dtype=torch.float32) if rgb_u8 is not None else None
        tgt = torch.empty(B, mif_u8.shape[3], H, W, device=self.device, dtype=torch.float32) if mif_u8 is not None else None
        ops.augment_tiles(rgb_u8.contiguous() if rgb_u8 is not None else None, mif_u8.contiguous() if mif_u8 is not None else None,
                          img, tgt, None, (H, W), 0, 0, self.mean, self.std, p_hflip=0.0, p_vflip=0.0, p_drop=0.0, hole_frac=0.0)
        return img, tgt

    def image(self, rgb_u8):
        if rgb_u8.shape[3] != 3:
            raise ValueError("expected uint8 RGB tiles [B,H,W,3]")
        return self._run(rgb_u8, None)[0]

    def target(self, mif_u8):
        return self._run(None, mif_u8)[1]


class TrainAugmenter:
    """Training-time input stage on the device: the spatial half of the reference's albumentations pipeline
    (``get_augmentations(training=True)``, ``/root/reference/src/dataset.py:458-468``: ``RandomCrop`` -> ``HorizontalFlip(0.5)`` ->
    ``VerticalFlip(0.5)`` -> ``CoarseDropout(p=0.1, one hole, <= 30 % of each side)``, applied jointly to image and target) fused
    with both ``NormalizationLayer`` modes, on raw uint8 tiles.  At ~430 tiles/s per GPU a CPU loader (``dataset.py:43,118-133``)
    cannot feed one GPU, let alone eight; here a batch costs one HBM-bound launch.

    Draws are counter-based -- sample ``n`` of the run (``n = step * batch + b``, plus ``rank_offset``) always gets the same
    crop / flips / hole for a given seed, independent of batch size and rank layout -- and ``params(n)`` recomputes them on the
    host.  (The colour augmentations of the reference -- HED jitter, brightness / contrast, blur, noise -- are not part of this
    stage.)  ``__call__`` returns the batch dict ``training_step`` takes; ``image_nhwc8`` is the engine's bf16 NHWC decoder image
    buffer, written by the same launch (the engine then skips its own NCHW -> NHWC conversion)."""

    def __init__(self, device, crop, seed=0, mean=HOPTIMUS_MEAN, std=HOPTIMUS_STD, p_hflip=0.5, p_vflip=0.5, p_drop=0.1,
                 hole_frac=0.3, rank_offset=0):
        self.device, self.crop, self.seed = device, (int(crop[0]), int(crop[1])), int(seed)
        # f32 constants exactly as NormalizationLayer builds them (np.float32 of the f64 mean / std)
        self.mean = [float(torch.tensor(v, dtype=torch.float64).float()) for v in mean]
        self.std = [float(torch.tensor(v, dtype=torch.float64).float()) for v in std]
        self.p = dict(p_hflip=p_hflip, p_vflip=p_vflip, p_drop=p_drop, hole_frac=hole_frac)
        self.rank_offset = int(rank_offset)

    def params(self, sample, src_size):
        return ops.augment_draw(src_size[0], src_size[1], self.crop[0], self.crop[1], self.seed, self.rank_offset + sample, **self.p)

    def __call__(self, rgb_u8, mif_u8, sample0, nhwc8=True):
        B = rgb_u8.shape[0]
        H, W = self.crop
        img = torch.empty(B, 3, H, W, device=self.device, dtype=torch.float32)
        tgt = torch.empty(B, mif_u8.shape[3], H, W, device=self.device, dtype=torch.float32) if mif_u8 is not None else None
        n8 = torch.empty(B, H, W, 8, device=self.device, dtype=torch.bfloat16) if nhwc8 else None
        ops.augment_tiles(rgb_u8.contiguous(), mif_u8.contiguous() if mif_u8 is not None else None, img, tgt, n8, self.crop,
                          self.seed, self.rank_offset + int(sample0), self.mean, self.std, **self.p)
        batch = {"image": img, "target": tgt}
        if n8 is not None:
            batch["image_nhwc8"] = n8
        return batch


def export_uint8(pred):
    """f32 predictions [B,C,H,W] -> uint8, ((y+0.9)/1.8).clamp(0,1)*255 (truncated)."""
    out = torch.empty(pred.shape, device=pred.device, dtype=torch.uint8)
    return ops.f32_to_u8_export(pred.contiguous(), out)


_PERM_CACHE: dict = {}


def shuffled_indices(n0: int, count: int, n_tiles: int, seed: int = 0, device=None) -> torch.Tensor:
    """Tile indices of the global samples n0 .. n0 + count - 1 of a shuffled, endlessly repeated pass over `n_tiles` tiles.

    The reference's training DataLoader draws a fresh permutation per epoch (``shuffle=True, drop_last=True``,
    ``/root/reference/src/dataset.py:117-121``).  Here global sample n belongs to epoch n // n_tiles and takes entry n % n_tiles of
    that epoch's permutation, drawn from a CPU generator seeded by (seed, epoch): a pure function of (n, seed), so every rank
    -- whatever the rank layout -- agrees on it without an exchange and a resumed run continues the same stream.  (The stream is
    continuous: an epoch's tail is not dropped, a batch may straddle two epochs.)"""
    if n_tiles <= 0 or count < 0:
        raise ValueError("shuffled_indices: need n_tiles > 0 and count >= 0")
    dev = torch.device("cpu") if device is None else torch.device(device)
    parts = []
    i = 0
    while i < count:
        n = n0 + i
        epoch, pos = divmod(n, n_tiles)
        key = (int(seed), int(epoch), int(n_tiles), str(dev))
        perm = _PERM_CACHE.get(key)
        if perm is None:
            g = torch.Generator(device="cpu")
            g.manual_seed((int(seed) * 1000003 + int(epoch) * 7919 + 12345) & 0x7FFFFFFFFFFFFFFF)
            # an epoch's permutation moves to `device` ONCE; a step then slices it there (no pageable host-to-device copy inside
            # the step: such a copy waits for the whole stream and leaves the step's prologue launch-bound)
            perm = torch.randperm(n_tiles, generator=g).to(dev)
            if len(_PERM_CACHE) > 4:
                _PERM_CACHE.clear()
            _PERM_CACHE[key] = perm
        take = min(count - i, n_tiles - pos)
        parts.append(perm[pos:pos + take])
        i += take
    if not parts:
        return torch.empty(0, dtype=torch.int64, device=dev)
    return parts[0] if len(parts) == 1 else torch.cat(parts)
