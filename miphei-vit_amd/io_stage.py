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
    """uint8 RGB [B,H,W,3] -> f32 NCHW (x - mean) / std; uint8 mIF [B,H,W,C] -> f32 NCHW x/255*1.8 - 0.9."""

    def __init__(self, device, mean=HOPTIMUS_MEAN, std=HOPTIMUS_STD):
        m = torch.tensor(mean, dtype=torch.float64)
        s = torch.tensor(std, dtype=torch.float64)
        self.scale = (1.0 / s).float().to(device)
        self.shift = (-m / s).float().to(device)
        self.device = device

    def image(self, rgb_u8):
        B, H, W, C = rgb_u8.shape
        out = torch.empty(B, C, H, W, device=self.device, dtype=torch.float32)
        return ops.u8_nhwc_to_f32_nchw(rgb_u8.contiguous(), out, self.scale, self.shift)

    def target(self, mif_u8):
        B, H, W, C = mif_u8.shape
        out = torch.empty(B, C, H, W, device=self.device, dtype=torch.float32)
        scale = torch.full((C,), 1.8 / 255.0, device=self.device)
        shift = torch.full((C,), -0.9, device=self.device)
        return ops.u8_nhwc_to_f32_nchw(mif_u8.contiguous(), out, scale, shift)


def export_uint8(pred):
    """f32 predictions [B,C,H,W] -> uint8, ((y+0.9)/1.8).clamp(0,1)*255 (truncated)."""
    out = torch.empty(pred.shape, device=pred.device, dtype=torch.uint8)
    return ops.f32_to_u8_export(pred.contiguous(), out)
