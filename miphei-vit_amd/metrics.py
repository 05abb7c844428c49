"""Per-step image metrics of ModelModule (reference src/models.py:35-52,140-143,207-213): the torchmetrics
MetricCollection {psnr_metric: PeakSignalNoiseRatio, ssim_metric: StructuralSimilarityIndexMeasure}, both with
data_range=(-0.9, 0.9), with the state kept on the device and updated by one HIP entry point (no host sync)."""
from __future__ import annotations

import math

import torch

from . import ops


class PixMetrics:
    """`update(preds, target)` / `compute()` / `reset()` / `clone(prefix=)` of the reference's pixel MetricCollection."""

    def __init__(self, data_range=(-0.9, 0.9), prefix: str = ""):
        self.data_range = (float(data_range[0]), float(data_range[1]))
        self.prefix = prefix
        self._state = None      # [sum_squared_error, n_obs, sum of per-image SSIM, n_images] (f64, device)
        self._scratch = None

    def clone(self, prefix: str = ""):
        return PixMetrics(self.data_range, prefix)

    def _ensure(self, device, B):
        if self._state is None or self._state.device != device:
            self._state = torch.zeros(4, device=device, dtype=torch.float64)
            self._scratch = None
        need = ops.pix_metrics_scratch_doubles(B)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.zeros(need, device=device, dtype=torch.float64)

    @torch.no_grad()
    def update(self, preds: torch.Tensor, target: torch.Tensor):
        if preds.shape != target.shape or preds.dim() != 4:
            raise ValueError("expected [B, C, H, W] predictions and targets of one shape")
        if not preds.is_cuda:
            raise RuntimeError("PixMetrics runs on the ROCm device (no CPU fallback)")
        p = preds.detach().to(torch.float32).contiguous()
        t = target.detach().to(device=p.device, dtype=torch.float32).contiguous()
        B, C, H, W = p.shape
        self._ensure(p.device, B)
        ops.pix_metrics_update(p, t, self._state, self._scratch, B, C, H, W, *self.data_range)

    def compute(self) -> dict:
        """{prefix+'psnr_metric', prefix+'ssim_metric'} as Python floats (the one host read, at epoch end)."""
        if self._state is None:
            return {}
        sse, n, ssim, ni = (float(v) for v in self._state.cpu())
        r = self.data_range[1] - self.data_range[0]
        out = {}
        if n > 0:
            out[self.prefix + "psnr_metric"] = 10.0 * math.log10(r * r / (sse / n)) if sse > 0 else float("inf")
        if ni > 0:
            out[self.prefix + "ssim_metric"] = ssim / ni
        return out

    def reset(self):
        if self._state is not None:
            self._state.zero_()
