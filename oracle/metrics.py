"""CPU restatement of the per-step image metrics of the reference's ModelModule (test infrastructure only).

Reference call sites: src/models.py:35-52 (MetricCollection{psnr_metric, ssim_metric}, data_range=(-0.9, 0.9)),
:140-143 (update on the clipped prediction after every training step), :294-296 (evaluation_step), :207-213 (compute / reset
at epoch end).  The arithmetic lives in the un-vendored dependency torchmetrics==1.6.2 (requirements.txt:11), absent from this
image: the functions below restate its published algorithm (functional/image/psnr.py `_psnr_update/_psnr_compute`,
functional/image/ssim.py `_ssim_update`, `_gaussian`) - parity unpinned by reference tests, pinned to this restatement.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _gaussian(kernel_size: int, sigma: float) -> torch.Tensor:
    dist = torch.arange(start=(1 - kernel_size) / 2, end=(1 + kernel_size) / 2, step=1, dtype=torch.float32)
    gauss = torch.exp(-torch.pow(dist / sigma, 2) / 2)
    return (gauss / gauss.sum()).unsqueeze(0)


def psnr_update(preds: torch.Tensor, target: torch.Tensor, data_range=(-0.9, 0.9)):
    """-> (sum_squared_error, n_obs) of the clamped tensors (PeakSignalNoiseRatio.update with a tuple data_range)."""
    p = torch.clamp(preds, min=data_range[0], max=data_range[1]).double()
    t = torch.clamp(target, min=data_range[0], max=data_range[1]).double()
    return ((p - t) ** 2).sum(), target.numel()


def psnr_compute(sum_squared_error, total, data_range=(-0.9, 0.9), base: float = 10.0):
    r = data_range[1] - data_range[0]
    return (2 * math.log(r) - math.log(float(sum_squared_error) / float(total))) * (10 / math.log(base))


def ssim_update(preds: torch.Tensor, target: torch.Tensor, data_range=(-0.9, 0.9), sigma: float = 1.5, k1: float = 0.01,
                k2: float = 0.03) -> torch.Tensor:
    """per-image SSIM [B] (StructuralSimilarityIndexMeasure.update accumulates its sum and the image count)."""
    preds = torch.clamp(preds.float(), min=data_range[0], max=data_range[1])
    target = torch.clamp(target.float(), min=data_range[0], max=data_range[1])
    r = data_range[1] - data_range[0]
    c1, c2 = (k1 * r) ** 2, (k2 * r) ** 2
    channel = preds.size(1)
    ks = int(3.5 * sigma + 0.5) * 2 + 1
    pad = (ks - 1) // 2
    preds = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    target = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    g = _gaussian(ks, sigma)
    kernel = torch.matmul(g.t(), g).expand(channel, 1, ks, ks)
    inputs = torch.cat((preds, target, preds * preds, target * target, preds * target))
    out = F.conv2d(inputs, kernel, groups=channel).split(preds.shape[0])
    mu_pp, mu_tt, mu_pt = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    s_pp = torch.clamp(out[2] - mu_pp, min=0.0)
    s_tt = torch.clamp(out[3] - mu_tt, min=0.0)
    s_pt = out[4] - mu_pt
    full = ((2 * mu_pt + c1) * (2 * s_pt + c2)) / ((mu_pp + mu_tt + c1) * (s_pp + s_tt + c2))
    idx = full[..., pad:-pad, pad:-pad]
    return idx.reshape(idx.shape[0], -1).mean(-1)
