"""Reconstruction loss of the MIPHEI-ViT default configuration (reference ``src/loss.py:47-57``)."""
from __future__ import annotations

import torch
import torch.nn as nn


class WeightedMSELoss(nn.Module):
    """mean_c( mean_{b,h,w}((y_pred - y_true)^2)_c * w_c ) * lambda_factor.

    ``forward`` is differentiable torch math for generic callers; ``ModelModule.training_step`` uses the fused
    HIP kernel (``mvit_wmse_fwd_bwd``) that produces the loss value and dL/dy_pred in one pass.
    """

    def __init__(self, lambda_factor, marker_weights):
        super().__init__()
        self.lambda_factor = lambda_factor
        self.register_buffer("marker_weights", torch.as_tensor(marker_weights, dtype=torch.float32))

    def forward(self, y_true, y_pred):
        loss = (y_pred - y_true) ** 2
        loss = loss.mean(dim=(0, 2, 3)) * self.marker_weights
        return loss.mean() * self.lambda_factor


def marker_weights_from_stats(stds):
    """weights = (1/std) / min(1/std)   (reference ``src/train.py:137-140``)."""
    inv = 1.0 / torch.as_tensor(stds, dtype=torch.float64)
    return (inv / inv.min()).float()


def marker_weights_from_file(path, channel_names):
    """Per-marker weights from a ``channel_stats.json``-shaped file ({name: {"std": ...}}), as ``src/train.py:137-142``
    builds them.  A missing marker is a ``KeyError`` naming it (the reference fails the same way)."""
    import json
    with open(path) as f:
        stats = json.load(f)
    return marker_weights_from_stats([stats[n]["std"] for n in channel_names])
