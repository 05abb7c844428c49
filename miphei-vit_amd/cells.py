"""Validation-time per-nucleus mean extractor.

Same call surface as the reference's ``MeanCellExtrator`` (``/root/reference/src/utils.py:16-121``) for the
``scale_factor == 1`` configuration: ``forward(pred, target, nuclei) -> (pred_means, target_means, cell_ids)`` with the
labels of each image in ascending order (``torch.unique``) and images concatenated.  The per-label sums / counts come
from one HIP pass over the label map (``mvit_cell_sums``); the compaction of non-empty labels is a host-visible
``nonzero`` exactly like the reference's ``unique``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class MeanCellExtrator(nn.Module):
    def __init__(self, scale_factor=1.):
        super().__init__()
        if not (0. < scale_factor <= 1):
            raise ValueError("scale_factor should be between 0 and 1")
        if scale_factor != 1.:
            raise NotImplementedError("only scale_factor == 1 runs on the HIP path")
        self.scale_factor = scale_factor

    def forward(self, pred, target, nuclei):
        if nuclei.ndim == 4:
            nuclei = nuclei[:, 0]
        B, C, H, W = pred.shape
        pred = pred.float().contiguous()
        tgt = target.float().contiguous() if target is not None else None
        lab = nuclei.to(torch.int32).contiguous()
        L = int(lab.max())  # one host read, as torch.unique implies in the reference
        if L <= 0:
            z = torch.zeros(0, C, dtype=pred.dtype, device=pred.device)
            return z, z.clone(), torch.empty(0, dtype=nuclei.dtype, device=pred.device)
        sums_p = torch.zeros(B, L + 1, C, device=pred.device)
        sums_t = torch.zeros(B, L + 1, C, device=pred.device)
        counts = torch.zeros(B, L + 1, device=pred.device)
        ops.cell_sums(pred, tgt, lab, sums_p, sums_t, counts, L)
        idx = counts.nonzero(as_tuple=False)  # rows sorted by (image, label): the reference's per-image unique() order
        cnt = counts[idx[:, 0], idx[:, 1]].unsqueeze(1)
        pm = sums_p[idx[:, 0], idx[:, 1]] / cnt
        tm = sums_t[idx[:, 0], idx[:, 1]] / cnt
        return pm, tm, idx[:, 1].to(nuclei.dtype)
