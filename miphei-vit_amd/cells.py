"""Validation-time per-nucleus statistics.

``MeanCellExtrator`` keeps the call surface of the reference class (``/root/reference/src/utils.py:16-121``):
``forward(pred, target, nuclei) -> (pred_means, target_means, cell_ids)``, labels of each image ascending (``torch.unique``), images
concatenated, ``scale_factor < 1`` = area down-sampling of the images + nearest-exact of the label map.  ``CellMetrics`` mirrors
the per-batch ``update`` of the reference metric (``/root/reference/src/metrics.py:38-74``: clip to [-0.9, 0.9] -> [0, 1],
per-nucleus sums * 255 as uint32, areas as uint16, ids as uint32, appended per slide); its ``compute`` (logistic regression / AUC
over the slides' nuclei CSV tables) is evaluation code outside the hot path.

Both run on one segmented-reduction pass of the HIP library (``mvit_cell_means``): no dense table sized by the label value, no
per-pixel global atomics.  The only host read is the per-image count of distinct nuclei (the reference's ``torch.unique`` implies
one per image).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def _segmented(pred, target, nuclei, scale_factor, want_sums):
    if nuclei.ndim == 4:
        nuclei = nuclei[:, 0]
    if nuclei.dtype not in (torch.int32, torch.int64):
        nuclei = nuclei.long()
    pred = pred.float().contiguous()
    tgt = target.float().contiguous() if target is not None else None
    rec_count, n_unique, ids, cnt, op, ot = ops.cell_means(pred, tgt, nuclei.contiguous(), scale_factor, want_sums)
    host = torch.stack([rec_count, n_unique]).cpu()              # one small device-to-host read
    if int(host[0].max()) > ids.shape[1]:
        raise RuntimeError(f"cell extractor: {int(host[0].max())} nucleus fragments in one image exceed the scratch capacity "
                           f"({ids.shape[1]}); split the batch into smaller tiles")
    return [int(v) for v in host[1]], ids, cnt, op, ot


class MeanCellExtrator(nn.Module):
    def __init__(self, scale_factor=1.):
        super().__init__()
        if not (0. < scale_factor <= 1):
            raise ValueError("scale_factor should be between 0 and 1")
        self.scale_factor = scale_factor

    def forward(self, pred, target, nuclei):
        B, C = pred.shape[:2]
        n, ids, _, op, ot = _segmented(pred, target, nuclei, self.scale_factor, False)
        if sum(n) == 0:
            z = torch.zeros(0, C, dtype=pred.dtype, device=pred.device)
            return z, z.clone(), torch.empty(0, dtype=torch.long, device=pred.device)
        pm = torch.cat([op[b, :n[b]] for b in range(B)])
        tm = torch.cat([ot[b, :n[b]] for b in range(B)]) if ot is not None else torch.zeros_like(pm)
        cid = torch.cat([ids[b, :n[b]] for b in range(B)]).long()
        return pm.to(pred.dtype), tm.to(pred.dtype), cid


class CellMetrics:
    """State accumulation of the reference's cell-level metric (``update`` only)."""

    def __init__(self, slide_names, marker_names, min_area=20):
        excluded = ["Hoechst", "Dapi"]
        kept = [(i, n) for i, n in enumerate(marker_names) if n not in excluded]
        self.marker_names = [n for _, n in kept]
        self.marker_idxs = [i for i, _ in kept]
        self.min_area = min_area
        self.slide_names = list(slide_names)
        self.state = {s: {"cell_id": [], "sum": [], "area": []} for s in self.slide_names}

    def update(self, preds, nuclei_masks, slide_names):
        p = torch.clip(preds[:, self.marker_idxs], -0.9, 0.9).float()
        p = (p + 0.9) / 1.8
        n, ids, cnt, sums, _ = _segmented(p, None, nuclei_masks, 1.0, True)
        for b, name in enumerate(slide_names):
            if n[b] == 0:
                continue
            st = self.state[name]
            st["cell_id"].append(ids[b, :n[b]].to(torch.int64).cpu())                   # (uint32 in the reference)
            st["sum"].append((sums[b, :n[b]] * 255).to(torch.int64).cpu())
            st["area"].append(cnt[b, :n[b]].to(torch.int64).cpu().unsqueeze(-1))

    def reset(self):
        for st in self.state.values():
            for v in st.values():
                v.clear()

    def compute(self, *a, **k):
        raise NotImplementedError("CellMetrics.compute reads the slides' nuclei tables and fits per-marker classifiers: "
                                  "evaluation code outside the MI355X hot path (SURVEY.md section 2)")
