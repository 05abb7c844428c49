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


RMAX_STEPS = (8192, 16384)      # partial-record capacity per image (one record per (16x128 tile, nucleus) pair)


def _run(pred, tgt, nuclei, scale_factor, want_sums, rmax):
    rec_count, n_unique, ids, cnt, op, ot = ops.cell_means(pred, tgt, nuclei, scale_factor, want_sums, rmax=rmax)
    host = torch.stack([rec_count, n_unique]).cpu()              # one small device-to-host read
    return int(host[0].max()), [int(v) for v in host[1]], ids, cnt, op, ot


def _segmented(pred, target, nuclei, scale_factor, want_sums):
    """Per image: (number of distinct nuclei, ascending ids, pixel counts, pred means|sums, target means|sums).

    The reference (``torch.unique`` + ``scatter_add_``) has no capacity limit; the segmented kernel keeps one partial record per
    (tile, nucleus) pair in a scratch of ``rmax`` records per image.  Dense label maps that overflow it are retried with the
    larger capacity and, beyond that, on row chunks of the images whose per-nucleus partial sums are merged afterwards."""
    if nuclei.ndim == 4:
        nuclei = nuclei[:, 0]
    if nuclei.dtype not in (torch.int32, torch.int64):
        nuclei = nuclei.long()
    if nuclei.dtype == torch.int64 and nuclei.numel() and int(nuclei.max()) > 0x7fffffff:
        raise ValueError("cell extractor: nucleus ids above INT32_MAX are not supported (slide-global ids fit int32)")
    pred = pred.float().contiguous()
    tgt = target.float().contiguous() if target is not None else None
    nuclei = nuclei.contiguous()
    worst = 0
    for rmax in RMAX_STEPS:
        worst, n, ids, cnt, op, ot = _run(pred, tgt, nuclei, scale_factor, want_sums, rmax)
        if worst <= rmax:
            return n, ids, cnt, op, ot
    return _segmented_chunked(pred, tgt, nuclei, scale_factor, want_sums, worst)


def _segmented_chunked(pred, tgt, nuclei, scale_factor, want_sums, worst):
    """Row chunks of every image as separate images (sums + counts), merged per nucleus id: only for label maps with more
    (tile, nucleus) fragments than the largest scratch holds."""
    B, C, H, W = pred.shape
    step = 64                                                    # source rows per chunk unit (4 tile rows at scale 1)
    if abs(step * scale_factor - round(step * scale_factor)) > 1e-6 or H % step:
        raise RuntimeError(f"cell extractor: {worst} nucleus fragments in one image exceed the scratch capacity "
                           f"({RMAX_STEPS[-1]}) and H={H}, scale_factor={scale_factor} cannot be cut into aligned row chunks")
    k = min(H // step, max(2, 2 * -(-worst // RMAX_STEPS[-1])))
    rows = -(-(H // step) // k) * step
    parts = []
    for r0 in range(0, H, rows):
        r1 = min(H, r0 + rows)
        w_, n, ids, cnt, op, ot = _run(pred[:, :, r0:r1].contiguous(), tgt[:, :, r0:r1].contiguous() if tgt is not None else None,
                                       nuclei[:, r0:r1].contiguous(), scale_factor, True, RMAX_STEPS[-1])
        if w_ > RMAX_STEPS[-1]:
            raise RuntimeError(f"cell extractor: {w_} nucleus fragments in a {r1 - r0}-row chunk exceed the scratch capacity")
        parts.append((n, ids, cnt, op, ot))
    dev = pred.device
    cap = max(sum(p[0][b] for p in parts) for b in range(B)) or 1
    n_out = []
    ids_o = torch.zeros(B, cap, device=dev, dtype=torch.int32)
    cnt_o = torch.zeros(B, cap, device=dev, dtype=torch.float32)
    op_o = torch.zeros(B, cap, C, device=dev, dtype=torch.float32)
    ot_o = torch.zeros(B, cap, C, device=dev, dtype=torch.float32) if tgt is not None else None
    for b in range(B):
        cid = torch.cat([p[1][b, :p[0][b]] for p in parts]).long()
        if cid.numel() == 0:
            n_out.append(0)
            continue
        u, inv = torch.unique(cid, return_inverse=True)            # ascending, as the single-pass kernel emits them
        m = u.numel()
        n_out.append(m)
        ids_o[b, :m] = u.to(torch.int32)
        cnt_o[b, :m].index_add_(0, inv, torch.cat([p[2][b, :p[0][b]] for p in parts]))
        op_o[b, :m].index_add_(0, inv, torch.cat([p[3][b, :p[0][b]] for p in parts]))
        if ot_o is not None:
            ot_o[b, :m].index_add_(0, inv, torch.cat([p[4][b, :p[0][b]] for p in parts]))
        if not want_sums:
            op_o[b, :m] /= cnt_o[b, :m, None]
            if ot_o is not None:
                ot_o[b, :m] /= cnt_o[b, :m, None]
    return n_out, ids_o, cnt_o, op_o, ot_o


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
        self.marker_cols = [f"{n}_pos" for n in self.marker_names]          # reference metrics.py:21-22 (ModelModule sizes its
        self.marker_pred_cols = [f"{n}_pred" for n in self.marker_names]    # logreg_layer from them, models.py:57-59)
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
