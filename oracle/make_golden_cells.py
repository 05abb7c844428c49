"""Fixtures for the validation-time cell extractor from the REFERENCE's own classes (build container only; import machinery of
make_golden.py):
  comp_cells.npz : MeanCellExtrator(scale_factor = 1 and 0.5).forward on a label map with slide-global (sparse, large) nucleus
                   ids, an image without nuclei and background gaps (src/utils.py:16-121), and the per-batch state the
                   reference's CellMetrics.update appends (src/metrics.py:38-74: clip -> [0,1], per-nucleus sums * 255 as
                   uint32, areas as uint16, ids as uint32) for two slides.
The label map is drawn here (numpy RNG) and stored in the fixture; predictions / targets come from oracle.detgen.
Usage:  python oracle/make_golden_cells.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.detgen import det_normal  # noqa: E402
from oracle.make_golden import REF, _exec, load_reference  # noqa: E402


def label_map(rng, B, H, W):
    """blobs of 5-9 px radius with ids like a slide-global nucleus table: sparse, up to ~2e6, ascending nowhere in particular"""
    lab = np.zeros((B, H, W), dtype=np.int64)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        if b == 1:
            continue                       # an image without nuclei
        for _ in range(14):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(3, 8)
            ident = int(rng.integers(1, 2_000_000))
            lab[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = ident
    lab[2, :, : W // 2] = 0                # a half-empty image
    lab[3] = np.where(lab[3] > 0, lab[0].max() + 7, 0)   # one nucleus id shared by every blob of image 3 (non-contiguous region)
    return lab


def main():
    load_reference()
    import pandas as pd

    class Metric(nn.Module):               # torchmetrics.Metric: only add_state is used by CellMetrics.__init__/update
        def __init__(self, **kw):
            super().__init__()

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, list(default) if isinstance(default, list) else default)

    sys.modules["torchmetrics"].Metric = Metric
    _exec("refsrc", "metrics", f"{REF}/src/metrics.py")
    Extr = sys.modules["refsrc.utils"].MeanCellExtrator
    CellMetrics = sys.modules["refsrc.metrics"].CellMetrics

    seed, B, C, H, W = 41, 4, 5, 48, 64
    rng = np.random.default_rng(seed)
    nuclei = label_map(rng, B, H, W)
    T = lambda name, shape, std=1.0: torch.from_numpy(np.asarray(det_normal(seed, name, shape, 0.0, std), dtype=np.float32))
    pred, target = torch.tanh(T("pred", (B, C, H, W))), T("target", (B, C, H, W), 0.5).clamp(-0.9, 0.9)
    nt = torch.from_numpy(nuclei)
    rec = dict(seed=seed, B=B, C=C, H=H, W=W, nuclei=nuclei)
    for tag, sf in (("s1", 1.0), ("s05", 0.5), ("s025", 0.25)):
        pm, tm, ids = Extr(scale_factor=sf)(pred, target, nt)            # nuclei [B,H,W]: the 3-d branch of forward
        rec[f"pm_{tag}"], rec[f"tm_{tag}"], rec[f"ids_{tag}"] = pm.numpy(), tm.numpy(), ids.numpy()
    pm, tm, ids = Extr(1.0)(pred, None, nt.unsqueeze(1))                    # target=None, 4-d label map
    rec["pm_notarget"], rec["tm_notarget"] = pm.numpy(), tm.numpy()
    # CellMetrics.update state for two slides (marker 0 is Hoechst -> excluded)
    names = ["Hoechst", "CD31", "CD45", "CD68", "CD4"]
    df = pd.DataFrame({"in_slide_name": ["slideA", "slideB"], "nuclei_csv_path": ["a.csv", "b.csv"]})
    cm = CellMetrics(df, names)
    cm.update(pred, nt, ["slideA", "slideB", "slideA", "slideB"])
    for s in ("slideA", "slideB"):
        rec[f"cm_{s}_n"] = len(getattr(cm, f"{s}_cell_id"))
        for i, (cid, sm, ar) in enumerate(zip(getattr(cm, f"{s}_cell_id"), getattr(cm, f"{s}_sum"), getattr(cm, f"{s}_area"))):
            rec[f"cm_{s}_id{i}"] = cid.numpy().astype(np.int64)
            rec[f"cm_{s}_sum{i}"] = sm.numpy().astype(np.int64)
            rec[f"cm_{s}_area{i}"] = ar.numpy().astype(np.int64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "comp_cells.npz"), **rec)
    print("wrote comp_cells.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in rec.items() if k.startswith(("ids", "cm_slideA_n", "cm_slideB_n"))})


if __name__ == "__main__":
    main()
