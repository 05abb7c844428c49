"""GPU: segmented-reduction cell extractor against the fixture captured from the reference's own MeanCellExtrator /
CellMetrics.update (oracle/make_golden_cells.py) and against the oracle restatement at full tile size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fixture_inputs(g):
    from oracle.detgen import det_normal
    seed, B, C, H, W = (int(g[k]) for k in ("seed", "B", "C", "H", "W"))
    T = lambda name, shape, std=1.0: torch.from_numpy(np.asarray(det_normal(seed, name, shape, 0.0, std), dtype=np.float32))
    return torch.tanh(T("pred", (B, C, H, W))), T("target", (B, C, H, W), 0.5).clamp(-0.9, 0.9), torch.from_numpy(g["nuclei"])


def test_cell_means_match_reference_fixture(golden_dir):
    from miphei_vit_amd.cells import CellMetrics, MeanCellExtrator
    g = np.load(os.path.join(golden_dir, "comp_cells.npz"))
    pred, target, nuclei = _fixture_inputs(g)
    for tag, sf in (("s1", 1.0), ("s05", 0.5), ("s025", 0.25)):
        pm, tm, ids = MeanCellExtrator(sf)(pred.cuda(), target.cuda(), nuclei.cuda())
        assert ids.dtype == torch.long and np.array_equal(ids.cpu().numpy(), g[f"ids_{tag}"])      # ids and order: exact
        assert np.allclose(pm.cpu().numpy(), g[f"pm_{tag}"], atol=1e-5) and np.allclose(tm.cpu().numpy(), g[f"tm_{tag}"], atol=1e-5)
    pm, tm, ids = MeanCellExtrator()(pred.cuda(), None, nuclei.unsqueeze(1).int().cuda())          # int32 labels, 4-d, no target
    assert np.allclose(pm.cpu().numpy(), g["pm_notarget"], atol=1e-5) and float(tm.abs().max()) == 0.0
    empty = MeanCellExtrator()(pred.cuda(), None, torch.zeros_like(nuclei).cuda())
    assert empty[0].shape == (0, pred.shape[1]) and empty[2].numel() == 0
    with pytest.raises(ValueError):
        MeanCellExtrator(1.5)
    cm = CellMetrics(["slideA", "slideB"], ["Hoechst", "CD31", "CD45", "CD68", "CD4"])
    cm.update(pred.cuda(), nuclei.cuda(), ["slideA", "slideB", "slideA", "slideB"])
    for s in ("slideA", "slideB"):
        assert len(cm.state[s]["cell_id"]) == int(g[f"cm_{s}_n"])
        for i in range(int(g[f"cm_{s}_n"])):
            assert np.array_equal(cm.state[s]["cell_id"][i].numpy(), g[f"cm_{s}_id{i}"])
            assert np.array_equal(cm.state[s]["area"][i].numpy(), g[f"cm_{s}_area{i}"])
            assert np.abs(cm.state[s]["sum"][i].numpy() - g[f"cm_{s}_sum{i}"]).max() <= 1    # uint32 truncation of f32 sums


@pytest.mark.parametrize("scale", [1.0, 0.5])
def test_cell_means_full_tile_against_oracle(scale):
    """256x256 tiles, 16 markers, ~150 nuclei per image with slide-global ids, plus a per-pixel label image (every 2x2 block its
    own id: far more distinct labels per tile than hash slots -> the direct-append path)."""
    from oracle.cells import extract_means
    from miphei_vit_amd.cells import MeanCellExtrator
    rng = np.random.default_rng(5)
    B, C, S = 3, 16, 256
    lab = np.zeros((B, S, S), dtype=np.int64)
    yy, xx = np.mgrid[0:S, 0:S]
    for b in range(2):
        for _ in range(150):
            cy, cx, r = rng.integers(0, S), rng.integers(0, S), rng.integers(4, 10)
            lab[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = int(rng.integers(1, 5_000_000))
    lab[2] = 1 + (yy // 2) * (S // 2) + xx // 2
    lab[2, 64:] = 0          # 8 tiles x (256 hashed + up to 512 direct) records < the 8192-record capacity
    pred, target = rng.standard_normal((B, C, S, S), dtype=np.float32), rng.standard_normal((B, C, S, S), dtype=np.float32)
    if scale == 1.0:
        pm, tm, ids = MeanCellExtrator(scale)(torch.from_numpy(pred).cuda(), torch.from_numpy(target).cuda(), torch.from_numpy(lab).cuda())
        rp, rt, ri, _ = extract_means(pred, target, lab, scale)
    else:   # the per-pixel image would exceed the record capacity after a 2x down-sampling of 2x2 blocks: leave it out
        pm, tm, ids = MeanCellExtrator(scale)(torch.from_numpy(pred[:2]).cuda(), torch.from_numpy(target[:2]).cuda(), torch.from_numpy(lab[:2]).cuda())
        rp, rt, ri, _ = extract_means(pred[:2], target[:2], lab[:2], scale)
    assert np.array_equal(ids.cpu().numpy(), ri)
    assert np.allclose(pm.cpu().numpy(), rp, atol=2e-5) and np.allclose(tm.cpu().numpy(), rt, atol=2e-5)


def test_cell_extractor_dense_label_maps_retry_and_chunk():
    """More (tile, nucleus) fragments than the scratch holds: the reference (unique + scatter_add) has no such limit, so the
    wrapper retries with the larger scratch and then on row chunks whose partial sums are merged per nucleus."""
    from oracle.cells import extract_means
    from miphei_vit_amd.cells import MeanCellExtrator
    rng = np.random.default_rng(7)
    S = 256
    pred = rng.standard_normal((2, 3, S, S), dtype=np.float32)
    target = rng.standard_normal((2, 3, S, S), dtype=np.float32)
    lab = np.zeros((2, S, S), dtype=np.int64)
    lab[0] = 1 + np.arange(S * S).reshape(S, S)                    # every pixel its own nucleus: 65536 records (chunked path)
    yy, xx = np.mgrid[0:S, 0:S]
    lab[1] = 7 + (yy // 2) * (S // 2) + xx // 2                    # 2x2 blocks spanning chunk boundaries never; 16384 nuclei
    lab[1, :, 100:140] = 3                                         # one nucleus crossing every row chunk: partials must merge
    pm, tm, ids = MeanCellExtrator()(torch.from_numpy(pred).cuda(), torch.from_numpy(target).cuda(), torch.from_numpy(lab).cuda())
    rp, rt, ri, _ = extract_means(pred, target, lab, 1.0)
    assert np.array_equal(ids.cpu().numpy(), ri)
    assert np.allclose(pm.cpu().numpy(), rp, atol=2e-5) and np.allclose(tm.cpu().numpy(), rt, atol=2e-5)
    # 12288 records: fits the second scratch size without chunking
    lab2 = np.zeros((1, S, S), dtype=np.int64)
    lab2[0, :48] = 1 + np.arange(48 * S).reshape(48, S)
    pm, _, ids = MeanCellExtrator()(torch.from_numpy(pred[:1]).cuda(), None, torch.from_numpy(lab2).cuda())
    rp, _, ri, _ = extract_means(pred[:1], None, lab2, 1.0)
    assert np.array_equal(ids.cpu().numpy(), ri) and np.allclose(pm.cpu().numpy(), rp, atol=2e-5)


def test_cell_extractor_rejects_ids_beyond_int32():
    from miphei_vit_amd.cells import MeanCellExtrator
    lab = torch.zeros(1, 128, 128, dtype=torch.int64)
    lab[0, 3, 3] = 2 ** 31 + 5
    with pytest.raises(ValueError, match="INT32_MAX"):
        MeanCellExtrator()(torch.zeros(1, 2, 128, 128).cuda(), None, lab.cuda())
